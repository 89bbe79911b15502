// host_loci.cpp -- the per-LOCUS host work of step 3 as plain native code (no HIP, no Python API): built by
// warpstr_amd/build.py into warpstr_amd/_host_loci.so and loaded with ctypes.CDLL, so every call runs WITHOUT the GIL and the
// `threads` of main_wrapper_loci are threads of one process.
//
// Upstream does this work per locus in Python: the overview table through pandas (src/caller/overview.py:37-45, 57-115), two
// StateAutomata (src/caller/automata.py:36-226), the FASTA files (overview.py:76-100), the complex-unit table
// (src/caller/wrapper.py:220-248, overview.py:11-34).  A run of thousands of loci with tens of reads each spends 4-5 ms per
// locus there, a hundred times what the GPU needs for the locus's reads.  The Python forms (warpstr_amd/automata.py,
// overview.py, units.py) stay the definition -- they are what the fixtures recorded from upstream pin --; this file restates
// them and tests/test_host_native.py holds the two against each other byte for byte.  Whatever this file is not sure to
// reproduce (an overview.csv that pandas would re-format on its way through, a pattern the Python compiler raises on) it
// REFUSES with a positive status and the caller takes the Python path for that locus.
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_set>
#include <utility>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// repr(float): the shortest digits that round-trip (std::to_chars), laid out by CPython's rule for format code 'r'
// (Python/pystrtod.c: format_float_short -- exponent form iff decpt <= -4 or decpt > 16, at least two exponent digits, ".0"
// appended to a plain integer).  pandas writes a float64 column that way (DataFrame.to_csv without float_format).
int format_repr(double x, char *out)
{
    if (std::isnan(x)) { memcpy(out, "nan", 3); return 3; }
    if (std::isinf(x)) { if (x < 0) { memcpy(out, "-inf", 4); return 4; } memcpy(out, "inf", 3); return 3; }
    char tmp[48];
    auto r = std::to_chars(tmp, tmp + sizeof tmp, x, std::chars_format::scientific);
    *r.ptr = 0;   // (to_chars does not terminate; atoi below reads the exponent)
    const char *p = tmp, *end = r.ptr;
    char *o = out;
    if (*p == '-') { *o++ = '-'; p++; }
    char digits[24] = {'0'};
    int nd = 0;
    const char *e = p;
    while (e < end && *e != 'e') { if (*e != '.') digits[nd++] = *e; e++; }
    const int ex = atoi(e + 1);       // value = d.ddd x 10^ex
    const int decpt = ex + 1;         // value = 0.ddd x 10^decpt
    if (decpt <= -4 || decpt > 16) {
        *o++ = digits[0];
        if (nd > 1) { *o++ = '.'; memcpy(o, digits + 1, nd - 1); o += nd - 1; }
        *o++ = 'e';
        int ax = ex;
        if (ax < 0) { *o++ = '-'; ax = -ax; } else *o++ = '+';
        char eb[8];
        int ne = 0;
        do { eb[ne++] = char('0' + ax % 10); ax /= 10; } while (ax);
        if (ne < 2) eb[ne++] = '0';
        while (ne) *o++ = eb[--ne];
    } else if (decpt <= 0) {
        *o++ = '0'; *o++ = '.';
        for (int i = 0; i < -decpt; i++) *o++ = '0';
        memcpy(o, digits, nd); o += nd;
    } else if (decpt >= nd) {
        memcpy(o, digits, nd); o += nd;
        for (int i = nd; i < decpt; i++) *o++ = '0';
        *o++ = '.'; *o++ = '0';
    } else {
        memcpy(o, digits, decpt); o += decpt;
        *o++ = '.';
        memcpy(o, digits + decpt, nd - decpt); o += nd - decpt;
    }
    return int(o - out);
}

int format_int(int64_t v, char *out)
{
    auto r = std::to_chars(out, out + 24, v);
    return int(r.ptr - out);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Pattern -> k-mer state automaton (warpstr_amd/automata.py, which follows src/caller/automata.py:57-226 node for node).
const char *iupac(char c)
{
    switch (c) {  // src/templates.py:32-44
    case 'R': return "AG"; case 'Y': return "CT"; case 'S': return "GC"; case 'W': return "AT"; case 'K': return "GT";
    case 'M': return "AC"; case 'B': return "CGT"; case 'D': return "AGT"; case 'H': return "ACT"; case 'V': return "ACG";
    case 'N': return "ACGT"; default: return nullptr;
    }
}

int base_code(char c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

struct LoopHead { bool many; int id; std::vector<int> ids; };

struct Automaton {
    int n_states = 0, endstate = -1, repstart = -1, repend = -1;
    std::vector<double> value;
    std::vector<int32_t> seq_idx, pred_ptr, pred_idx;
    std::vector<uint8_t> repeat_mask, last_base;
    std::vector<uint32_t> kmer;
};

// 0 = built; 1 = the Python compiler would raise on this pattern (the caller lets it)
int compile_automaton(const char *pat, int64_t n, const double *levels, int k, Automaton &A)
{
    if (n < 1 || k < 2 || k > 12) return 1;
    // one node per (expanded) base; the edges in the order they are made (a node's successor list is its edges in that order)
    std::vector<char> base;
    base.reserve(size_t(n) + 8);
    base.push_back(pat[0]);
    std::vector<std::pair<int, int>> edges;
    edges.reserve(size_t(n) + 16);
    std::vector<int> tails{0}, opt_tails, pending, created;
    std::vector<LoopHead> heads;
    int rep_first = -1, rep_last = -1, nxt = 1;
    for (int64_t i = 1; i < n; i++) {
        const char ch = pat[i];
        if (ch == '(') {
            heads.push_back({false, nxt, {}});
            if (rep_first == -1) rep_first = nxt;
        } else if (ch == ')') {
            rep_last = nxt;
            if (heads.empty()) return 1;
            LoopHead h = std::move(heads.back());
            heads.pop_back();
            for (int t : tails) {
                if (h.many) for (int id : h.ids) edges.emplace_back(t, id);
                else edges.emplace_back(t, h.id);
            }
        } else if (ch == '{') {
            opt_tails.push_back(nxt - 1);
        } else if (ch == '}') {
            if (opt_tails.empty()) return 1;
            pending.push_back(opt_tails.back());
            opt_tails.pop_back();
        } else if (const char *alts = iupac(ch)) {
            bool opens = !heads.empty() && !heads.back().many && heads.back().id == nxt;
            opens = opens || (!opt_tails.empty() && opt_tails.back() == nxt);
            created.clear();
            for (const char *a = alts; *a; a++) {
                base.push_back(*a);
                created.push_back(nxt);
                for (int t : tails) edges.emplace_back(t, nxt);
                nxt++;
            }
            if (opens) {
                if (heads.empty()) return 1;
                heads.back().many = true;
                heads.back().ids = created;
            }
            tails = created;
        } else {
            base.push_back(ch);
            for (int t : tails) edges.emplace_back(t, nxt);
            tails.assign(1, nxt);
            for (int t : pending) edges.emplace_back(t, nxt);
            pending.clear();
            nxt++;
        }
    }
    const int n_nodes = int(base.size());
    if (n_nodes < k) return 1;
    std::vector<int8_t> code(n_nodes);
    for (int i = 0; i < n_nodes; i++) code[i] = int8_t(base_code(base[i]));
    std::vector<int> sp(n_nodes + 1, 0), sx(edges.size());     // successor lists, CSR (a stable counting sort keeps their order)
    for (auto &e : edges) sp[e.first + 1]++;
    for (int i = 0; i < n_nodes; i++) sp[i + 1] += sp[i];
    {
        std::vector<int> at(sp.begin(), sp.end() - 1);
        for (auto &e : edges) sx[at[e.first]++] = e.second;
    }

    // k-mer states, bucketed by the node of their last base (a linked list per node; a state's slot = its rank in its bucket);
    // depth first with an explicit LIFO stack; the transitions out of a state are one block of `next`
    std::vector<uint32_t> s_code;
    std::vector<int> s_node, s_from, s_slot, s_link, s_nb, s_ne;
    std::vector<std::pair<int, int>> next;   // (node, slot)
    std::vector<int> b_head(n_nodes, -1), b_size(n_nodes, 0);
    const size_t guess = size_t(n_nodes) + 64;
    s_code.reserve(guess); s_node.reserve(guess); s_from.reserve(guess); s_slot.reserve(guess); s_link.reserve(guess);
    s_nb.reserve(guess); s_ne.reserve(guess); next.reserve(guess + 64);
    auto add_state = [&](uint32_t c, int node, int from) {
        const int id = int(s_code.size());
        s_code.push_back(c); s_node.push_back(node); s_from.push_back(from); s_slot.push_back(b_size[node]++);
        s_link.push_back(b_head[node]); s_nb.push_back(0); s_ne.push_back(0);
        b_head[node] = id;
        return id;
    };
    const uint32_t tail_mod = 1u << (2 * (k - 1));
    uint32_t first = 0;
    for (int i = 0; i < k; i++) {
        if (code[i] < 0) return 1;
        first = first * 4 + uint32_t(code[i]);
    }
    std::vector<int> stack{add_state(first, k - 1, -1)};
    while (!stack.empty()) {
        const int cur = stack.back();
        stack.pop_back();
        const uint32_t tail = (s_code[cur] % tail_mod) * 4;
        const int cnode = s_node[cur];
        s_nb[cur] = int(next.size());
        for (int q = sp[cnode]; q < sp[cnode + 1]; q++) {
            const int node = sx[q];
            if (code[node] < 0) return 1;
            const uint32_t km = tail + uint32_t(code[node]);
            bool linked = false;
            for (int o = b_head[node]; o >= 0; o = s_link[o]) {
                if (s_code[o] == km && s_from[o] == cnode) {
                    next.emplace_back(node, s_slot[o]);
                    linked = true;
                }
            }
            if (!linked) {
                const int id = add_state(km, node, cnode);
                next.emplace_back(node, s_slot[id]);
                stack.push_back(id);
            }
        }
        s_ne[cur] = int(next.size());
    }

    // flatten in node order
    std::vector<int> base_of(n_nodes + 1, 0);
    for (int i = 0; i < n_nodes; i++) base_of[i + 1] = base_of[i] + b_size[i];
    const int S = base_of[n_nodes];
    A.n_states = S;
    A.repstart = rep_first;
    A.repend = rep_last;
    A.value.resize(S); A.seq_idx.resize(S); A.repeat_mask.resize(S); A.last_base.resize(S); A.kmer.resize(S);
    std::vector<int> by_flat(S);   // state id at every flat index
    for (int id = 0; id < S; id++) by_flat[base_of[s_node[id]] + s_slot[id]] = id;
    A.pred_ptr.assign(S + 1, 0);
    for (int j = 0; j < S; j++) {
        const int id = by_flat[j], nd = s_node[id];
        A.kmer[j] = s_code[id];
        A.seq_idx[j] = nd;
        A.value[j] = levels[s_code[id]];
        A.repeat_mask[j] = (rep_first - 1 <= nd && nd <= rep_last + 10) ? 1 : 0;
        A.last_base[j] = uint8_t(base[nd]);
        for (int q = s_nb[id]; q < s_ne[id]; q++) A.pred_ptr[base_of[next[q].first] + next[q].second + 1]++;
    }
    A.endstate = b_size[n_nodes - 1] == 0 ? -1 : S - 1;
    for (int j = 0; j < S; j++) A.pred_ptr[j + 1] += A.pred_ptr[j];
    A.pred_idx.resize(A.pred_ptr[S]);
    std::vector<int32_t> fill(A.pred_ptr.begin(), A.pred_ptr.end() - 1);
    for (int j = 0; j < S; j++) {      // sources in state order: `incoming` comes out ordered by source index
        const int id = by_flat[j];
        for (int q = s_nb[id]; q < s_ne[id]; q++) A.pred_idx[fill[base_of[next[q].first] + next[q].second]++] = j;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// overview.csv without pandas.  The file goes through pandas twice upstream (read_csv, then to_csv with four columns
// replaced); here the rows keep their TEXT and only the new columns are formatted -- which equals the pandas round trip exactly
// when every cell is already written the way pandas would write it back.  classify() decides that per column, conservatively.
enum Cell : uint8_t { C_EMPTY = 0, C_INT, C_FLOAT, C_BOOL, C_STR, C_BAD };
enum Col : uint8_t { K_EMPTY = 0, K_INT, K_FLOAT, K_BOOL, K_BOOL_OBJ, K_STR, K_BAD };

bool is_na_token(std::string_view s)
{   // pandas' default NA strings (pandas/_libs/parsers.pyx: STR_NA_VALUES)
    static const std::unordered_set<std::string_view> na = {"#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND",
                                                            "1.#QNAN", "<NA>", "N/A", "NA", "NULL", "NaN", "None", "n/a", "nan", "null"};
    return na.count(s) != 0;
}

bool ieq(std::string_view s, const char *lit)
{
    size_t n = strlen(lit);
    if (s.size() != n) return false;
    for (size_t i = 0; i < n; i++)
        if (tolower((unsigned char)s[i]) != lit[i]) return false;
    return true;
}

// pandas' default float converter (pandas/_libs/src/parser/tokenizer.c: precise_xstrtod, `float_precision=None` since pandas
// 1.2): at most 17 digits -- leading zeros included -- accumulated in a double, then ONE multiplication or division by a cached
// power of ten.  It is not correctly rounded, so a float column does not survive read_csv -> to_csv unchanged
// ("0.0015732835352270625" comes back as "0.001573283535227"); the writer below sends every cell of a float column through
// this function and repr(), which is what the pandas round trip does to it.
double pandas_strtod(std::string_view s)
{
    static double e[309];
    static bool init = false;
    if (!init) {
        for (int i = 0; i <= 308; i++) {
            char b[16];
            snprintf(b, sizeof b, "1e%d", i);
            e[i] = strtod(b, nullptr);
        }
        init = true;
    }
    const char *p = s.data(), *end = s.data() + s.size();
    auto dig = [&](const char *q) { return q < end && *q >= '0' && *q <= '9'; };
    bool negative = false;
    if (p < end && *p == '-') { negative = true; p++; }
    double number = 0.0;
    int exponent = 0, num_digits = 0, num_decimals = 0;
    const int max_digits = 17;
    while (dig(p)) {
        if (num_digits < max_digits) { number = number * 10. + (*p - '0'); num_digits++; }
        else ++exponent;
        p++;
    }
    if (p < end && *p == '.') {
        p++;
        while (num_digits < max_digits && dig(p)) { number = number * 10. + (*p - '0'); p++; num_digits++; num_decimals++; }
        if (num_digits >= max_digits)
            while (dig(p)) ++p;
        exponent -= num_decimals;
    }
    if (negative) number = -number;
    if (p < end && (*p == 'e' || *p == 'E')) {
        bool neg = false;
        ++p;
        if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
        int n = 0, nd = 0;
        while (nd < max_digits && dig(p)) { n = n * 10 + (*p - '0'); nd++; p++; }
        if (neg) exponent -= n; else exponent += n;
    }
    if (exponent > 308) return negative ? -HUGE_VAL : HUGE_VAL;
    if (exponent > 0) number *= e[exponent];
    else if (exponent < -308) {
        if (exponent < -616) number = 0.;
        else { number /= e[-308 - exponent]; number /= e[308]; }
    } else number /= e[-exponent];
    return number;
}

// -?digits+ ( . digits* )? ( [eE] [+-]? digits{1,3} )? with a '.' or an exponent present; or inf / -inf
bool float_grammar(std::string_view s)
{
    if (s == "inf" || s == "-inf") return true;
    size_t i = 0, n = s.size();
    auto dig = [&](size_t q) { return q < n && s[q] >= '0' && s[q] <= '9'; };
    if (i < n && s[i] == '-') i++;
    if (!dig(i)) return false;
    size_t nd = 0;
    while (dig(i)) { i++; nd++; }
    bool marked = false;
    if (i < n && s[i] == '.') { marked = true; i++; while (dig(i)) { i++; nd++; } }
    if (i < n && (s[i] == 'e' || s[i] == 'E')) {
        marked = true;
        i++;
        if (i < n && (s[i] == '+' || s[i] == '-')) i++;
        size_t ne = 0;
        while (dig(i)) { i++; ne++; }
        if (ne < 1 || ne > 3) return false;
    }
    return marked && i == n && nd <= 40;
}

Cell classify_cell(std::string_view s, double *val)
{
    if (s.empty()) return C_EMPTY;
    if (s == "True" || s == "False") { *val = s[0] == 'T'; return C_BOOL; }
    if (is_na_token(s) || ieq(s, "true") || ieq(s, "false")) return C_BAD;   // pandas turns these into something else
    // canonical int64: -?(0|[1-9][0-9]*), no "-0"
    {
        size_t i = 0;
        const bool neg = s[0] == '-';
        if (neg) i = 1;
        if (i < s.size() && s.size() - i <= 18 && s[i] >= '0' && s[i] <= '9' && !(s[i] == '0' && (s.size() - i > 1 || neg))) {
            int64_t v = 0;
            size_t j = i;
            for (; j < s.size() && s[j] >= '0' && s[j] <= '9'; j++) v = v * 10 + (s[j] - '0');
            if (j == s.size()) { *val = double(neg ? -v : v); return C_INT; }
        }
    }
    if (float_grammar(s)) {
        if (s == "inf" || s == "-inf") {
            *val = s[0] == '-' ? -HUGE_VAL : HUGE_VAL;
            return C_FLOAT;
        }
        *val = pandas_strtod(s);
        // (out of range: pandas' converter reports an error there and the column is read as text -- its business)
        return std::isinf(*val) ? C_BAD : C_FLOAT;
    }
    // anything else strtod takes whole (after blanks) is a number to pandas too, or close enough to one not to be trusted as text
    char buf[64];
    if (s.size() < sizeof buf) {
        memcpy(buf, s.data(), s.size());
        buf[s.size()] = 0;
        char *endp = nullptr;
        (void)strtod(buf, &endp);
        while (*endp == ' ' || *endp == '\t') endp++;
        if (endp != buf && *endp == 0) return C_BAD;
    } else {
        bool digits_only = true;   // a very long token of digits and number punctuation: let pandas decide
        for (char c : s) digits_only = digits_only && (isdigit((unsigned char)c) || c == '.' || c == '-' || c == '+' || c == 'e' || c == 'E');
        if (digits_only) return C_BAD;
    }
    for (char c : s)
        if (c == '"' || c == '\r' || c == '\n') return C_BAD;
    if (s.front() == ' ' || s.back() == ' ' || s.front() == '\t' || s.back() == '\t') return C_BAD;
    return C_STR;
}

struct Locus {
    std::string text;                    // overview.csv as read
    int n_rows = 0, n_cols = 0;
    std::vector<int32_t> hb, he;         // header cells
    std::vector<int32_t> cb, ce;         // data cells, row-major
    std::vector<uint8_t> kind;           // per column (Col)
    std::vector<double> num;             // numeric value of a cell (int / float / bool), row-major
    std::vector<uint8_t> cls;            // per cell (Cell)
    int c_name = -1, c_saved = -1, c_reverse = -1, c_lo = -1, c_hi = -1, c_run = -1, c_f5 = -1;
    std::vector<int32_t> saved_rows;
    std::vector<uint8_t> reverse;        // per saved row
    std::vector<int64_t> lo, hi;         // per saved row
    std::string names, runs, f5s;        // per saved row, back to back
    std::vector<int64_t> names_off, runs_off, f5s_off;
    std::string out_text;                // the table after store_results (wsh_locus_store)
    std::string err;
};

std::string_view cell(const Locus &L, int r, int c) { return std::string_view(L.text).substr(L.cb[size_t(r) * L.n_cols + c], L.ce[size_t(r) * L.n_cols + c] - L.cb[size_t(r) * L.n_cols + c]); }
std::string_view head(const Locus &L, int c) { return std::string_view(L.text).substr(L.hb[c], L.he[c] - L.hb[c]); }

bool read_file(const char *path, std::string &out)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return false; }
    out.resize(size_t(sb.st_size));
    size_t got = 0;
    while (got < out.size()) {
        const ssize_t n = read(fd, &out[got], out.size() - got);
        if (n <= 0) break;
        got += size_t(n);
    }
    close(fd);
    out.resize(got);
    return true;
}

bool write_file(const std::string &path, const std::string &data)
{
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return false;
    size_t put = 0;
    while (put < data.size()) {
        const ssize_t n = write(fd, data.data() + put, data.size() - put);
        if (n <= 0) { close(fd); return false; }
        put += size_t(n);
    }
    return close(fd) == 0;
}

bool make_dirs(const std::string &path)
{   // os.makedirs(path, exist_ok=True)
    struct stat sb;
    if (stat(path.c_str(), &sb) == 0) return S_ISDIR(sb.st_mode);
    const size_t cut = path.find_last_of('/');
    if (cut != std::string::npos && cut > 0 && !make_dirs(path.substr(0, cut))) return false;
    return mkdir(path.c_str(), 0777) == 0 || errno == EEXIST;
}

// 0 ok; > 0: not sure to equal the pandas round trip (reason in L.err) -> the caller takes the pandas path
int parse_overview(Locus &L)
{
    const std::string &t = L.text;
    const size_t n = t.size();
    if (n == 0 || t.back() != '\n') { L.err = "no final newline"; return 2; }
    if (n >= (size_t(1) << 31)) { L.err = "file too large"; return 2; }
    if (memchr(t.data(), '"', n) || memchr(t.data(), '\r', n)) { L.err = "quoted fields or carriage returns"; return 2; }
    // header
    size_t pos = 0;
    {
        size_t b = 0;
        for (;; pos++) {
            if (t[pos] == ',' || t[pos] == '\n') {
                L.hb.push_back(int32_t(b));
                L.he.push_back(int32_t(pos));
                b = pos + 1;
                if (t[pos] == '\n') break;
            }
        }
        pos++;
    }
    L.n_cols = int(L.hb.size());
    for (int c = 0; c < L.n_cols; c++) {
        std::string_view h = head(L, c);
        if (h.empty() || h.front() == ' ' || h.back() == ' ') { L.err = "empty or padded column name"; return 2; }
        for (int d = 0; d < c; d++)
            if (head(L, d) == h) { L.err = "duplicate column name"; return 2; }
        if (h == "read_name") L.c_name = c;
        else if (h == "saved") L.c_saved = c;
        else if (h == "reverse") L.c_reverse = c;
        else if (h == "l_start_raw") L.c_lo = c;
        else if (h == "r_end_raw") L.c_hi = c;
        else if (h == "run_id") L.c_run = c;
        else if (h == "fast5_path") L.c_f5 = c;
    }
    if (L.c_name < 0 || L.c_saved < 0 || L.c_reverse < 0 || L.c_lo < 0 || L.c_hi < 0) { L.err = "a column the caller needs is missing"; return 3; }
    // rows
    while (pos < n) {
        int c = 0;
        size_t b = pos;
        for (;; pos++) {
            if (t[pos] == ',' || t[pos] == '\n') {
                if (c >= L.n_cols) { L.err = "a row with too many fields"; return 2; }
                L.cb.push_back(int32_t(b));
                L.ce.push_back(int32_t(pos));
                c++;
                b = pos + 1;
                if (t[pos] == '\n') break;
            }
        }
        pos++;
        if (c != L.n_cols) { L.err = c == 1 && L.ce.back() == L.cb.back() ? "blank line" : "a row with too few fields"; return 2; }
        L.n_rows++;
    }
    // classify
    L.cls.resize(L.cb.size());
    L.num.assign(L.cb.size(), 0.0);
    L.kind.resize(L.n_cols);
    for (int c = 0; c < L.n_cols; c++) {
        int cnt[6] = {0, 0, 0, 0, 0, 0};
        for (int r = 0; r < L.n_rows; r++) {
            const size_t i = size_t(r) * L.n_cols + c;
            L.cls[i] = classify_cell(cell(L, r, c), &L.num[i]);
            cnt[L.cls[i]]++;
        }
        Col k;
        if (cnt[C_BAD]) k = K_BAD;
        else if (cnt[C_EMPTY] == L.n_rows) k = K_EMPTY;
        else if (cnt[C_INT] == L.n_rows) k = K_INT;
        else if (cnt[C_FLOAT] + cnt[C_INT] + cnt[C_EMPTY] == L.n_rows) k = K_FLOAT;   // ints beside floats or blanks: a float column
        else if (cnt[C_BOOL] == L.n_rows) k = K_BOOL;
        else if (cnt[C_BOOL] && cnt[C_BOOL] + cnt[C_EMPTY] == L.n_rows) k = K_BOOL_OBJ;
        else if (cnt[C_STR] && cnt[C_STR] + cnt[C_EMPTY] == L.n_rows) k = K_STR;
        else k = K_BAD;   // numbers beside text, booleans beside numbers
        L.kind[c] = k;
        if (k == K_FLOAT)
            for (int r = 0; r < L.n_rows; r++) {
                const size_t i = size_t(r) * L.n_cols + c;
                if (L.cls[i] == C_INT) L.num[i] = pandas_strtod(cell(L, r, c));
            }
        if (k == K_BAD) { L.err = "column '" + std::string(head(L, c)) + "' would be re-formatted by pandas"; return 2; }
    }
    if (L.n_rows == 0) { L.err = "no rows"; return 2; }   // (pandas makes every column of an empty table `object`)
    if (L.kind[L.c_name] == K_EMPTY || L.kind[L.c_name] == K_BOOL_OBJ || L.kind[L.c_name] == K_FLOAT) { L.err = "read names are not plain"; return 2; }
    for (int c : {L.c_run, L.c_f5})   // (a float column: str() of what pandas' converter made of the text -- leave it to pandas)
        if (c >= 0 && L.kind[c] == K_FLOAT) { L.err = "run_id / fast5_path is a column of floats"; return 3; }
    for (int c : {L.c_name, L.c_run, L.c_f5})   // str(nan) is what the Python path would make of a blank: leave those tables to it
        for (int r = 0; c >= 0 && r < L.n_rows; r++)
            if (L.cls[size_t(r) * L.n_cols + c] == C_EMPTY) { L.err = "blank read name / run_id / fast5_path"; return 3; }
    // the columns the caller reads: np.asarray(df[col]).astype(bool / int64) of an int, float or bool column
    auto truthy = [&](int c) { return L.kind[c] == K_INT || L.kind[c] == K_BOOL; };
    if (!truthy(L.c_saved) || !truthy(L.c_reverse)) { L.err = "`saved` / `reverse` are not integer or boolean columns"; return 3; }
    for (int c : {L.c_lo, L.c_hi})
        if (L.kind[c] != K_INT && L.kind[c] != K_FLOAT) { L.err = "`l_start_raw` / `r_end_raw` are not numeric columns"; return 3; }
    for (int r = 0; r < L.n_rows; r++) {
        if (L.num[size_t(r) * L.n_cols + L.c_saved] == 0.0) continue;
        const size_t i = size_t(r) * L.n_cols;
        for (int c : {L.c_lo, L.c_hi}) {
            const double v = L.num[i + c];
            if (L.cls[i + c] == C_EMPTY || !(std::fabs(v) < 9.0e15)) { L.err = "a saved read without l_start_raw / r_end_raw"; return 3; }
        }
        L.saved_rows.push_back(r);
        L.reverse.push_back(L.num[i + L.c_reverse] != 0.0);
        L.lo.push_back(int64_t(L.num[i + L.c_lo]));   // (astype(int64) truncates towards zero, as the cast does)
        L.hi.push_back(int64_t(L.num[i + L.c_hi]));
    }
    auto gather = [&](int c, std::string &blob, std::vector<int64_t> &off) {
        off.push_back(0);
        if (c < 0) return;
        for (int r : L.saved_rows) {
            blob.append(cell(L, r, c));
            off.push_back(int64_t(blob.size()));
        }
    };
    gather(L.c_name, L.names, L.names_off);
    gather(L.c_run, L.runs, L.runs_off);
    gather(L.c_f5, L.f5s, L.f5s_off);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The rest of a locus's set-up: the flank file (overview.py: load_flanks; src/extractor/tr_extractor.py:108-140), the reverse
// strand's pattern (automata.py: reverse_pattern; src/caller/wrapper.py:78-84) and summaries/state_similarity.csv with
// upstream's warnings (caller.py: similarity_report; src/caller/wrapper.py:122-160, src/squiggler/pore_model.py:34-71).
char strand_swap(char c)
{
    switch (c) {  // src/templates.py:45-65
    case '(': return ')'; case ')': return '('; case '{': return '}'; case '}': return '{';
    case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G'; case 'M': return 'K'; case 'K': return 'M';
    case 'N': return 'N'; case 'W': return 'W'; case 'S': return 'S'; case 'R': return 'Y'; case 'Y': return 'R'; case 'B': return 'V';
    case 'D': return 'H'; case 'H': return 'D'; case 'V': return 'B'; default: return 0;
    }
}

bool reverse_pattern(const std::string &seq, std::string &out)
{
    out.clear();
    for (size_t i = seq.size(); i-- > 0;) {
        const char c = strand_swap(seq[i]);
        if (!c) return false;
        out.push_back(c);
    }
    return true;
}

std::string rstrip(std::string_view s)
{
    size_t n = s.size();
    while (n && (s[n - 1] == ' ' || s[n - 1] == '\t' || s[n - 1] == '\r' || s[n - 1] == '\n' || s[n - 1] == '\f' || s[n - 1] == '\v')) n--;
    return std::string(s.substr(0, n));
}

// the four flank sequences; false = let the Python reader run (it raises what upstream raises, or handles what this does not)
bool load_flanks(const std::string &path, std::string out[4])
{
    std::string t;
    if (!read_file(path.c_str(), t)) return false;
    for (unsigned char c : t)
        if (c >= 0x80 || c == '\r') return false;   // (text-mode decoding / universal newlines: Python's business)
    std::vector<std::string_view> lines;
    size_t b = 0;
    while (b < t.size() && lines.size() < 5) {
        size_t e = t.find('\n', b);
        if (e == std::string::npos) e = t.size();
        lines.push_back(std::string_view(t).substr(b, e - b));
        b = e + 1;
    }
    if (lines.size() < 5) return false;
    auto split = [](std::string_view l) {
        std::vector<std::string_view> f;
        size_t b = 0;
        for (;;) {
            size_t e = l.find(',', b);
            if (e == std::string_view::npos) { f.push_back(l.substr(b)); break; }
            f.push_back(l.substr(b, e - b));
            b = e + 1;
        }
        return f;
    };
    size_t seqid = 0;
    {
        auto cols = split(lines[0]);
        for (size_t i = 0; i < cols.size(); i++)
            if (rstrip(cols[i]) == "sequence") { seqid = i; break; }
    }
    for (int r = 0; r < 4; r++) {
        auto f = split(lines[r + 1]);
        if (seqid >= f.size()) return false;
        out[r] = rstrip(f[seqid]);
    }
    return true;
}

double np_pairwise_sum(const double *a, int n)
{   // np.add.reduce of up to 128 doubles (numpy: pairwise_sum)
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

// {unit: (mean, median)} in insertion order for one strand's pattern; false = the Python form must run (it raises, or the
// unit is longer than this restates)
bool unit_diffs(const std::string &seq, const double *levels, int k, std::vector<std::string> &keys, std::vector<std::pair<double, double>> &vals)
{
    size_t pos = 0;
    while (pos < seq.size()) {
        size_t i = seq.find_first_of("({", pos);
        if (i == std::string::npos) break;
        size_t j = seq.find_first_of(")}", i + 1);
        if (j == std::string::npos) break;
        std::vector<std::string> pats{""};
        for (size_t q = i; q <= j; q++) {
            const char c = seq[q];
            if (c == '(' || c == ')' || c == '{' || c == '}') continue;
            if (const char *alts = iupac(c)) {
                std::vector<std::string> nx;
                for (const char *a = alts; *a; a++)
                    for (auto &p : pats) nx.push_back(p + *a);
                pats.swap(nx);
            } else {
                for (auto &p : pats) p.push_back(c);
            }
            if (pats.size() > 4096) return false;
        }
        for (auto &p : pats) {
            const int L = int(p.size());
            if (L + k > L * k || L > 128) return false;   // (a one-base unit: IndexError upstream and in the Python form)
            double v[130], d[130];
            for (int a = 0; a <= L; a++) {
                uint32_t code = 0;
                for (int b = 0; b < k; b++) {
                    const int bc = base_code(p[size_t(a + b) % size_t(L)]);
                    if (bc < 0) return false;
                    code = code * 4 + uint32_t(bc);
                }
                v[a] = levels[code];
            }
            for (int a = 0; a < L; a++) d[a] = std::fabs(v[a + 1] - v[a]);
            const double mean = np_pairwise_sum(d, L) / L;
            std::sort(d, d + L);
            const double med = (L & 1) ? d[L / 2] : (d[L / 2 - 1] + d[L / 2]) / 2.0;
            size_t at = 0;
            while (at < keys.size() && keys[at] != p) at++;
            if (at == keys.size()) { keys.push_back(p); vals.emplace_back(mean, med); }
            else vals[at] = {mean, med};
        }
        pos = j + 1;
    }
    return true;
}

bool similarity(const std::string &seq, const std::string &rev, const double *levels, int k, double lim, std::string &csv, std::string &warn)
{
    std::vector<std::string> kt, kr;
    std::vector<std::pair<double, double>> vt, vr;
    if (!unit_diffs(seq, levels, k, kt, vt) || !unit_diffs(rev, levels, k, kr, vr)) return false;
    csv = "pattern,strand,mean_diff,median_diff\n";
    char buf[96];
    for (size_t i = 0; i < kt.size(); i++) {
        snprintf(buf, sizeof buf, ",template,%.3f,%.3f\n", vt[i].first, vt[i].second);
        csv += kt[i]; csv += buf;
    }
    for (size_t i = 0; i < kr.size(); i++) {
        snprintf(buf, sizeof buf, ",reverse,%.3f,%.3f\n", vr[i].first, vr[i].second);
        csv += kr[i]; csv += buf;
    }
    warn.clear();
    for (size_t i = 0; i < kt.size(); i++)
        if (lim > vt[i].first || lim > vt[i].second) warn += "Warning: Template has repeat unit " + kt[i] + " with high state similarity\n";
    for (size_t i = 0; i < kr.size(); i++)
        if (lim > vr[i].first || lim > vr[i].second) warn += "Warning: high similarity of state values in reverse pattern " + kr[i] + "\n";
    return true;
}

struct Setup {
    Locus *locus = nullptr;
    Automaton aut[2];
    std::string sim_csv, warnings;
};

bool starts_with(std::string_view s, const char *p) { return s.size() >= strlen(p) && memcmp(s.data(), p, strlen(p)) == 0; }

}  // namespace

extern "C" {

#define WSH_EXPORT __attribute__((visibility("default")))

struct wsh_automaton {   // what AutomatonTable (warpstr_amd/automata.py) holds; arrays owned by the object behind `owner`
    int32_t n_states, endstate, repstart, repend, n_edges, reserved;
    const double *value;
    const int32_t *seq_idx, *pred_ptr, *pred_idx;
    const uint8_t *repeat_mask, *last_base;
    const uint32_t *kmer;
    void *owner;
};

struct wsh_overview_info {
    int32_t n_rows, n_cols, n_saved, has_run_id, has_fast5_path, reserved;
    const int32_t *saved_rows;
    const uint8_t *reverse;
    const int64_t *lo, *hi;
    const char *names; const int64_t *names_off;
    const char *runs; const int64_t *runs_off;
    const char *f5s; const int64_t *f5s_off;
};

WSH_EXPORT int wsh_abi_version(void) { return 2; }   // (2: the exports guard against C++ exceptions; wsh_loci_counts / _columns, wsh_vbz_unpack, wsh_vbz_context)

WSH_EXPORT int wsh_format_float(double x, char *out) { return format_repr(x, out); }

// status 0: *out filled (free with wsh_automaton_free); 1: the Python compiler would raise on this pattern -- run it
static int wsh_automaton_compile_impl(const char *pattern, int64_t len, const double *levels, int32_t k, wsh_automaton *out);
WSH_EXPORT int wsh_automaton_compile(const char *pattern, int64_t len, const double *levels, int32_t k, wsh_automaton *out)
{
    // (no C++ exception may cross into ctypes: std::bad_alloc and friends become this function's error value)
    try { return wsh_automaton_compile_impl(pattern, len, levels, k, out); } catch (...) { return 1; }
}
static int wsh_automaton_compile_impl(const char *pattern, int64_t len, const double *levels, int32_t k, wsh_automaton *out)
{
    Automaton *A = new Automaton;
    const int rc = compile_automaton(pattern, len, levels, k, *A);
    if (rc != 0) { delete A; return rc; }
    out->n_states = A->n_states; out->endstate = A->endstate; out->repstart = A->repstart; out->repend = A->repend;
    out->n_edges = int32_t(A->pred_idx.size()); out->reserved = 0;
    out->value = A->value.data(); out->seq_idx = A->seq_idx.data(); out->pred_ptr = A->pred_ptr.data(); out->pred_idx = A->pred_idx.data();
    out->repeat_mask = A->repeat_mask.data(); out->last_base = A->last_base.data(); out->kmer = A->kmer.data();
    out->owner = A;
    return 0;
}

WSH_EXPORT void wsh_automaton_free(wsh_automaton *a)
{
    delete static_cast<Automaton *>(a->owner);
    a->owner = nullptr;
}

// <locus_path>/overview.csv read and parsed.  0: *out is a handle (wsh_locus_free); -1: no such file; > 0: the table is one
// pandas might re-format or read differently (reason: wsh_locus_error) -- the handle is still returned for the message and the
// caller takes the pandas path.
WSH_EXPORT int wsh_locus_open(const char *overview_path, void **out)
{
    *out = nullptr;
    Locus *L = nullptr;
    try {
        L = new Locus;
        *out = L;
        if (!read_file(overview_path, L->text)) { L->err = "cannot read the file"; return -1; }
        return parse_overview(*L);
    } catch (...) {   // (out of memory on a huge table: declined -- 1 --, the pandas path reports what it finds; *out may be null)
        if (L) { try { L->err = "out of memory while parsing the table"; } catch (...) {} }
        return 1;
    }
}

WSH_EXPORT const char *wsh_locus_error(void *h) { return h ? static_cast<Locus *>(h)->err.c_str() : "out of memory"; }

WSH_EXPORT void wsh_locus_free(void *h) { delete static_cast<Locus *>(h); }

WSH_EXPORT void wsh_locus_info(void *h, wsh_overview_info *o)
{
    const Locus &L = *static_cast<Locus *>(h);
    o->n_rows = L.n_rows; o->n_cols = L.n_cols; o->n_saved = int32_t(L.saved_rows.size());
    o->has_run_id = L.c_run >= 0; o->has_fast5_path = L.c_f5 >= 0; o->reserved = 0;
    o->saved_rows = L.saved_rows.data(); o->reverse = L.reverse.data(); o->lo = L.lo.data(); o->hi = L.hi.data();
    o->names = L.names.data(); o->names_off = L.names_off.data();
    o->runs = L.runs.data(); o->runs_off = L.runs_off.data();
    o->f5s = L.f5s.data(); o->f5s_off = L.f5s_off.data();
}

// The same for a chunk of loci in two calls instead of one and a dozen copies per locus (a run of thousands of loci spent more time
// taking the columns over, locus by locus under the interpreter's lock, than parsing them): first the sizes -- per locus the saved
// rows, all rows and flags (1 = has a run_id column, 2 = has a fast5_path column), and the chunk's bytes of names, run ids, paths --
// then the columns of all loci back to back into arrays of those sizes; the *_off arrays hold total + 1 offsets into the chunk's
// blobs (a locus without the column: empty strings).
WSH_EXPORT void wsh_loci_counts(int32_t n, void *const *handles, int64_t *n_saved, int64_t *n_rows, int32_t *flags, int64_t *string_bytes)
{
    string_bytes[0] = string_bytes[1] = string_bytes[2] = 0;
    for (int32_t i = 0; i < n; i++) {
        const Locus &L = *static_cast<Locus *>(handles[i]);
        n_saved[i] = int64_t(L.saved_rows.size());
        n_rows[i] = L.n_rows;
        flags[i] = (L.c_run >= 0 ? 1 : 0) | (L.c_f5 >= 0 ? 2 : 0);
        string_bytes[0] += int64_t(L.names.size());
        string_bytes[1] += int64_t(L.runs.size());
        string_bytes[2] += int64_t(L.f5s.size());
    }
}

WSH_EXPORT void wsh_loci_columns(int32_t n, void *const *handles, int64_t *saved, uint8_t *reverse, int64_t *lo, int64_t *hi, char *names,
                                 int64_t *names_off, char *runs, int64_t *runs_off, char *f5s, int64_t *f5s_off)
{
    int64_t at = 0, b0 = 0, b1 = 0, b2 = 0;
    auto strings = [](const std::string &blob, const std::vector<int64_t> &off, int64_t rows, char *out, int64_t *out_off, int64_t at, int64_t &base) {
        const bool have = int64_t(off.size()) == rows + 1;
        for (int64_t r = 0; r < rows; r++) out_off[at + r] = base + (have ? off[size_t(r)] : 0);
        if (have && !blob.empty()) memcpy(out + base, blob.data(), blob.size());
        base += have ? int64_t(blob.size()) : 0;
    };
    for (int32_t i = 0; i < n; i++) {
        const Locus &L = *static_cast<Locus *>(handles[i]);
        const int64_t rows = int64_t(L.saved_rows.size());
        for (int64_t r = 0; r < rows; r++) saved[at + r] = L.saved_rows[size_t(r)];
        if (rows) {
            memcpy(reverse + at, L.reverse.data(), size_t(rows));
            memcpy(lo + at, L.lo.data(), size_t(rows) * 8);
            memcpy(hi + at, L.hi.data(), size_t(rows) * 8);
        }
        strings(L.names, L.names_off, rows, names, names_off, at, b0);
        strings(L.runs, L.runs_off, rows, runs, runs_off, at, b1);
        strings(L.f5s, L.f5s_off, rows, f5s, f5s_off, at, b2);
        at += rows;
    }
    names_off[at] = b0;
    runs_off[at] = b1;
    f5s_off[at] = b2;
}

// The text of overview.csv (the bytes that were read), for the DataFrame a caller may ask for later.
WSH_EXPORT const char *wsh_locus_text(void *h, int64_t *len)
{
    const Locus &L = *static_cast<Locus *>(h);
    *len = int64_t(L.text.size());
    return L.text.data();
}

// store_results (warpstr_amd/overview.py; src/caller/overview.py:48-115) for the locus behind `h`: the four result columns
// (len2 -> results, len1 -> orig, cost1 -> dtw_cost1, cost2 -> dtw_cost2; -1 on rows that are not saved; a NaN cost is an empty
// field) replace / join the table's columns the way DataFrame assignment does after the `result*` columns were dropped, and the
// three FASTA files get the called sequences (seq2[off2[r] .. off2[r] + len2[r]) of saved read r).  flags: 1 = write overview.csv,
// 2 = write the FASTA files.  If `table_out` is not NULL the new overview text is returned there (malloc'ed: wsh_free).
// 0 = done; -2 = a file could not be written (wsh_locus_error).
static int wsh_locus_store_impl(void *h, const char *locus_path, const int32_t *len1, const int32_t *len2, const double *cost1,
                               const double *cost2, const uint8_t *seq2, const int64_t *off2, int32_t flags);
WSH_EXPORT int wsh_locus_store(void *h, const char *locus_path, const int32_t *len1, const int32_t *len2, const double *cost1,
                               const double *cost2, const uint8_t *seq2, const int64_t *off2, int32_t flags)
{
    // (no C++ exception may cross into ctypes: std::bad_alloc and friends become this function's error value)
    try { return wsh_locus_store_impl(h, locus_path, len1, len2, cost1, cost2, seq2, off2, flags); } catch (...) { return -2; }
}
static int wsh_locus_store_impl(void *h, const char *locus_path, const int32_t *len1, const int32_t *len2, const double *cost1,
                               const double *cost2, const uint8_t *seq2, const int64_t *off2, int32_t flags)
{
    Locus &L = *static_cast<Locus *>(h);
    const int ns = int(L.saved_rows.size());
    // output columns: the index first, then the surviving columns in their order, the four new ones in place or at the end
    static const char *const NEW[4] = {"results", "orig", "dtw_cost1", "dtw_cost2"};
    std::vector<int> src;   // >= 0: source column; -1 - k: new column k
    for (int c = 0; c < L.n_cols; c++) {
        if (c == L.c_name) continue;
        std::string_view hd = head(L, c);
        if (starts_with(hd, "result")) continue;
        int k = -1;
        for (int q = 1; q < 4; q++)
            if (hd == NEW[q]) k = q;
        src.push_back(k >= 0 ? -1 - k : c);
    }
    for (int k = 0; k < 4; k++)
        if (std::find(src.begin(), src.end(), -1 - k) == src.end()) src.push_back(-1 - k);
    std::string &out = L.out_text;
    out.clear();
    out.reserve(L.text.size() + size_t(L.n_rows) * 64 + 64);
    out.append("read_name");
    for (int s : src) {
        out.push_back(',');
        if (s >= 0) out.append(head(L, s)); else out.append(NEW[-1 - s]);
    }
    out.push_back('\n');
    char buf[48];
    int next_saved = 0;
    for (int r = 0; r < L.n_rows; r++) {
        const bool sv = next_saved < ns && L.saved_rows[next_saved] == r;
        out.append(cell(L, r, L.c_name));
        for (int s : src) {
            out.push_back(',');
            if (s >= 0) {
                const size_t i = size_t(r) * L.n_cols + s;
                if (L.kind[s] == K_FLOAT && L.cls[i] != C_EMPTY) out.append(buf, size_t(format_repr(L.num[i], buf)));   // what pandas' parser made of it
                else out.append(cell(L, r, s));
                continue;
            }
            const int k = -1 - s;
            int n;
            if (k < 2) n = format_int(sv ? (k == 0 ? len2[next_saved] : len1[next_saved]) : -1, buf);
            else {
                const double v = sv ? (k == 2 ? cost1[next_saved] : cost2[next_saved]) : -1.0;
                n = std::isnan(v) ? 0 : format_repr(v, buf);
            }
            out.append(buf, size_t(n));
        }
        out.push_back('\n');
        next_saved += sv;
    }
    const std::string root(locus_path);
    if (flags & 2) {
        const std::string dir = root + "/predictions/sequences";
        if (!make_dirs(dir)) { L.err = "cannot create " + dir; return -2; }
        std::string all, tmpl, rev;
        for (int i = 0; i < ns; i++) {
            std::string rec(">");
            rec.append(L.names.data() + L.names_off[i], size_t(L.names_off[i + 1] - L.names_off[i]));
            rec.push_back('\n');
            rec.append(reinterpret_cast<const char *>(seq2) + off2[i], size_t(len2[i] > 0 ? len2[i] : 0));
            rec.append("\n\n");
            all.append(rec);
            (L.reverse[i] ? rev : tmpl).append(rec);
        }
        if (!write_file(dir + "/all.fasta", all) || !write_file(dir + "/sequences_template.fasta", tmpl) ||
            !write_file(dir + "/sequences_reverse.fasta", rev)) { L.err = "cannot write the FASTA files under " + dir; return -2; }
    }
    if ((flags & 1) && !write_file(root + "/overview.csv", out)) { L.err = "cannot write " + root + "/overview.csv"; return -2; }
    return 0;
}

// The table wsh_locus_store made (the text of the new overview.csv), for the DataFrame a caller may ask for later.
WSH_EXPORT const char *wsh_locus_table(void *h, int64_t *len)
{
    const Locus &L = *static_cast<Locus *>(h);
    *len = int64_t(L.out_text.size());
    return L.out_text.data();
}

// wsh_locus_store for n loci in one call (one call without the GIL per chunk of loci): locus i's reads are
// [start[i], start[i + 1]) of the run's per-read arrays, off2 absolute into seq2.  status[i] receives each locus's return value;
// returns the number of loci that failed.
WSH_EXPORT int wsh_loci_store(int32_t n, void *const *handles, const char *const *locus_paths, const int64_t *start, const int32_t *len1,
                              const int32_t *len2, const double *cost1, const double *cost2, const uint8_t *seq2, const int64_t *off2,
                              int32_t flags, int32_t *status)
{
    int bad = 0;
    for (int i = 0; i < n; i++) {
        const int64_t a = start[i];
        status[i] = wsh_locus_store(handles[i], locus_paths[i], len1 + a, len2 + a, cost1 + a, cost2 + a, seq2, off2 + a, flags);
        bad += status[i] != 0;
    }
    return bad;
}

struct wsh_setup {
    void *owner;                 // wsh_setup_free
    void *locus;                 // overview handle for wsh_locus_* (owned by `owner`), valid when overview_status == 0
    int32_t overview_status;     // 0 parsed; > 0 declined (wsh_locus_error(locus)): the pandas path; -1 no such file
    int32_t automata_status;     // 0 built; 1: the Python loader / compiler must run (and raises what upstream raises)
    int32_t similarity_status;   // 0 done (and written if asked); 1: the Python form must run
    int32_t reserved;
    wsh_automaton aut[2];        // template, reverse (arrays owned by `owner`)
    const char *similarity_csv;  // text of summaries/state_similarity.csv
    const char *warnings;        // upstream's high-similarity warnings, one per line
};

// Everything main_wrapper does for a locus before its reads are called, in one call without the GIL: overview.csv parsed,
// the flank file read, both automata compiled, summaries/state_similarity.csv made (flags & 1: and written).
static void wsh_locus_setup_impl(const char *locus_path, const char *sequence, const double *levels, int32_t k, double min_state_similarity,
                                 int32_t flags, wsh_setup *out);
WSH_EXPORT void wsh_locus_setup(const char *locus_path, const char *sequence, const double *levels, int32_t k, double min_state_similarity,
                                int32_t flags, wsh_setup *out)
{
    memset(out, 0, sizeof *out);
    try {
        wsh_locus_setup_impl(locus_path, sequence, levels, k, min_state_similarity, flags, out);
    } catch (...) {   // (std::bad_alloc: everything the library had not finished is the Python form's to do -- or to fail on)
        out->automata_status = out->similarity_status = 1;   // (a table that was parsed before the failure stays parsed)
        static const char empty[1] = {0};
        out->similarity_csv = out->warnings = empty;
    }
}
static void wsh_locus_setup_impl(const char *locus_path, const char *sequence, const double *levels, int32_t k, double min_state_similarity,
                                 int32_t flags, wsh_setup *out)
{
    Setup *S = new Setup;
    out->owner = S;
    S->locus = new Locus;
    out->locus = S->locus;
    out->overview_status = 1;   // (until the table is parsed)
    const std::string root(locus_path), seq(sequence);
    if (!read_file((root + "/overview.csv").c_str(), S->locus->text)) { S->locus->err = "cannot read the file"; out->overview_status = -1; }
    else out->overview_status = parse_overview(*S->locus);
    std::string fl[4], rev;
    out->automata_status = 1;
    const bool have_rev = reverse_pattern(seq, rev);
    if (have_rev && load_flanks(root + "/expected_signals/sequences.csv", fl)) {
        const std::string t = fl[0] + seq + fl[1], r = fl[2] + rev + fl[3];
        if (compile_automaton(t.data(), int64_t(t.size()), levels, k, S->aut[0]) == 0 &&
            compile_automaton(r.data(), int64_t(r.size()), levels, k, S->aut[1]) == 0) {
            out->automata_status = 0;
            for (int a = 0; a < 2; a++) {
                Automaton &A = S->aut[a];
                wsh_automaton &o = out->aut[a];
                o.n_states = A.n_states; o.endstate = A.endstate; o.repstart = A.repstart; o.repend = A.repend;
                o.n_edges = int32_t(A.pred_idx.size()); o.reserved = 0;
                o.value = A.value.data(); o.seq_idx = A.seq_idx.data(); o.pred_ptr = A.pred_ptr.data(); o.pred_idx = A.pred_idx.data();
                o.repeat_mask = A.repeat_mask.data(); o.last_base = A.last_base.data(); o.kmer = A.kmer.data();
                o.owner = nullptr;
            }
        }
    }
    out->similarity_status = 1;
    if (have_rev && similarity(seq, rev, levels, k, min_state_similarity, S->sim_csv, S->warnings)) {
        out->similarity_status = 0;
        if (flags & 1) {
            const std::string dir = root + "/summaries";
            if (!make_dirs(dir) || !write_file(dir + "/state_similarity.csv", S->sim_csv)) out->similarity_status = 1;   // (Python reports the OSError)
        }
    }
    out->similarity_csv = S->sim_csv.c_str();
    out->warnings = S->warnings.c_str();
}

// wsh_locus_setup for n loci in one call.
WSH_EXPORT void wsh_loci_setup(int32_t n, const char *const *locus_paths, const char *const *sequences, const double *levels, int32_t k,
                               double min_state_similarity, int32_t flags, wsh_setup *out)
{
    for (int i = 0; i < n; i++) wsh_locus_setup(locus_paths[i], sequences[i], levels, k, min_state_similarity, flags, out + i);
}

WSH_EXPORT void wsh_setup_free(wsh_setup *s)
{
    Setup *S = static_cast<Setup *>(s->owner);
    if (S) { delete S->locus; delete S; }
    s->owner = s->locus = nullptr;
}

WSH_EXPORT void wsh_free(void *p) { free(p); }

// collapse_repeats (warpstr_amd/units.py; src/caller/wrapper.py:220-248) for n called sequences at once and the table of
// store_collapsed (overview.py:11-34) as CSV text with pandas' default index column.  units: n_units repeat units, unit u has
// n_alt[u] alternative strings (alts, back to back; alt_off) and is preceded by offsets[u] plain bases.  counts (n x total alts,
// row-major) receives every count.  header: the column names, comma separated, without the index column and the final newline;
// sel[c]: which of the generated columns (per unit: the sum and one per further alternative, or the single count) output column
// c shows -- units of the same name share a column, as the keys of upstream's dict do.
// flags 1: write <locus_path>/predictions/complexSTR_analysis/complex_repeat_units.csv.  Returns 0, or -2 (cannot write), or
// 1 (an empty alternative: the Python form raises after its iteration limit -- let it).
static int wsh_collapse_store_impl(const char *locus_path, int32_t n, const uint8_t *seq, const int64_t *off, const int32_t *len,
                                  const uint8_t *reverse, int32_t n_units, const int32_t *n_alt, const char *alts,
                                  const int32_t *alt_off, const int32_t *offsets, const char *header, int32_t n_out, const int32_t *sel,
                                  int32_t flags, int64_t *counts, char **table_out, int64_t *table_len);
WSH_EXPORT int wsh_collapse_store(const char *locus_path, int32_t n, const uint8_t *seq, const int64_t *off, const int32_t *len,
                                  const uint8_t *reverse, int32_t n_units, const int32_t *n_alt, const char *alts,
                                  const int32_t *alt_off, const int32_t *offsets, const char *header, int32_t n_out, const int32_t *sel,
                                  int32_t flags, int64_t *counts, char **table_out, int64_t *table_len)
{
    // (no C++ exception may cross into ctypes: std::bad_alloc and friends become this function's error value)
    try { return wsh_collapse_store_impl(locus_path, n, seq, off, len, reverse, n_units, n_alt, alts, alt_off, offsets, header, n_out, sel, flags, counts, table_out, table_len); } catch (...) { return -2; }
}
static int wsh_collapse_store_impl(const char *locus_path, int32_t n, const uint8_t *seq, const int64_t *off, const int32_t *len,
                                  const uint8_t *reverse, int32_t n_units, const int32_t *n_alt, const char *alts,
                                  const int32_t *alt_off, const int32_t *offsets, const char *header, int32_t n_out, const int32_t *sel,
                                  int32_t flags, int64_t *counts, char **table_out, int64_t *table_len)
{
    int total = 0;
    for (int u = 0; u < n_units; u++) total += n_alt[u];
    for (int a = 0; a < total; a++)
        if (alt_off[a + 1] == alt_off[a]) return 1;
    std::string out;
    out.reserve(size_t(n) * (8 + 4 * size_t(total)) + strlen(header) + 8);
    out.push_back(',');
    out.append(header);
    out.push_back('\n');
    char buf[32];
    std::vector<int64_t> gen;
    for (int i = 0; i < n; i++) {
        const char *s = reinterpret_cast<const char *>(seq) + off[i];
        int64_t rem = len[i] > 0 ? len[i] : 0;
        int64_t *cnt = counts + size_t(i) * total;
        std::fill(cnt, cnt + total, 0);
        int a0 = 0;
        for (int u = 0; u < n_units; u++) {
            const int64_t skip = std::min<int64_t>(offsets[u], rem);   // slide = slide[off:]
            s += skip;
            rem -= skip;
            while (rem > 0) {
                int64_t adv = -1;
                for (int k = 0; k < n_alt[u]; k++) {
                    const int64_t al = alt_off[a0 + k + 1] - alt_off[a0 + k];
                    if (al <= rem && memcmp(alts + alt_off[a0 + k], s, size_t(al)) == 0) {
                        cnt[a0 + k]++;
                        adv = al;   // the LAST matching alternative advances
                    }
                }
                if (adv < 0) break;
                s += adv;
                rem -= adv;
            }
            a0 += n_alt[u];
        }
        out.append(buf, size_t(format_int(i, buf)));
        // the columns store_collapsed makes, in its order: per unit `main` (all its counts) + one per further alternative, or the count
        gen.clear();
        a0 = 0;
        for (int u = 0; u < n_units; u++) {
            if (n_alt[u] > 1) {
                int64_t sum = 0;
                for (int k = 0; k < n_alt[u]; k++) sum += cnt[a0 + k];
                gen.push_back(sum);
                for (int k = 1; k < n_alt[u]; k++) gen.push_back(cnt[a0 + k]);
            } else {
                gen.push_back(cnt[a0]);
            }
            a0 += n_alt[u];
        }
        for (int c = 0; c < n_out; c++) {   // (two units of the same name share a column: the later one's values, `sel` says which)
            out.push_back(',');
            out.append(buf, size_t(format_int(gen[size_t(sel[c])], buf)));
        }
        out.append(reverse[i] ? ",True\n" : ",False\n");
    }
    if (flags & 1) {
        const std::string dir = std::string(locus_path) + "/predictions/complexSTR_analysis";
        if (!make_dirs(dir) || !write_file(dir + "/complex_repeat_units.csv", out)) return -2;
    }
    if (table_out) {
        *table_out = static_cast<char *>(malloc(out.size() + 1));
        memcpy(*table_out, out.data(), out.size());
        (*table_out)[out.size()] = 0;
        *table_len = int64_t(out.size());
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// One VBZ-filtered HDF5 chunk of 16-bit samples -> the samples, in one call without the GIL (warpstr_amd/fast5.py reads the chunk's
// bytes with H5Dread_chunk; the reader processes of a run decode straight into the staging buffer the GPU upload starts from).
// Layout (the published VBZ version 0): u32 uncompressed byte count, then -- zstd_level != 0 -- a zstd frame holding a StreamVByte
// block: ceil(n/4) key bytes (2 bits per value = byte length - 1, first value in the low bits), then the little-endian value
// bytes; the values are deltas, zig-zag mapped if `zigzag`.  zstd comes as two function pointers (ZSTD_getFrameContentSize,
// ZSTD_decompress of whatever libzstd the process loaded): this library links against nothing.
// Returns the number of samples written (the chunk's, or its first `cap`), or -1 chunk too short, -2 not a sized zstd frame,
// -3 zstd failed, -4 StreamVByte block shorter than its key area, -5 shorter than its keys say.
typedef unsigned long long (*zstd_size_fn)(const void *, size_t);
typedef size_t (*zstd_decompress_fn)(void *, size_t, const void *, size_t);
typedef size_t (*zstd_decompress_dctx_fn)(void *, void *, size_t, const void *, size_t);

// (a decompression context the process keeps -- wsh_vbz_context -- saves ZSTD_decompress the context it otherwise makes and
// drops per call: 9 % of a 115 KB frame's time)
struct VbzContext {
    void *dctx = nullptr;
    zstd_decompress_dctx_fn fn = nullptr;
};
static thread_local VbzContext g_vbz_ctx;

static inline size_t zstd_run(void *decompress_fn, void *dst, size_t cap, const void *src, size_t n)
{
    if (g_vbz_ctx.dctx && g_vbz_ctx.fn) return g_vbz_ctx.fn(g_vbz_ctx.dctx, dst, cap, src, n);
    return reinterpret_cast<zstd_decompress_fn>(decompress_fn)(dst, cap, src, n);
}

// dctx: a ZSTD_DCtx the caller made (ZSTD_createDCtx) and keeps alive, decompress_dctx_fn: ZSTD_decompressDCtx -- used by this
// THREAD's wsh_vbz_decode_i16 / wsh_vbz_unpack from now on instead of the plain ZSTD_decompress they are handed (null: back to that)
WSH_EXPORT void wsh_vbz_context(void *dctx, void *decompress_dctx_fn)
{
    g_vbz_ctx.dctx = dctx;
    g_vbz_ctx.fn = reinterpret_cast<zstd_decompress_dctx_fn>(decompress_dctx_fn);
}

static int64_t wsh_vbz_decode_i16_impl(const uint8_t *chunk, int64_t n_chunk, int32_t zigzag, int32_t zstd_level, void *size_fn, void *decompress_fn,
                                      int16_t *out, int64_t cap);
WSH_EXPORT int64_t wsh_vbz_decode_i16(const uint8_t *chunk, int64_t n_chunk, int32_t zigzag, int32_t zstd_level, void *size_fn, void *decompress_fn,
                                      int16_t *out, int64_t cap)
{
    // (no C++ exception may cross into ctypes: std::bad_alloc and friends become this function's error value)
    try { return wsh_vbz_decode_i16_impl(chunk, n_chunk, zigzag, zstd_level, size_fn, decompress_fn, out, cap); } catch (...) { return -3; }
}
static int64_t wsh_vbz_decode_i16_impl(const uint8_t *chunk, int64_t n_chunk, int32_t zigzag, int32_t zstd_level, void *size_fn, void *decompress_fn,
                                      int16_t *out, int64_t cap)
{
    if (n_chunk < 4) return -1;
    uint32_t n_bytes;
    memcpy(&n_bytes, chunk, 4);
    const int64_t coded = n_bytes / 2;   // values in the block: its key area is ceil(coded / 4) bytes
    // (HDF5 hands a filter the whole chunk: the last chunk of a dataset codes chunk-length samples of which the dataset holds
    // fewer -- `cap` of them are wanted, the first)
    const int64_t n = coded < cap ? coded : cap;
    const uint8_t *svb = chunk + 4;
    int64_t svb_bytes = n_chunk - 4;
    static thread_local std::vector<uint8_t> scratch;
    if (zstd_level != 0) {
        const unsigned long long size = reinterpret_cast<zstd_size_fn>(size_fn)(svb, size_t(svb_bytes));
        // (ceil(coded / 4) key bytes + at most 4 bytes a value: a frame that declares more is corrupt, and its size must not size
        // the scratch buffer)
        if (size >= (1ull << 62) || size > 5ull * uint64_t(coded) + 64) return -2;
        if (scratch.size() < size + 8) scratch.resize(size + 8);
        const size_t got = zstd_run(decompress_fn, scratch.data(), size, svb, size_t(svb_bytes));
        if (got != size) return -3;
        svb = scratch.data();
        svb_bytes = int64_t(size);
    }
    const int64_t n_keys = (coded + 3) / 4;
    if (svb_bytes < n_keys) return -4;
    const uint8_t *data = svb + n_keys, *end = svb + svb_bytes;
    uint32_t acc = 0;   // (the running sum wraps to 16 bits in the end: 32 are enough)
    auto put = [&](int64_t i, uint32_t v) {
        acc += zigzag ? (v >> 1) ^ (0u - (v & 1u)) : v;
        out[i] = int16_t(uint16_t(acc));
    };
    int64_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const uint8_t key = svb[i >> 2];
        if (key == 0 && data + 4 <= end) {   // four one-byte values: almost every group of a nanopore signal
            put(i, data[0]); put(i + 1, data[1]); put(i + 2, data[2]); put(i + 3, data[3]);
            data += 4;
            continue;
        }
        for (int q = 0; q < 4; q++) {
            const int len = ((key >> (2 * q)) & 3) + 1;
            if (data + len > end) return -5;
            uint32_t v = 0;
            for (int k = 0; k < len; k++) v |= uint32_t(data[k]) << (8 * k);
            data += len;
            put(i + q, v);
        }
    }
    for (; i < n; i++) {
        const int len = ((svb[i >> 2] >> ((i & 3) * 2)) & 3) + 1;
        if (data + len > end) return -5;
        uint32_t v = 0;
        for (int k = 0; k < len; k++) v |= uint32_t(data[k]) << (8 * k);
        data += len;
        put(i, v);
    }
    return n;
}

// The same chunk taken only as far as a GPU cannot take it (wsx_vbz_decode does the rest): the zstd frame's content -- the
// StreamVByte block -- written to `out` (a reader's page-locked arena), checked: the keys of its n values must ask for exactly
// the bytes that follow them (so that the device kernel never meets a block it has to flag).  Returns the block's size in bytes,
// or -1 .. -5 as above, -6 the block does not fit `cap` bytes; *n_out = the samples the chunk says it holds.  (Bytes behind the
// last value are not part of the block, as for the decoder above.)  (zstd_level == 0: the chunk's bytes are the block; copied.)
WSH_EXPORT int64_t wsh_vbz_unpack(const uint8_t *chunk, int64_t n_chunk, int32_t zstd_level, void *size_fn, void *decompress_fn, uint8_t *out,
                                  int64_t cap, int64_t *n_out)
{
    if (n_chunk < 4) return -1;
    uint32_t n_bytes;
    memcpy(&n_bytes, chunk, 4);
    const int64_t n = n_bytes / 2;
    *n_out = n;
    const uint8_t *body = chunk + 4;
    const int64_t body_bytes = n_chunk - 4;
    int64_t size = body_bytes;
    if (zstd_level != 0) {
        const unsigned long long zs = reinterpret_cast<zstd_size_fn>(size_fn)(body, size_t(body_bytes));
        if (zs >= (1ull << 62)) return -2;
        if (int64_t(zs) > cap) return -6;
        if (zstd_run(decompress_fn, out, size_t(zs), body, size_t(body_bytes)) != zs) return -3;
        size = int64_t(zs);
    } else {
        if (size > cap) return -6;
        memcpy(out, body, size_t(size));
    }
    const int64_t n_keys = (n + 3) / 4;
    if (size < n_keys) return -4;
    // bytes the keys ask for: n + the sum of their two-bit fields (eight key bytes at a time: fields -> nibble sums -> byte sums)
    uint64_t extra = 0;
    const int64_t full = n / 4;   // key bytes whose four values all exist
    int64_t k = 0;
    for (; k + 8 <= full; k += 8) {
        uint64_t x;
        memcpy(&x, out + k, 8);
        uint64_t t = (x & 0x3333333333333333ull) + ((x >> 2) & 0x3333333333333333ull);
        t = (t + (t >> 4)) & 0x0f0f0f0f0f0f0f0full;
        extra += (t * 0x0101010101010101ull) >> 56;
    }
    for (; k < full; k++) {
        const uint8_t key = out[k];
        extra += (key & 3) + ((key >> 2) & 3) + ((key >> 4) & 3) + (key >> 6);
    }
    for (int64_t i = full * 4; i < n; i++) extra += (out[i >> 2] >> ((i & 3) * 2)) & 3;
    const int64_t need = n_keys + n + int64_t(extra);
    if (need > size) return -5;
    return need;
}

// n pieces of host memory laid end to end into dst (the raw reads of a batch into the page-locked staging buffer the upload
// starts from), split by bytes over up to `threads` threads: 275 MB in 50 000 pieces is 30 ms of one core's memcpy.
static void wsh_gather_impl(const void *const *src, const int64_t *bytes, int64_t n, void *dst, int32_t threads);
WSH_EXPORT void wsh_gather(const void *const *src, const int64_t *bytes, int64_t n, void *dst, int32_t threads)
{
    // (no C++ exception may cross into ctypes: std::bad_alloc and friends become this function's error value)
    try { wsh_gather_impl(src, bytes, n, dst, threads); } catch (...) { int64_t at = 0; for (int64_t i = 0; i < n; i++) if (bytes[i] > 0) { memcpy(static_cast<char *>(dst) + at, src[i], size_t(bytes[i])); at += bytes[i]; } }
}
static void wsh_gather_impl(const void *const *src, const int64_t *bytes, int64_t n, void *dst, int32_t threads)
{
    std::vector<int64_t> at((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; i++) at[i + 1] = at[i] + (bytes[i] > 0 ? bytes[i] : 0);
    const int64_t total = at[n];
    auto run = [&](int64_t lo, int64_t hi) {   // pieces [lo, hi)
        for (int64_t i = lo; i < hi; i++)
            if (bytes[i] > 0) memcpy(static_cast<char *>(dst) + at[i], src[i], size_t(bytes[i]));
    };
    const int nt = int(std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(threads, 16), total >> 22)));   // >= 4 MB per thread
    if (nt <= 1) { run(0, n); return; }
    std::vector<std::thread> th;
    int64_t lo = 0;
    for (int t = 0; t < nt; t++) {
        const int64_t want = total * (t + 1) / nt;
        int64_t hi = t == nt - 1 ? n : int64_t(std::upper_bound(at.begin(), at.end(), want) - at.begin()) - 1;
        hi = std::max(hi, lo);
        if (t == nt - 1) run(lo, hi);
        else th.emplace_back(run, lo, hi);
        lo = hi;
    }
    for (auto &x : th) x.join();
}

}  // extern "C"
