/*
 * flank_oracle.c -- CPU restatement of WarpSTR's flank localisation (TEST INFRASTRUCTURE ONLY; only tests/ may load it).
 *
 * What it follows (paths relative to the upstream repository):
 *   find_sequence          src/extractor/tr_extractor.py:196-250   local alignment of a flank (pattern) in the
 *                          basecalled read (text) + the position/score/identity arithmetic on the aligned strings
 *   transform_moves        tr_extractor.py:147-163                  Guppy move table -> context index per block
 *   extract_from_moves     tr_extractor.py:166-193                  context positions -> raw-signal positions
 *
 * PARITY UNPINNED.  find_sequence delegates the alignment to Biopython (Pipfile: biopython ==1.75,
 * Bio.pairwise2.align.localms(text, pattern, match, mismatch, open, extend, one_alignment_only=True)), which is not
 * installed here, and the upstream test data carries no basecalls or move tables (test_input/.../batch_0.fast5 holds
 * Analyses/Basecall_1D_000/Summary only), so neither golden vectors nor a live cross-check exist for this path.  The
 * published algorithm is restated: Smith-Waterman with the affine-gap recurrence of pairwise2; with upstream's
 * parameters (open = extend = -3, alignment_config, src/config.py:135-141) the recurrence is the linear-gap one below.
 * Where several optimal alignments exist pairwise2 returns the first its stack-based traceback produces; THIS
 * restatement fixes its own, documented rule instead (and the HIP kernels use the same rule):
 *   - end cell: the best-scoring cell with the largest text index, then the largest pattern index;
 *   - traceback: diagonal first, then "text base against a gap" (up), then "pattern base against a gap" (left);
 *   - the traceback runs on the DP restricted to the last WR = p + match*p/(-gap) + 2 text rows before the end cell
 *     (an alignment with a positive score cannot span more), rows above that window count as score 0.
 * Flanks of 100+ bases have a unique optimal placement in practice, where every rule gives the same answer.
 *
 * H[i][j] = max(0, H[i-1][j-1] + s(text_i, pat_j), H[i-1][j] + gap, H[i][j-1] + gap),  i = 1..n, j = 1..p.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t match, mismatch, gap_open, gap_extend;
} flo_scores;

typedef struct {
    int32_t status;     /* 0 found; 1 no positive-scoring alignment (pairwise2 returns []: upstream IndexError);
                           2 unsupported scores */
    int32_t score;      /* pairwise2 score + find_sequence's correction (tr_extractor.py:243-245) */
    int32_t start, end; /* Position(real_start, end) before origin_offset (text coordinates) */
    int32_t matches;    /* identity = matches / span (tr_extractor.py:234-246) */
    int32_t span;       /* len(ref) */
    int32_t row0, col0; /* cell where the local alignment starts: text / pattern bases consumed before it */
    int32_t row1, col1; /* cell where it ends */
    int32_t gaps_text;  /* '-' in the aligned text inside the local region (nums_gaps) */
    int32_t gaps_pattern; /* '-' in the aligned pattern inside the local region (nums_gaps2) */
    int32_t raw_score;  /* the alignment's own score */
    int32_t n_ops;      /* operations of the local region */
    int32_t n_best_cells; /* cells of the full matrix that reach the best score */
    int32_t tie_steps;    /* traceback steps with more than one predecessor reproducing the score */
} flo_hit;

int32_t flo_window_rows(int32_t p, const flo_scores *sc) { return p + (sc->match * p) / (-sc->gap_open) + 2; }

/* ops: forward order after return -- 'M' diagonal, 'U' text base against a gap, 'L' pattern base against a gap;
 * at most ops_cap are stored (n_ops reports the true count; callers pass ops_cap >= 2p + 2) */
int flo_find_sequence(const uint8_t *text, int32_t n, const uint8_t *pat, int32_t p, const flo_scores *sc, flo_hit *hit,
                      uint8_t *ops, int32_t ops_cap)
{
    memset(hit, 0, sizeof(*hit));
    hit->start = hit->end = -1;
    if (sc->gap_open != sc->gap_extend || sc->gap_open >= 0 || sc->match <= 0) {
        hit->status = 2;
        return 2;
    }
    const int32_t g = sc->gap_open;
    /* ---- stage 1: best cell of the full matrix (two rolling rows) ---- */
    int32_t *prev = (int32_t *)calloc((size_t)p + 1, sizeof(int32_t)), *cur = (int32_t *)calloc((size_t)p + 1, sizeof(int32_t));
    int32_t best = 0, bi = 0, bj = 0, nbest = 0;
    for (int32_t i = 1; i <= n; i++) {
        cur[0] = 0;
        for (int32_t j = 1; j <= p; j++) {
            int32_t h = prev[j - 1] + (text[i - 1] == pat[j - 1] ? sc->match : sc->mismatch);
            if (prev[j] + g > h) h = prev[j] + g;
            if (cur[j - 1] + g > h) h = cur[j - 1] + g;
            if (h < 0) h = 0;
            cur[j] = h;
            if (h >= best && h > 0) { /* later cells win ties: largest i, then largest j */
                nbest = h > best ? 1 : nbest + 1;
                best = h;
                bi = i;
                bj = j;
            }
        }
        int32_t *t = prev;
        prev = cur;
        cur = t;
    }
    free(prev);
    free(cur);
    if (best <= 0) {
        hit->status = 1;
        return 1;
    }
    /* ---- stage 2: windowed DP with full matrix, traceback from (bi, bj) ---- */
    const int32_t wr = flo_window_rows(p, sc);
    const int32_t i0 = bi - wr > 0 ? bi - wr : 0; /* rows i0+1 .. bi are recomputed, row i0 counts as zeros */
    const int32_t rows = bi - i0;
    int32_t *H = (int32_t *)calloc((size_t)(rows + 1) * (p + 1), sizeof(int32_t));
#define HW(i, j) H[(size_t)((i) - i0) * (p + 1) + (j)]
    for (int32_t i = i0 + 1; i <= bi; i++)
        for (int32_t j = 1; j <= p; j++) {
            int32_t h = HW(i - 1, j - 1) + (text[i - 1] == pat[j - 1] ? sc->match : sc->mismatch);
            if (HW(i - 1, j) + g > h) h = HW(i - 1, j) + g;
            if (HW(i, j - 1) + g > h) h = HW(i, j - 1) + g;
            if (h < 0) h = 0;
            HW(i, j) = h;
        }
    int32_t i = bi, j = bj, nops = 0, g1 = 0, g2 = 0, ties = 0;
    while (i > i0 && j > 0 && HW(i, j) > 0) {
        const int32_t h = HW(i, j);
        uint8_t op;
        if ((h == HW(i - 1, j - 1) + (text[i - 1] == pat[j - 1] ? sc->match : sc->mismatch)) + (h == HW(i - 1, j) + g) +
                (h == HW(i, j - 1) + g) > 1)
            ties++;
        if (h == HW(i - 1, j - 1) + (text[i - 1] == pat[j - 1] ? sc->match : sc->mismatch)) {
            op = 'M';
            i--;
            j--;
        } else if (h == HW(i - 1, j) + g) {
            op = 'U';
            i--;
            g2++;
        } else {
            op = 'L';
            j--;
            g1++;
        }
        if (nops < ops_cap) ops[nops] = op;
        nops++;
    }
#undef HW
    free(H);
    for (int32_t a = 0, b = (nops < ops_cap ? nops : ops_cap) - 1; a < b; a++, b--) {
        const uint8_t t = ops[a];
        ops[a] = ops[b];
        ops[b] = t;
    }
    hit->row0 = i;
    hit->col0 = j;
    hit->row1 = bi;
    hit->col1 = bj;
    hit->gaps_text = g1;
    hit->gaps_pattern = g2;
    hit->raw_score = best;
    hit->n_ops = nops;
    hit->n_best_cells = nbest;
    hit->tie_steps = ties;
    /* ---- find_sequence's arithmetic on the aligned strings (tr_extractor.py:226-250) ----
     * aligned strings: lead = |row0 - col0| gap characters in front of the shorter unaligned prefix, then the prefixes,
     * the local region, the unaligned suffixes, and gap characters after the shorter suffix. */
    const int32_t lead_text = j > i ? j - i : 0; /* gaps in front of the aligned text */
    const int32_t lead_pat = i > j ? i - j : 0;  /* gaps in front of the aligned pattern */
    const int32_t begin = (i > j ? i : j);       /* index of the local region in the aligned strings */
    const int32_t real_start = lead_pat;         /* al2[:begin].count('-') */
    const int32_t end = real_start + p + g2 - g1;
    const int32_t suf_t = n - bi, suf_p = p - bj;
    const int32_t alen = begin + nops + (suf_t > suf_p ? suf_t : suf_p);
    const int32_t lo = real_start < alen ? real_start : alen, hi = end < alen ? (end > lo ? end : lo) : alen;
    int32_t matches = 0;
    /* walk aligned index t in [lo, hi) and fetch both characters; ti / pj = text / pattern bases of the local region
     * consumed before the element being visited */
    int32_t ti = 0, pj = 0;
    for (int32_t q = 0; q < lo - begin && q < nops; q++) {
        if (ops[q] != 'L') ti++;
        if (ops[q] != 'U') pj++;
    }
    for (int32_t t = lo; t < hi; t++) {
        int a = -1, b = -1; /* -1 = gap character */
        if (t < begin) {
            if (t >= lead_text) a = text[t - lead_text];
            if (t >= lead_pat) b = pat[t - lead_pat];
        } else if (t < begin + nops) {
            const uint8_t op = ops[t - begin];
            if (op != 'L') a = text[i + ti++];
            if (op != 'U') b = pat[j + pj++];
        } else {
            const int32_t u = t - begin - nops;
            if (u < suf_t) a = text[bi + u];
            if (u < suf_p) b = pat[bj + u];
        }
        if (a >= 0 && a == b) matches++;
    }
    hit->start = real_start;
    hit->end = end;
    hit->span = hi - lo;
    hit->matches = matches;
    /* diff = (len(seq2) - (len(query) - nums_gaps2)) * gap_extend */
    hit->score = best + (p - ((hi - lo) - g2)) * sc->gap_extend;
    return 0;
}

/* transform_moves (tr_extractor.py:147-163): moves_r[0] = 0, moves_r[k] = moves_r[k-1] + (moves[k] != 0).
 * extract_from_moves (166-193): start = strand_start + first k with moves_r[k] == pos_start, times block_stride;
 *                               end   = strand_start + last  k with moves_r[k] == pos_end,  times block_stride; -1 if none. */
void flo_extract_from_moves(const uint8_t *moves, int64_t n_moves, int32_t pos_start, int32_t pos_end, int64_t strand_start,
                            int32_t block_stride, int64_t *raw_start, int64_t *raw_end)
{
    int64_t first = -1, last = -1;
    int32_t ctx = 0;
    for (int64_t k = 0; k < n_moves; k++) {
        if (k > 0 && moves[k]) ctx++;
        if (ctx == pos_start && first < 0) first = k;
        if (ctx == pos_end) last = k;
    }
    *raw_start = first >= 0 ? strand_start + first * block_stride : -1;
    *raw_end = last >= 0 ? strand_start + last * block_stride : -1;
}
