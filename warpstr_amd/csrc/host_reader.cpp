// host_reader.cpp -- the reads of a chunk from their .fast5 files into a reader's arena, in ONE call without the GIL (part of
// warpstr_amd/_host_loci.so; warpstr_amd/_readers.py: pack_arena).
//
// Upstream opens one fast5 per read with h5py and lets the HDF5 filter plugin 32020 decode the `Signal` dataset
// (Fast5.get_data_processed, src/schemas/fast5.py:45-57; get_workload, src/caller/wrapper.py:44-54).  Here a reader process hands
// the device the StreamVByte block inside each chunk's zstd frame (wsx_vbz_decode does the rest), and what it spends per read
// decides how many reads per second a box's CPU share can feed the GPU: through ctypes one read cost 0.047 ms in libhdf5 (an
// H5Dopen2 by name, dataspace / filter / chunk queries, H5Dread_chunk's own copy), ~0.03 ms of interpreter between those calls
// and 0.066 ms in zstd.  This file keeps what libhdf5 has to do (open the dataset, say where its chunks lie) and does the rest
// itself: the chunk's bytes come straight from the file (pread at the address H5Dget_chunk_info_by_coord names, checked once per
// file against H5Dread_chunk), the frame is decompressed straight to its place in the arena, no interpreter in between.
//
// libhdf5 and libzstd are the ones the process has loaded already (paths from warpstr_amd/_h5core.lib_paths), bound with
// dlopen / dlsym: this library links against neither.  Whatever this file is not sure about -- a layout, filter or integer size
// other than chunked VBZ version 0 on int16 -- it leaves to the Python reader (status 1 for that read).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#define WSH_EXPORT extern "C" __attribute__((visibility("default")))

extern "C" int64_t wsh_vbz_unpack(const uint8_t *chunk, int64_t n_chunk, int32_t zstd_level, void *size_fn, void *decompress_fn, uint8_t *out,
                                  int64_t cap, int64_t *n_out);

namespace {

typedef int64_t hid_t;
typedef unsigned long long hsize_t;
typedef uint64_t haddr_t;
constexpr int VBZ_FILTER = 32020;
constexpr haddr_t HADDR_UNDEF = ~haddr_t(0);

struct Api {
    bool ready = false;
    hid_t (*H5Fopen)(const char *, unsigned, hid_t) = nullptr;
    int (*H5Fclose)(hid_t) = nullptr;
    hid_t (*H5Dopen2)(hid_t, const char *, hid_t) = nullptr;
    int (*H5Dclose)(hid_t) = nullptr;
    hid_t (*H5Dget_space)(hid_t) = nullptr;
    int (*H5Sclose)(hid_t) = nullptr;
    long long (*H5Sget_simple_extent_npoints)(hid_t) = nullptr;
    hid_t (*H5Dget_create_plist)(hid_t) = nullptr;
    int (*H5Pclose)(hid_t) = nullptr;
    int (*H5Pget_nfilters)(hid_t) = nullptr;
    int (*H5Pget_filter2)(hid_t, unsigned, unsigned *, size_t *, unsigned *, size_t, char *, unsigned *) = nullptr;
    int (*H5Pget_layout)(hid_t) = nullptr;
    int (*H5Pget_chunk)(hid_t, int, hsize_t *) = nullptr;
    int (*H5Dget_chunk_info_by_coord)(hid_t, const hsize_t *, unsigned *, haddr_t *, hsize_t *) = nullptr;   // (HDF5 >= 1.10.5)
    int (*H5Dget_chunk_storage_size)(hid_t, const hsize_t *, hsize_t *) = nullptr;
    int (*H5Dread_chunk)(hid_t, hid_t, const hsize_t *, uint32_t *, void *) = nullptr;
    hid_t (*H5Gopen2)(hid_t, const char *, hid_t) = nullptr;
    int (*H5Gclose)(hid_t) = nullptr;
    int (*H5Gget_num_objs)(hid_t, hsize_t *) = nullptr;
    long (*H5Gget_objname_by_idx)(hid_t, hsize_t, char *, size_t) = nullptr;
    int (*H5Lexists)(hid_t, const char *, hid_t) = nullptr;
    int (*H5Eset_auto2)(hid_t, void *, void *) = nullptr;
    int (*H5open)(void) = nullptr;
    void *zstd_size = nullptr, *zstd_decompress = nullptr;
} g;

struct File {
    hid_t fid = -1;
    int fd = -1;
    int pread_ok = -1;   // -1: not compared with H5Dread_chunk yet; 1: the chunk addresses are file offsets; 0: they are not
    uint64_t used = 0;
};
// (per thread: a reader process has one thread; an in-process reader thread of main_wrapper_loci its own files)
struct Files : std::unordered_map<std::string, File> {
    ~Files();   // (a thread that ends closes what it opened)
};
thread_local Files t_files;
thread_local uint64_t t_tick = 0;
thread_local std::vector<uint8_t> t_chunk, t_check;
constexpr size_t MAX_OPEN = 64;

void close_file(File &f)
{
    if (f.fid >= 0) g.H5Fclose(f.fid);
    if (f.fd >= 0) close(f.fd);
    f.fid = -1;
    f.fd = -1;
}

Files::~Files()
{
    if (g.ready)
        for (auto &kv : *this) close_file(kv.second);
}

File *open_file(const std::string &path)
{
    auto it = t_files.find(path);
    if (it != t_files.end()) {
        it->second.used = ++t_tick;
        return &it->second;
    }
    if (t_files.size() >= MAX_OPEN) {   // (the least recently used file goes)
        auto old = t_files.begin();
        for (auto q = t_files.begin(); q != t_files.end(); ++q)
            if (q->second.used < old->second.used) old = q;
        close_file(old->second);
        t_files.erase(old);
    }
    File f;
    f.fid = g.H5Fopen(path.c_str(), 0 /* H5F_ACC_RDONLY */, 0);
    if (f.fid < 0) return nullptr;
    f.fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
    f.used = ++t_tick;
    return &(t_files[path] = f);
}

bool exists(hid_t fid, const char *a, const char *b = nullptr, const char *c = nullptr)
{
    std::string cur;
    for (const char *part : {a, b, c}) {
        if (!part) break;
        cur = cur.empty() ? part : cur + "/" + part;
        if (g.H5Lexists(fid, cur.c_str(), 0) <= 0) return false;
    }
    return true;
}

// the dataset of a read: `read_<id>/Raw/Signal` of a multi-read file, else `Raw/Reads/<first>/Signal` of a single-read file (as
// upstream takes the first read, src/schemas/fast5.py:50-52), else the one read of a multi-read file when no id was given
hid_t open_signal(hid_t fid, const char *read_id)
{
    char name[768];
    if (read_id && *read_id) {
        snprintf(name, sizeof name, "read_%s/Raw/Signal", read_id);
        const hid_t d = g.H5Dopen2(fid, name, 0);
        if (d >= 0) return d;
    }
    if (exists(fid, "Raw", "Reads")) {
        const hid_t grp = g.H5Gopen2(fid, "Raw/Reads", 0);
        if (grp >= 0) {
            hsize_t n = 0;
            char first[512];
            const bool ok = g.H5Gget_num_objs(grp, &n) >= 0 && n > 0 && g.H5Gget_objname_by_idx(grp, 0, first, sizeof first) > 0;
            g.H5Gclose(grp);
            if (ok) {
                snprintf(name, sizeof name, "Raw/Reads/%s/Signal", first);
                return g.H5Dopen2(fid, name, 0);
            }
        }
    }
    return -1;   // (a multi-read file asked without an id: the Python reader walks its root group)
}

// Can the device decode this frame (csrc/wsx_zstd.hip)?  Read from its headers alone: no dictionary, a declared content size, at
// most 32 blocks, no block of a reserved type, no "treeless" literals section before a block has brought a Huffman tree.
// *content = the content size it declares.
bool frame_for_device(const uint8_t *b, int64_t n, int64_t *content)
{
    if (n < 6 || b[0] != 0x28 || b[1] != 0xB5 || b[2] != 0x2F || b[3] != 0xFD) return false;
    const int fhd = b[4], flag = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
    if (did || (fhd & 8)) return false;
    int64_t pos = 5 + (single ? 0 : 1);
    const int fcs = flag == 0 ? (single ? 1 : 0) : flag == 1 ? 2 : flag == 2 ? 4 : 8;
    if (fcs == 0 || pos + fcs > n) return false;
    uint64_t v = 0;
    for (int i = 0; i < fcs; i++) v |= uint64_t(b[pos + i]) << (8 * i);
    if (fcs == 2) v += 256;
    pos += fcs;
    if (v > (uint64_t(32) << 17)) return false;
    *content = int64_t(v);
    bool tree = false;
    for (int blocks = 0;; blocks++) {
        if (blocks >= 32 || pos + 3 > n) return false;
        const uint32_t bh = b[pos] | (b[pos + 1] << 8) | (uint32_t(b[pos + 2]) << 16);
        pos += 3;
        const int last = bh & 1, type = (bh >> 1) & 3;
        const int64_t size = bh >> 3;
        if (type == 3 || pos + (type == 1 ? 1 : size) > n) return false;
        if (type == 2) {
            if (size < 1 || ((b[pos] & 3) == 3 && !tree)) return false;
            tree = tree || (b[pos] & 3) == 2;
        }
        pos += type == 1 ? 1 : size;
        if (last) return true;
    }
}

}  // namespace

// Binds libhdf5 and libzstd (the paths the process loaded them from).  0: ready; 1: a symbol is missing (an HDF5 older than
// 1.10.5 has no H5Dget_chunk_info_by_coord: the Python reader does the work); -1: a library cannot be opened.
WSH_EXPORT int wsh_reader_init(const char *libhdf5, const char *libzstd)
{
    if (g.ready) return 0;
    void *h = dlopen(libhdf5, RTLD_NOW | RTLD_GLOBAL), *z = dlopen(libzstd, RTLD_NOW | RTLD_GLOBAL);
    if (!h || !z) return -1;
    bool ok = true;
    auto sym = [&](void *lib, const char *name) { void *p = dlsym(lib, name); ok = ok && p != nullptr; return p; };
#define BIND(name) g.name = reinterpret_cast<decltype(g.name)>(sym(h, #name))
    BIND(H5Fopen); BIND(H5Fclose); BIND(H5Dopen2); BIND(H5Dclose); BIND(H5Dget_space); BIND(H5Sclose); BIND(H5Sget_simple_extent_npoints);
    BIND(H5Dget_create_plist); BIND(H5Pclose); BIND(H5Pget_nfilters); BIND(H5Pget_filter2); BIND(H5Pget_layout); BIND(H5Pget_chunk);
    BIND(H5Dget_chunk_info_by_coord); BIND(H5Dget_chunk_storage_size); BIND(H5Dread_chunk); BIND(H5Gopen2); BIND(H5Gclose);
    BIND(H5Gget_num_objs); BIND(H5Gget_objname_by_idx); BIND(H5Lexists); BIND(H5Eset_auto2); BIND(H5open);
#undef BIND
    g.zstd_size = sym(z, "ZSTD_getFrameContentSize");
    g.zstd_decompress = sym(z, "ZSTD_decompress");
    if (!ok) return 1;
    g.H5open();
    g.H5Eset_auto2(0, nullptr, nullptr);   // errors come back as return codes, the library prints nothing
    g.ready = true;
    return 0;
}

// Closes the files this THREAD has open (a run's end; the files of a reader process close with it).
WSH_EXPORT void wsh_reader_close(void)
{
    for (auto &kv : t_files) close_file(kv.second);
    t_files.clear();
}

// Reads [first, n) of a chunk, one after the other, into the arena [arena, arena + cap): every chunk of a read's dataset as the
// StreamVByte block inside its zstd frame (kind 1 zig-zag, 2 plain differences; wsx_vbz_block of include/warpstr_hip.h) or, where
// the filter was skipped when the chunk was written, as plain int16 samples (kind 0), each block at the next 16-byte boundary
// from *at on.  path[i]: the read's own (annotated, single-read) file; fallback[i] (or NULL): the multi-read file that holds it
// when that one does not exist (caller-only input); name[i]: the read's id in a multi-read file.
// Per read: lens[i] = its samples.  Per block, SEVEN int64 in `table` (at most table_cap blocks): read index, kind, first byte in
// the arena, bytes, samples wanted from it, values it codes, and -- device_zstd != 0: a chunk whose zstd frame the device can decode
// (frame_for_device) is left as it is, kind 3 / 4 = the frame of a kind 1 / 2 block -- the content size the frame declares (else 0).
// Returns the index of the first read NOT done: n when all are; i < n when read i does not fit (the arena's room: *need bytes
// from the arena's start, or the table's) -- the caller makes room and calls again with first = i -- or when status[i] != 0:
//   1  the Python reader must take this read (layout / filter / integer size this file does not do)
//  -1  the file cannot be opened, -2 no signal dataset, -3 a chunk is missing or cannot be read,
//  -4 .. -9  the VBZ errors -1 .. -6 of wsh_vbz_unpack (too short, no sized frame, zstd failed, block shorter than its keys ...),
// -10  the dataset's chunks do not add up to its length.
WSH_EXPORT int64_t wsh_reader_pack(int64_t first, int64_t n, const char *const *path, const char *const *fallback, const char *const *name,
                                   uint8_t *arena, int64_t cap, int64_t *at, int64_t *lens, int32_t *status, int64_t *table,
                                   int64_t table_cap, int64_t *n_blocks, int64_t *need, int32_t device_zstd)
{
    if (!g.ready) return -1;
    try {
        for (int64_t i = first; i < n; i++) {
            status[i] = 0;
            const char *p = path[i], *rid = nullptr;
            struct stat sb;
            if (fallback && fallback[i] && stat(p, &sb) != 0) {   // caller-only input: the read is still in its multi-read file
                p = fallback[i];
                rid = name[i];
            }
            File *f = open_file(p);
            if (!f) { status[i] = -1; return i; }
            const hid_t d = open_signal(f->fid, rid);
            if (d < 0) { status[i] = rid ? -2 : 1; return i; }
            struct Close { hid_t d; ~Close() { g.H5Dclose(d); } } closer{d};
            const hid_t sp = g.H5Dget_space(d);
            const long long ns = sp >= 0 ? g.H5Sget_simple_extent_npoints(sp) : -1;
            if (sp >= 0) g.H5Sclose(sp);
            const hid_t pl = g.H5Dget_create_plist(d);
            if (ns < 0 || pl < 0) { if (pl >= 0) g.H5Pclose(pl); status[i] = -2; return i; }
            unsigned cd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            bool vbz = false, other_filter = false;
            const int nf = g.H5Pget_nfilters(pl);
            for (int q = 0; q < nf; q++) {
                unsigned flags = 0, cfg = 0, v[8] = {0};
                size_t ne = 8;
                char fname[64];
                const int id = g.H5Pget_filter2(pl, unsigned(q), &flags, &ne, v, sizeof fname, fname, &cfg);
                if (id == VBZ_FILTER) { vbz = true; memcpy(cd, v, sizeof cd); }
                else other_filter = true;
            }
            hsize_t chunk_len = 0;
            const bool chunked = g.H5Pget_layout(pl) == 2 && g.H5Pget_chunk(pl, 1, &chunk_len) == 1 && chunk_len > 0;
            g.H5Pclose(pl);
            // (version 0, two-byte integers; anything else -- gzip, contiguous, 4-byte samples -- is the Python reader's)
            if (!vbz || other_filter || !chunked || cd[0] != 0 || cd[1] != 2) { status[i] = 1; return i; }
            const int kind = cd[2] ? 1 : 2, level = int(cd[3]);
            const int64_t n_chunks = (ns + int64_t(chunk_len) - 1) / int64_t(chunk_len);
            if (*n_blocks + n_chunks > table_cap) { *need = 0; return i; }
            const int64_t at0 = *at, blocks0 = *n_blocks;
            auto fail = [&](int32_t code) { status[i] = code; *at = at0; *n_blocks = blocks0; return i; };   // (nothing of the read stays)
            int64_t done = 0;
            for (int64_t c = 0; c < n_chunks; c++) {
                const hsize_t off = hsize_t(c) * chunk_len;
                const int64_t want = std::min<int64_t>(int64_t(chunk_len), ns - int64_t(off));
                unsigned mask = 0;
                haddr_t addr = HADDR_UNDEF;
                hsize_t size = 0;
                if (g.H5Dget_chunk_info_by_coord(d, &off, &mask, &addr, &size) < 0 || size == 0) return fail(-3);
                if (t_chunk.size() < size) t_chunk.resize(size + size / 2);
                bool have = false;
                if (f->fd >= 0 && addr != HADDR_UNDEF && f->pread_ok != 0) {
                    have = pread(f->fd, t_chunk.data(), size, off_t(addr)) == ssize_t(size);
                    if (have && f->pread_ok < 0) {   // once per file: do the addresses mean what this code takes them to mean?
                        if (t_check.size() < size) t_check.resize(size + size / 2);
                        uint32_t m2 = 0;
                        f->pread_ok = g.H5Dread_chunk(d, 0, &off, &m2, t_check.data()) >= 0 && memcmp(t_check.data(), t_chunk.data(), size) == 0;
                        have = f->pread_ok == 1;
                    }
                }
                if (!have) {
                    uint32_t m2 = 0;
                    if (g.H5Dread_chunk(d, 0, &off, &m2, t_chunk.data()) < 0) return fail(-3);
                    mask = m2;
                }
                const int64_t pos = (*at + 15) & ~int64_t(15);
                int64_t *row = table + 7 * *n_blocks;
                row[6] = 0;
                if (mask & 1u) {   // the filter was skipped when this chunk was written: plain samples
                    const int64_t got = std::min<int64_t>(int64_t(size) / 2, want);
                    if (pos + 2 * got > cap) { *at = at0; *n_blocks = blocks0; *need = pos + 2 * got; return i; }
                    memcpy(arena + pos, t_chunk.data(), size_t(2 * got));
                    row[0] = i; row[1] = 0; row[2] = pos; row[3] = 2 * got; row[4] = got; row[5] = got;
                    *at = pos + 2 * got;
                    done += got;
                } else {
                    if (size < 4) return fail(-4);
                    int64_t content = 0;
                    if (device_zstd && level != 0 && frame_for_device(t_chunk.data() + 4, int64_t(size) - 4, &content) &&
                        content <= 5 * std::max<int64_t>(int64_t(chunk_len), want) + 64) {
                        // the frame as it lies in the file: wsx_zstd_decode undoes it (kinds 3 / 4 = a zstd frame around kinds 1 / 2)
                        const int64_t fb = int64_t(size) - 4;
                        if (pos + fb > cap) { *at = at0; *n_blocks = blocks0; *need = pos + fb; return i; }
                        memcpy(arena + pos, t_chunk.data() + 4, size_t(fb));
                        uint32_t nbytes;
                        memcpy(&nbytes, t_chunk.data(), 4);
                        const int64_t coded = nbytes / 2, got = std::min<int64_t>(want, coded);
                        row[0] = i; row[1] = kind + 2; row[2] = pos; row[3] = fb; row[4] = got; row[5] = coded; row[6] = content;
                        *at = pos + fb;
                        done += got;
                        ++*n_blocks;
                        continue;
                    }
                    int64_t room = int64_t(size) - 4;
                    if (level != 0) {
                        typedef unsigned long long (*size_fn)(const void *, size_t);
                        const unsigned long long zs = reinterpret_cast<size_fn>(g.zstd_size)(t_chunk.data() + 4, size_t(size) - 4);
                        // (ceil(n / 4) key bytes + at most 4 bytes a value: a frame that declares more is corrupt)
                        if (zs >= (1ull << 62) || zs > 5ull * uint64_t(std::max<int64_t>(int64_t(chunk_len), want)) + 64) return fail(-5);
                        room = int64_t(zs);
                    }
                    if (pos + room > cap) { *at = at0; *n_blocks = blocks0; *need = pos + room; return i; }
                    int64_t coded = 0;
                    const int64_t nb = wsh_vbz_unpack(t_chunk.data(), int64_t(size), level, g.zstd_size, g.zstd_decompress, arena + pos, room, &coded);
                    if (nb < 0) return fail(int32_t(nb - 3));
                    const int64_t got = std::min<int64_t>(want, coded);
                    row[0] = i; row[1] = kind; row[2] = pos; row[3] = nb; row[4] = got; row[5] = coded;
                    *at = pos + nb;
                    done += got;
                }
                ++*n_blocks;
            }
            if (done != ns) return fail(-10);
            lens[i] = ns;
        }
        return n;
    } catch (...) {   // (std::bad_alloc: no C++ exception may cross into ctypes)
        return -1;
    }
}
