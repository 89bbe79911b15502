// wsx_zstd.hip -- the Zstandard frames of VBZ chunks decoded on the GPU (the step in front of wsx_vbz.hip).
//
// Upstream reads `Raw/Signal` through h5py and the HDF5 filter plugin 32020 (Fast5.get_data_processed, src/schemas/fast5.py:50-52);
// the plugin hands every chunk to libzstd before it undoes StreamVByte.  Until round 6 that was the one part of a read's way from
// the file to the caller that stayed on the host: 0.06 ms of a reader process per read, more than libhdf5 and everything else a
// reader does together -- and the box's CPU share, not the GPU, decided how many reads per second a run could call from files.
// The format is RFC 8878.  What a VBZ chunk of nanopore samples holds is almost nothing but Huffman-coded literals (the value bytes
// of the StreamVByte block; the key bytes are the few matches): blocks of <= 128 KB, four interleaved-by-position streams each, a
// handful of sequences.  Hence the shape of the kernel:
//   * zstd_index_kernel: a thread per frame walks its block headers and lists its Huffman-coded blocks;
//   * zstd_order_kernel: the list again, the blocks with the most literals first (a launch is then as long as ONE full block's chain);
//   * zstd_literals_kernel: a wavefront takes four blocks of the list (16 lanes each): lane 0 of a group reads the tree description
//     (direct weights, or FSE-coded weights decoded with two states), the wave builds the 2^max_bits-cell decoding tables in LDS --
//     two symbols a cell where both codes fit --, lanes 0..3 of every group decode the block's four streams side by side (a stream is
//     a chain of dependent table look-ups: no parallelism inside it; the loop holds nothing but LDS, and away from a stream's ends
//     nothing but the chain: a refill that selects, two look-ups);
//   * zstd_sequences_kernel: a wavefront per frame walks the blocks in order: sequences section (predefined / RLE / FSE / repeat
//     tables, built by lane 0 in LDS), decoded by lane 0 a few hundred sequences at a time, executed by the whole wave (literal run,
//     match -- overlapping matches as a periodic copy), repeat offsets carried from block to block.
// Everything the format allows is decoded except dictionaries and frames of more than 32 blocks (treeless literals -- a block coded
// with the tree of an earlier block -- build their table from that block's description: blocks are decoded side by side and wait
// for nobody): the host never hands those over
// (warpstr_amd/_h5core.py / csrc/host_reader.cpp look at the headers and decompress such a frame themselves); a frame that turns out
// corrupt sets its status and leaves its output undefined.  Pinned against libzstd (tests/test_gpu_zstd.py) and oracle/zstd_oracle.c.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

#include "../../include/warpstr_hip.h"

int wsx_internal_device(wsx_caller *c);
hipStream_t wsx_internal_stream(wsx_caller *c);
void wsx_internal_set_error(const char *msg);
extern "C" int wsx_internal_on_exception(void);
hipError_t wsx_internal_vbz_slot(wsx_caller *c, size_t bytes, void **host, void **dev, hipEvent_t *last_use);
hipError_t wsx_internal_zstd_status(wsx_caller *c, size_t bytes, void **p);

namespace {

constexpr int Z_MAX_BLOCKS = 32;  // blocks per frame (4 MB of content)
constexpr int HUF_MAX_BITS = 11;
constexpr int SEQ_CHUNK = 256;    // sequences decoded (lane 0) before the wave executes them

enum { Z_OK = 0, Z_UNSUPPORTED = 1, Z_CORRUPT = 2 };

template <int N>
struct FseTableT {   // an FSE decoding table of up to N cells
    uint8_t symbol[N];
    uint8_t nbits[N];
    uint16_t base[N];
    int al;
    int valid;
};
typedef FseTableT<512> FseTable;      // the sequences' tables (accuracy <= 9)
typedef FseTableT<64> FseTableW;      // the table of a Huffman tree's weights (accuracy <= 6)

struct BlockRec {
    int type, src, size;   // block type, first byte behind its header (relative to the frame), Block_Size
    int lit_at, regen;     // where its literals lie in the frame's literal area, how many there are
    int seq_at;            // first byte of its sequences section (relative to the frame); compressed blocks only
    int tree;              // the literals section whose Huffman tree description its literals are coded with: its own, or -- "treeless"
                           // literals -- that of the latest block before it that brought one; -1: its literals are raw or RLE
};

__device__ __forceinline__ int hibit(uint32_t x) { return 31 - __builtin_clz(x); }   // x != 0

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

struct Fwd {
    const uint8_t *p;
    int len, bit;
    bool bad;
};
__device__ uint32_t fwd_bits(Fwd &s, int n)   // n <= 16
{
    uint32_t v = 0;
    const int byte = s.bit >> 3;
    for (int i = 0; i < 3; i++)
        if (byte + i < s.len) v |= (uint32_t)s.p[byte + i] << (8 * i);
    if (s.bit + n > 8 * s.len) s.bad = true;
    v = (v >> (s.bit & 7)) & ((1u << n) - 1u);
    s.bit += n;
    return v;
}

// n <= 32 bits below bit `off` of a backward stream; bits before the stream's start read as zero
__device__ uint64_t back_bits(const uint8_t *src, int n, int &off)
{
    off -= n;
    int at = off, take = n;
    if (at < 0) {
        take += at;
        at = 0;
    }
    uint64_t v = 0;
    if (take > 0) {
        const int b0 = at >> 3, nb = ((at + take + 7) >> 3) - b0;   // <= 5 bytes
        for (int i = 0; i < nb; i++) v |= (uint64_t)src[b0 + i] << (8 * i);
        v = (v >> (at & 7)) & ((1ull << take) - 1ull);
    }
    if (off < 0) v = -off >= 64 ? 0 : v << -off;
    return v;
}

// 4.1.1 / educational decoder: cells from normalised counts (lane 0; `next` = 256 uint16 of scratch)
template <int N>
__device__ int fse_build(FseTableT<N> &t, const int16_t *freq, int nsym, int al, uint16_t *next)
{
    if ((1 << al) > N || nsym > 256) return Z_CORRUPT;
    const int size = 1 << al;
    int high = size;
    t.al = al;
    for (int s = 0; s < nsym; s++)
        if (freq[s] == -1) {
            t.symbol[--high] = (uint8_t)s;
            next[s] = 1;
        }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        if (freq[s] <= 0) continue;
        next[s] = (uint16_t)freq[s];
        for (int i = 0; i < freq[s]; i++) {
            t.symbol[pos] = (uint8_t)s;
            do pos = (pos + step) & mask;
            while (pos >= high);
        }
    }
    if (pos != 0) return Z_CORRUPT;
    for (int i = 0; i < size; i++) {
        const uint16_t x = next[t.symbol[i]]++;
        const int nb = al - hibit(x);
        t.nbits[i] = (uint8_t)nb;
        t.base[i] = (uint16_t)(((uint32_t)x << nb) - size);
    }
    t.valid = 1;
    return Z_OK;
}

// the table description; *took = its bytes.  freq: 256 int16 of scratch
template <int N>
__device__ int fse_read(FseTableT<N> &t, const uint8_t *src, int len, int max_al, int max_sym, int16_t *freq, uint16_t *next, int *took)
{
    Fwd in{src, len, 0, false};
    const int al = 5 + (int)fwd_bits(in, 4);
    if (al > max_al) return Z_CORRUPT;
    int remaining = 1 << al, nsym = 0;
    while (remaining > 0 && nsym < 256) {
        const int bits = hibit((uint32_t)remaining + 1) + 1;
        uint32_t val = fwd_bits(in, bits);
        const uint32_t lower = (1u << (bits - 1)) - 1, thresh = (1u << bits) - 1 - (uint32_t)(remaining + 1);
        if ((val & lower) < thresh) {
            in.bit -= 1;
            val &= lower;
        } else if (val > lower) {
            val -= thresh;
        }
        const int p = (int)val - 1;
        remaining -= p < 0 ? -p : p;
        freq[nsym++] = (int16_t)p;
        if (p == 0) {
            uint32_t rep = fwd_bits(in, 2);
            for (;;) {
                for (uint32_t i = 0; i < rep && nsym < 256; i++) freq[nsym++] = 0;
                if (rep != 3) break;
                rep = fwd_bits(in, 2);
            }
        }
        if (in.bad) return Z_CORRUPT;
    }
    if (remaining != 0 || nsym > max_sym + 1 || in.bad) return Z_CORRUPT;
    *took = (in.bit + 7) >> 3;
    return fse_build(t, freq, nsym, al, next);
}

__constant__ int16_t LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ int16_t ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__constant__ int16_t OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
__constant__ uint32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512,
                                     1024, 2048, 4096, 8192, 16384, 32768, 65536};
__constant__ uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ uint32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
                                     34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
__constant__ uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                    0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};


// n bytes from s to d (not overlapping), by the 64 lanes of a wave: 16 bytes a lane and step where the two are aligned alike (the
// literals of a block without matches before it: the usual case), else a byte a lane.  Independent loads, four in flight per lane.
__device__ __forceinline__ void wave_copy(uint8_t *d, const uint8_t *s, long long n, int lane)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    long long at = 0;
    if ((((uintptr_t)d ^ (uintptr_t)s) & 15) == 0 && n >= 256) {
        const long long head = (long long)((16 - ((uintptr_t)d & 15)) & 15);
        if (lane < head) d[lane] = s[lane];
        const long long nvec = (n - head) >> 4;
        u32x4 *dv = (u32x4 *)(d + head);
        const u32x4 *sv = (const u32x4 *)(s + head);
        long long i = lane;
        for (; i + 192 < nvec; i += 256) {
            const u32x4 a = sv[i], b = sv[i + 64], c = sv[i + 128], e = sv[i + 192];
            dv[i] = a, dv[i + 64] = b, dv[i + 128] = c, dv[i + 192] = e;
        }
        for (; i < nvec; i += 64) dv[i] = sv[i];
        at = head + (nvec << 4);
    }
    else if (n >= 256) {
        // not aligned alike (a literal run behind a match whose length is no multiple of 16: most of a VBZ block's literals): whole
        // 16-byte stores to aligned places all the same, the loads from where the bytes lie (the memory system takes a dwordx4 load
        // at any address; a byte a lane was 0.3 ms of the sequence kernel's 0.6)
        typedef u32x4 u32x4_anywhere __attribute__((aligned(1)));
        const long long head = (long long)((16 - ((uintptr_t)d & 15)) & 15);
        if (lane < head) d[lane] = s[lane];
        const long long nvec = (n - head) >> 4;
        u32x4 *dv = (u32x4 *)(d + head);
        const u32x4_anywhere *sv = (const u32x4_anywhere *)(s + head);
        long long i = lane;
        for (; i + 192 < nvec; i += 256) {
            const u32x4 a = sv[i], b = sv[i + 64], c = sv[i + 128], e = sv[i + 192];
            dv[i] = a, dv[i + 64] = b, dv[i + 128] = c, dv[i + 192] = e;
        }
        for (; i < nvec; i += 64) dv[i] = sv[i];
        at = head + (nvec << 4);
    }
    long long k = at + lane;
    for (; k + 192 < n; k += 256) {
        const uint8_t a = s[k], b = s[k + 64], c = s[k + 128], e = s[k + 192];
        d[k] = a, d[k + 64] = b, d[k + 128] = c, d[k + 192] = e;
    }
    for (; k < n; k += 64) d[k] = s[k];
}

// The frame's header and the walk over its block headers (one lane).  on_block(index, record) is told of every block; returns the
// frame's status, *n_blocks = how many there were.  (A visitor, not an array: a per-thread array of records is scratch memory --
// 784 bytes a lane of zstd_index_kernel made the runtime set aside hundreds of megabytes for the queue.)
template <class Visit>
__device__ int walk_blocks(const uint8_t *f, int flen, long long cap, int *n_blocks, Visit &&on_block)
{
    int st = Z_OK, nb = 0, pos = 0;
    if (flen < 5 || f[0] != 0x28 || f[1] != 0xB5 || f[2] != 0x2F || f[3] != 0xFD) st = Z_CORRUPT;
    else {
        const int fhd = f[4], flag = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
        if (fhd & 8) st = Z_CORRUPT;
        else if (did) st = Z_UNSUPPORTED;
        pos = 5 + (single ? 0 : 1) + (flag == 0 ? (single ? 1 : 0) : flag == 1 ? 2 : flag == 2 ? 4 : 8);
    }
    int lit_at = 0, tree_at = -1;
    while (st == Z_OK) {
        if (pos + 3 > flen) { st = Z_CORRUPT; break; }
        const uint32_t bh = f[pos] | (f[pos + 1] << 8) | ((uint32_t)f[pos + 2] << 16);
        pos += 3;
        const int last = bh & 1, type = (bh >> 1) & 3, size = (int)(bh >> 3);
        if (nb >= Z_MAX_BLOCKS) { st = Z_UNSUPPORTED; break; }
        if (type == 3 || pos + (type == 1 ? 1 : size) > flen) { st = Z_CORRUPT; break; }
        BlockRec B;
        B.type = type, B.src = pos, B.size = size, B.lit_at = lit_at, B.regen = 0, B.seq_at = 0, B.tree = -1;
        if (type == 2) {   // the literals header says how many literals the block brings: their place in the literal area
            if (size < 1) { st = Z_CORRUPT; break; }
            const int b0 = f[pos], ltype = b0 & 3, sf = (b0 >> 2) & 3;
            int regen, hl, comp;
            if (ltype < 2) {
                hl = (sf == 0 || sf == 2) ? 1 : sf == 1 ? 2 : 3;
                if (size < hl) { st = Z_CORRUPT; break; }
                regen = hl == 1 ? b0 >> 3 : hl == 2 ? (b0 >> 4) + (f[pos + 1] << 4) : (b0 >> 4) + (f[pos + 1] << 4) + (f[pos + 2] << 12);
                comp = ltype == 0 ? regen : 1;
            } else {
                hl = sf < 2 ? 3 : sf + 2;
                if (size < hl) { st = Z_CORRUPT; break; }
                uint64_t v = 0;
                for (int i = 0; i < hl; i++) v |= (uint64_t)f[pos + i] << (8 * i);
                const int w = sf < 2 ? 10 : sf == 2 ? 14 : 18;
                regen = (int)((v >> 4) & ((1u << w) - 1));
                comp = (int)((v >> (4 + w)) & ((1u << w) - 1));
                if (ltype == 2) tree_at = pos;
                else if (tree_at < 0) { st = Z_CORRUPT; break; }   // treeless, and no block before it brought a tree
                B.tree = tree_at;
            }
            if (hl + comp > size || regen > (1 << 17) || lit_at + regen > cap) { st = Z_CORRUPT; break; }
            B.regen = regen;
            B.seq_at = pos + hl + comp;
            lit_at += regen;
        }
        on_block(nb++, B);
        pos += type == 1 ? 1 : size;
        if (last) break;
    }
    *n_blocks = nb;
    return st;
}

constexpr int Z_WIN = 256;     // bytes of a stream's compressed input a window holds; the ring of a stream is two windows
constexpr int Z_LOOKUPS = 56;  // look-ups a stream makes between two flushes (28 refills of two: at most 112 bytes and the 8 of the words around --
                               // less than half a window, which is what lets a window arrive a period after it was asked for)
constexpr int Z_OUT = 2 * Z_LOOKUPS + 2;   // ... and the room for their symbols (two a look-up, the last store may spill one byte)
constexpr int ZG = 4;          // Huffman blocks a wavefront decodes side by side (a group of 16 lanes each: 4 decode, 16 fetch and flush)

// A cell of the decoding table (one per max_bits-bit pattern): the symbol the pattern starts with -- and the next one where both
// codes fit the pattern (nanopore value bytes average ~5 bits a symbol: two fit most of the time, and a symbol is a chain of
// dependent look-ups, so two per look-up halves the chain).
//   bits 0-7 first symbol, 8-15 second symbol, 16-19 bits both take (or the first alone), 20-23 bits of the first, 24-25 count
__device__ __forceinline__ uint32_t cell2(uint32_t c1, uint32_t c2, int max_bits)
{
    const uint32_t l1 = c1 >> 8, l2 = c2 >> 8;
    if (l1 + l2 <= (uint32_t)max_bits) return (c1 & 255u) | ((c2 & 255u) << 8) | ((l1 + l2) << 16) | (l1 << 20) | (2u << 24);
    return (c1 & 255u) | (l1 << 16) | (l1 << 20) | (1u << 24);
}

constexpr int TREE_STAGE = 144;  // a literals header (<= 5 bytes) and a Huffman tree description (<= 129)
constexpr int SEQ_STAGE = 4096;  // bytes of a sequences section the sequence kernel reads from LDS
constexpr int Z_CLASSES = 9;   // a block by its literals, in steps of 16 K (0 .. 128 K)
__device__ __forceinline__ int size_class(int regen) { return regen >> 14 > Z_CLASSES - 1 ? Z_CLASSES - 1 : regen >> 14; }

// a Huffman block of a launch: what zstd_index_kernel leaves for zstd_literals_kernel
struct HufWork {
    int frame;     // index into frames[]
    int src;       // first byte of the block's literals section, relative to the frame
    int comp_end;  // ... and of its sequences section (where the literals section ends)
    int lit_at;    // where the block's literals go in the frame's literal area
    int regen;     // how many there are
    int tree;      // the literals section that holds the tree description (src itself unless the block's literals are treeless)
};

// ---- kernel 0: a thread per frame walks its block headers and lists its Huffman-coded blocks ------------------------------------
__global__ __launch_bounds__(64) void zstd_index_kernel(const uint8_t *__restrict__ src, const wsx_zstd_frame *__restrict__ frames, int n_frames,
                                                      HufWork *__restrict__ work, int *__restrict__ n_work, int32_t *__restrict__ status)
{
    const int frame = blockIdx.x * 64 + threadIdx.x;
    if (frame >= n_frames) return;
    const wsx_zstd_frame F = frames[frame];
    const uint8_t *f = src + F.src_offset;
    // a first walk decides whether the frame is one to decode at all (a frame that fails half-way must leave nothing on the list),
    // a second one lists its Huffman-coded blocks
    int nb = 0;
    const int st = walk_blocks(f, (int)F.src_bytes, F.dst_bytes, &nb, [](int, const BlockRec &) {});
    if (st != Z_OK) {
        status[frame] = st;
        return;
    }
    walk_blocks(f, (int)F.src_bytes, F.dst_bytes, &nb, [&](int, const BlockRec &B) {
        if (B.type != 2 || B.tree < 0) return;   // (raw and RLE literals: the sequence kernel reads them where they lie)
        const int at = atomicAdd(n_work, 1);
        work[at] = HufWork{frame, B.src, B.seq_at, B.lit_at, B.regen, B.tree};
        atomicAdd(n_work + 1 + size_class(B.regen), 1);
    });
}

// ---- kernel 0b: the list in the order it is worked off -- the blocks with the most literals first ---------------------------------
// A block's time is its longest stream's chain of look-ups, in proportion to its literals; a wavefront lasts as long as the longest of
// its four blocks, and the launch as long as the wavefront that ends last.  In the order the frames are walked a 128 KB block and
// the 50 KB rest of its chunk alternate: every wavefront lasted as long as a full block, and the last quarter of them started
// when the first three quarters were done (768 wavefronts fit the chip: 5.2 ms for 4 096 blocks; 2.9 ms ordered).
__global__ __launch_bounds__(256) void zstd_order_kernel(const HufWork *__restrict__ work, int *__restrict__ counters, HufWork *__restrict__ ordered)
{
    const int n_work = counters[0];
    int first[Z_CLASSES];   // (indexed by constants only: registers)
    int run = 0;
#pragma unroll
    for (int k = Z_CLASSES - 1; k >= 0; k--) {
        first[k] = run;
        run += counters[1 + k];
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_work; i += gridDim.x * 256) {
        const HufWork W = work[i];
        const int k = size_class(W.regen);
        int base = 0;
#pragma unroll
        for (int q = 0; q < Z_CLASSES; q++) base = q == k ? first[q] : base;
        ordered[base + atomicAdd(counters + 1 + Z_CLASSES + k, 1)] = W;
    }
}

// ---- kernel 1: the Huffman literals, four blocks to a wavefront ------------------------------------------------------------------
// Lane = 16 g + j: group g works on block 4 w + g of the list.  Its lane 0 reads the tree description; all 64 lanes build the four
// tables one after the other; then lanes j < 4 decode the block's four streams while all sixteen lanes of the group fetch the
// streams' next input windows and flush their symbols.  The symbol loop is branch-free and holds nothing but LDS: a stream is a chain
// of dependent look-ups, a global load or store between two of them (and the s_waitcnt they share with everything else) costs more
// than the look-up, and so does the bookkeeping of lanes that leave a loop at different times.
//   input : every stream has a ring of two 256-byte windows of its compressed bytes, aligned on absolute addresses; between two
//           flushes a stream moves at most 28 refills x 4 bytes, and its next window is asked for as soon as the upper one is used
//           up and put into the ring at the flush after -- so a refill never finds the ring short;
//   output: up to 112 symbols a stream collect in LDS and leave together.
__global__ __launch_bounds__(64) void zstd_literals_kernel(const uint8_t *__restrict__ src, long long src_total, const wsx_zstd_frame *__restrict__ frames,
                                                          const HufWork *__restrict__ work, const int *__restrict__ n_work_p, uint8_t *__restrict__ lits,
                                                          int32_t *__restrict__ status)
{
    __shared__ uint32_t huf2[ZG][1 << HUF_MAX_BITS];                  // the two-symbol tables
    // scratch of the set-up (normalised counts; the one-symbol table a two-symbol table is made from), then the streams' input rings
    // and output buffers
    constexpr int RING_BYTES = ZG * 4 * 2 * Z_WIN, OBUF_BYTES = ZG * 4 * Z_OUT;
    constexpr int WORK_BYTES = RING_BYTES + OBUF_BYTES > (2 << HUF_MAX_BITS) ? RING_BYTES + OBUF_BYTES : (2 << HUF_MAX_BITS);
    __shared__ __attribute__((aligned(16))) uint8_t workmem[WORK_BYTES];
    static_assert(2 * Z_LOOKUPS + 8 + 8 <= Z_WIN / 2, "a period moves a stream less than half a window");
    static_assert(ZG * 1024 <= WORK_BYTES, "the groups' count arrays fit the scratch");
    __shared__ uint8_t huf_w[ZG][256];           // weights, then code lengths
    __shared__ uint16_t huf_start[ZG][256];      // first cell of every symbol
    __shared__ FseTableW fse_w[ZG];
    __shared__ int info[ZG][4], strm[ZG][12];
    __shared__ uint8_t tree_stage[ZG][TREE_STAGE];

    const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
    const int n_work = *n_work_p;
    for (int base_item = blockIdx.x * ZG; base_item < n_work; base_item += gridDim.x * ZG) {
        const int item = base_item + g;
        const bool have = item < n_work;
        const HufWork W = have ? work[item] : HufWork{0, 0, 0, 0, 0, 0};
        const wsx_zstd_frame F = frames[W.frame];
        const uint8_t *f = src + F.src_offset;
        const uint8_t *p = f + W.src;
        // ---- the tree description of the group's block (its lane 0) ----
        // (the group fetches it first -- the literals header and at most 129 bytes of description --: lane 0 reads it a few bits at a
        // time, every read waiting for the one before)
        for (int i = j; i < TREE_STAGE; i += 16) tree_stage[g][i] = have && W.tree + i < F.src_bytes ? f[W.tree + i] : 0;
        wave_sync();
        if (j == 0) {
            int st = have ? Z_OK : Z_UNSUPPORTED, n = 0, took = 0, max_bits = 0, hl = 0, comp = 0, streams = 1;
            uint8_t *w = huf_w[g];
            int16_t *freq_s = (int16_t *)(workmem + g * 1024);
            uint16_t *next_s = (uint16_t *)(workmem + g * 1024 + 512);
            if (st == Z_OK) {
                const int sf = (p[0] >> 2) & 3;
                hl = sf < 2 ? 3 : sf + 2;
                comp = W.comp_end - W.src - hl;
                streams = sf == 0 ? 1 : 4;
                // the tree description: at the head of this block's literals section, or -- treeless literals -- of an earlier block's
                // (every work item builds its table from a description, so a treeless block waits for nobody)
                const uint8_t *tp = tree_stage[g];
                const int tsf = (tp[0] >> 2) & 3, thl = tsf < 2 ? 3 : tsf + 2, tw = tsf < 2 ? 10 : tsf == 2 ? 14 : 18;
                uint64_t tv = 0;
                for (int i = 0; i < thl; i++) tv |= (uint64_t)tp[i] << (8 * i);
                const int tcomp = (int)((tv >> (4 + tw)) & ((1u << tw) - 1));   // (walk_blocks found that section inside its block)
                const uint8_t *lp = tp + thl;
                const int hb = tcomp >= 1 ? lp[0] : 0;
                if (comp < 1 || tcomp < 1) st = Z_CORRUPT;
                else if (hb >= 128) {
                    n = hb - 127;
                    const int bytes = (n + 1) / 2;
                    if (1 + bytes > tcomp) st = Z_CORRUPT;
                    else
                        for (int i = 0; i < n; i++) w[i] = (i & 1) ? lp[1 + i / 2] & 15 : lp[1 + i / 2] >> 4;
                    took = 1 + bytes;
                } else {
                    int h = 0;
                    if (hb == 0 || 1 + hb > tcomp) st = Z_CORRUPT;
                    else st = fse_read(fse_w[g], lp + 1, hb, 6, 255, freq_s, next_s, &h);
                    if (st == Z_OK) {
                        const FseTableW &ft = fse_w[g];
                        const uint8_t *bs = lp + 1 + h;
                        const int bl = hb - h;
                        if (bl < 1 || bs[bl - 1] == 0) st = Z_CORRUPT;
                        else {
                            int off = bl * 8 - (8 - hibit(bs[bl - 1]));
                            uint32_t s1 = (uint32_t)back_bits(bs, ft.al, off), s2 = (uint32_t)back_bits(bs, ft.al, off);
                            for (;;) {
                                if (n >= 255) { st = Z_CORRUPT; break; }
                                w[n++] = ft.symbol[s1];
                                s1 = ft.base[s1] + (uint32_t)back_bits(bs, ft.nbits[s1], off);
                                if (off < 0) { w[n++] = ft.symbol[s2]; break; }
                                if (n >= 255) { st = Z_CORRUPT; break; }
                                w[n++] = ft.symbol[s2];
                                s2 = ft.base[s2] + (uint32_t)back_bits(bs, ft.nbits[s2], off);
                                if (off < 0) { w[n++] = ft.symbol[s1]; break; }
                            }
                        }
                    }
                    took = 1 + hb;
                }
                if (W.tree != W.src) took = 0;   // (treeless: the streams start right behind the literals header)
            }
            // weights -> code lengths (the last weight completes a power of two), then the first cell of every symbol
            if (st == Z_OK) {
                uint64_t sum = 0;
                for (int i = 0; i < n; i++) {
                    if (w[i] > HUF_MAX_BITS) st = Z_CORRUPT;
                    sum += w[i] ? 1ull << (w[i] - 1) : 0;
                }
                if (sum == 0 || n > 255) st = Z_CORRUPT;
                if (st == Z_OK) {
                    max_bits = 63 - __builtin_clzll(sum) + 1;
                    const uint64_t left = (1ull << max_bits) - sum;
                    if (max_bits > HUF_MAX_BITS || (left & (left - 1))) st = Z_CORRUPT;
                    else {
                        const int last = 63 - __builtin_clzll(left) + 1;
                        for (int i = 0; i < n; i++) w[i] = w[i] ? (uint8_t)(max_bits + 1 - w[i]) : 0;
                        w[n] = (uint8_t)(max_bits + 1 - last);
                        n++;
                        uint32_t count[HUF_MAX_BITS + 2], idx[HUF_MAX_BITS + 2];
                        for (int i = 0; i <= HUF_MAX_BITS + 1; i++) count[i] = 0;
                        for (int s = 0; s < n; s++) count[w[s]]++;
                        idx[max_bits] = 0;
                        for (int i = max_bits; i >= 1; i--) idx[i - 1] = idx[i] + count[i] * (1u << (max_bits - i));   // the longest codes first
                        if (idx[0] != (1u << max_bits)) st = Z_CORRUPT;
                        else
                            for (int s = 0; s < n; s++) {
                                huf_start[g][s] = w[s] ? (uint16_t)idx[w[s]] : 0;
                                if (w[s]) idx[w[s]] += 1u << (max_bits - w[s]);
                            }
                    }
                }
            }
            // the streams: their bytes and how many symbols each brings
            int *q = strm[g];
            for (int k = 0; k < 12; k++) q[k] = 0;
            int per = W.regen;
            if (st == Z_OK) {
                const uint8_t *sp = p + hl + took;
                const int sl = comp - took, first = (int)(sp - f);
                if (streams == 1) {
                    if (sl < 1) st = Z_CORRUPT;
                    q[0] = first, q[4] = first + sl, q[8] = W.regen;
                } else {
                    const int s1 = sl >= 6 ? sp[0] | (sp[1] << 8) : 0, s2 = sl >= 6 ? sp[2] | (sp[3] << 8) : 0, s3 = sl >= 6 ? sp[4] | (sp[5] << 8) : 0;
                    const int s4 = sl - 6 - s1 - s2 - s3, lastn = W.regen - 3 * ((W.regen + 3) / 4);
                    per = (W.regen + 3) / 4;
                    if (sl < 6 || s4 < 1 || s1 < 1 || s2 < 1 || s3 < 1 || lastn < 1) st = Z_CORRUPT;
                    q[0] = first + 6, q[1] = q[0] + s1, q[2] = q[1] + s2, q[3] = q[2] + s3;
                    q[4] = q[1], q[5] = q[2], q[6] = q[3], q[7] = q[3] + s4;
                    q[8] = q[9] = q[10] = per, q[11] = lastn;
                }
                if (st != Z_OK)
                    for (int k = 8; k < 12; k++) q[k] = 0;
            }
            info[g][0] = max_bits, info[g][1] = n, info[g][2] = st, info[g][3] = per;
            if (have && st != Z_OK && status) atomicMax(&status[W.frame], st);
        }
        wave_sync();
        // ---- the four tables, one after the other, by all 64 lanes ----
        for (int t = 0; t < ZG; t++) {
            if (info[t][2] != Z_OK) continue;
            const int max_bits = info[t][0], nsym = info[t][1];
            uint16_t *huf1 = (uint16_t *)workmem;
            for (int s = 0; s < nsym; s++) {   // the one-symbol table: symbol s fills 2^(max_bits - length) cells
                const int len_bits = huf_w[t][s];
                if (!len_bits) continue;
                const int cells = 1 << (max_bits - len_bits), c0 = huf_start[t][s];
                const uint16_t cell = (uint16_t)((len_bits << 8) | s);
                for (int i = lane; i < cells; i += 64) huf1[c0 + i] = cell;
            }
            wave_sync();
            for (int i = lane; i < (1 << max_bits); i += 64) {   // ... and the two-symbol table: the cell of what follows the first code
                const uint32_t c1 = huf1[i];
                huf2[t][i] = cell2(c1, huf1[(i << (c1 >> 8)) & ((1 << max_bits) - 1)], max_bits);
            }
            wave_sync();
        }
        // ---- the streams ----
        const int *q = strm[g];
        const bool ok = info[g][2] == Z_OK;
        const int max_bits = ok ? info[g][0] : HUF_MAX_BITS, per = info[g][3];
        uint8_t *L = lits + F.dst_offset + W.lit_at;
        const long long base_lo = -(long long)F.src_offset, base_hi = src_total - F.src_offset;   // what may be read, relative to f
        const uintptr_t ab = (uintptr_t)f;
        const int s = j & 3;                            // the stream a decoding lane works on (j < 4)
        const int fs = j >> 2, fj = j & 3;              // as a fetching / flushing lane: stream fs, its quarter fj
        const int lo_off = q[s], hi_off = q[4 + s], want = q[8 + s];
        const bool dec = ok && j < 4 && want > 0;
        const bool fetches = ok && q[8 + fs] > 0;
        int st = Z_OK;
        if (dec && (hi_off - lo_off < 1 || f[hi_off - 1] == 0)) st = Z_CORRUPT;
        uint64_t cont = 0;
        int used = 0, done = 0;
        int pp = hi_off - 8;                            // offset of the container's lowest byte (relative to the frame)
        if (dec && st == Z_OK) {
            for (int i = 0; i < 8; i++) {
                const int a = pp + i;
                cont |= (uint64_t)(a >= lo_off ? f[a] : 0) << (8 * i);
            }
            used = 8 - hibit(f[hi_off - 1]);
        }
        // (the bits of the stream not yet consumed = 8 (pp + 8 - lo_off) - used, at any time: a refill keeps it; zero when the last
        // symbol is out, or the stream is corrupt)
        const bool live = dec && st == Z_OK;
        uint8_t *ring = workmem + (g * 4) * 2 * Z_WIN, *obuf = workmem + RING_BYTES + (g * 4) * Z_OUT;
        // window w of a stream = absolute addresses [w * Z_WIN, (w + 1) * Z_WIN); ring index = address mod 2 Z_WIN
        long long fetch_lo = fetches ? (long long)(((ab + (uintptr_t)q[4 + fs] - 1) / Z_WIN) * Z_WIN) + Z_WIN : 0;   // lowest address held
        // The next lower window of stream fs (four lanes, four 16-byte pieces each) is ASKED FOR at one flush and PUT INTO THE RING at the
        // next: the loads are in flight while the streams decode a period (nearly every flush some stream of the wavefront wants a
        // window, and a load waited for on the spot was ~1.3 us of every ~8).  Safe: a window is asked for at the first flush that
        // finds the upper one used up -- the stream is then more than half a window above the lower one's bottom (it moves less than
        // half a window a period), so the period in between stays inside what the ring holds.
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 asked[4] = {};
        long long asked_at = 0;
        bool asked_for = false;
        auto ask = [&](bool go) {
            if (go) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const long long a = fetch_lo - Z_WIN + 64 * k + 16 * fj;
                    u32x4 v = {0, 0, 0, 0};
                    const long long rel = a - (long long)ab;
                    // (through `f`, a pointer into the kernel's argument: a GLOBAL load -- an address made from an integer is a flat one,
                    // which every wait for LDS in the symbol loop would wait for too)
                    if (rel >= base_lo && rel + 16 <= base_hi) v = *(const u32x4 *)(f + rel);
                    else if (rel + 16 > base_lo && rel < base_hi) {
                        uint32_t t4[4] = {0, 0, 0, 0};
                        for (int i = 0; i < 16; i++)
                            if (rel + i >= base_lo && rel + i < base_hi) t4[i >> 2] |= (uint32_t)f[rel + i] << (8 * (i & 3));
                        v = u32x4{t4[0], t4[1], t4[2], t4[3]};
                    }
                    asked[k] = v;
                }
                asked_at = fetch_lo - Z_WIN;
                asked_for = true;
                fetch_lo -= Z_WIN;
            }
        };
        auto put = [&]() {
            if (asked_for) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    *(u32x4 *)(ring + fs * 2 * Z_WIN + (int)((asked_at + 64 * k + 16 * fj) & (2 * Z_WIN - 1))) = asked[k];
                asked_for = false;
            }
        };
        ask(fetches), put();
        ask(fetches), put();
        wave_sync();
        const int shift = 64 - max_bits;
        const uint8_t *myring = ring + s * 2 * Z_WIN;
        uint8_t *myout = obuf + s * Z_OUT;
        const uint32_t *table = huf2[g];
        int flushed = 0;                                // symbols of MY fetch stream already written
        const uint32_t ab32 = (uint32_t)ab;             // (the ring is indexed by the low bits of the absolute address)
        for (;;) {
            int cnt = 0;
            // At the top of an iteration `used` <= 31: two look-ups of at most 11 bits leave it <= 53, and a refill (which needs 32
            // used bits to make room) brings that back to <= 21.  (Three look-ups a refill would reach 64 and come back as 32.)
            // A period (28 iterations) takes at most 112 symbols and 120 bytes: while every stream of the wavefront is further than
            // that from both its ends -- all but the last period or two of a block -- the loop is the bare chain: a refill that
            // selects, two look-ups, no end in sight to test for.
            const bool far = !live || (want - done >= 2 * Z_LOOKUPS && pp - lo_off >= 2 * Z_LOOKUPS + 8);
            if (__all(far)) {
                if (live) {
                    // (the four bytes below the container are read an iteration ahead: the look-ups do not wait for the ring)
                    auto below4 = [&](int at) {
                        const uint32_t a = ab32 + (uint32_t)at - 4u;
                        const uint32_t wlo = *(const uint32_t *)(myring + ((a & ~3u) & (2 * Z_WIN - 1)));
                        const uint32_t whi = *(const uint32_t *)(myring + (((a & ~3u) + 4u) & (2 * Z_WIN - 1)));
                        return __builtin_amdgcn_alignbyte(whi, wlo, a & 3u);
                    };
                    uint32_t w = below4(pp);
                    for (int it = 0; it < Z_LOOKUPS / 2; it++) {
                        const bool re = used >= 32;
                        cont = re ? (cont << 32) | w : cont;
                        used = re ? used - 32 : used;
                        pp = re ? pp - 4 : pp;
                        w = below4(pp);
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const uint32_t e = table[(uint32_t)((cont << used) >> shift)];
                            *(uint16_t *)(myout + cnt) = (uint16_t)e;
                            used += (int)((e >> 16) & 15), cnt += (int)(e >> 24);
                        }
                    }
                    done += cnt;
                }
            } else if (live) {
                for (int it = 0; it < Z_LOOKUPS / 2; it++) {
                    // four more bytes from below when 32 bits are used up (the ring holds them: see above); no branch
                    const uint32_t a = ab32 + (uint32_t)pp - 4u;
                    const uint32_t wlo = *(const uint32_t *)(myring + ((a & ~3u) & (2 * Z_WIN - 1)));
                    const uint32_t whi = *(const uint32_t *)(myring + (((a & ~3u) + 4u) & (2 * Z_WIN - 1)));
                    uint32_t w = __builtin_amdgcn_alignbyte(whi, wlo, a & 3u);
                    const int below = pp - lo_off;   // bytes of the stream below pp: what lies below the stream reads as zero
                    w = below >= 4 ? w : below <= 0 ? 0u : w & (~0u << (8 * (4 - below)));
                    const bool re = used >= 32;
                    cont = re ? (cont << 32) | w : cont;
                    used = re ? used - 32 : used;
                    pp = re ? pp - 4 : pp;
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const uint32_t e = table[(uint32_t)((cont << used) >> shift)];
                        const int ns = (int)(e >> 24), rem = want - done;
                        const int take = ns < rem ? ns : rem;
                        const int bits = take == ns ? (int)((e >> 16) & 15) : take ? (int)((e >> 20) & 15) : 0;
                        *(uint16_t *)(myout + cnt) = (uint16_t)e;   // (both symbols; what is not taken is overwritten or never flushed)
                        used += bits, cnt += take, done += take;
                    }
                }
            }
            wave_sync();
            // ---- flush: the four lanes of stream fs write a quarter of its buffer each; then its next window where the upper is used up ----
            const int src_lane = (lane & ~15) | fs;
            const int cnt_fs = __shfl(cnt, src_lane, 64);
            const int p_fs = __shfl(pp, src_lane, 64);
            put();   // (the window asked for a period ago; then the next one is asked for BEFORE the symbols leave: loads and stores
            ask(fetches && ((long long)ab + p_fs + 8) <= fetch_lo + Z_WIN);   // complete in order, a load behind them would wait for them)
            {        // (a refill reads the aligned words around pp - 4: up to pp + 3)
                uint8_t *d = L + (long long)fs * per + flushed;
                const uint8_t *o = obuf + fs * Z_OUT;
                const int q0 = (cnt_fs * fj) >> 2, q1 = (cnt_fs * (fj + 1)) >> 2;
                for (int k = q0; k < q1; k++) d[k] = o[k];
                flushed += cnt_fs;
            }
            wave_sync();
            if (!__any(live && done < want)) break;
        }
        if (live && 8 * (pp + 8 - lo_off) - used != 0) st = Z_CORRUPT;
        if (dec && st != Z_OK && status) atomicMax(&status[W.frame], st);
        wave_sync();
    }
}

// ---- kernel 2: a wavefront per frame walks its blocks in order -- raw, RLE, or sequences executed over the literals ---------------
__global__ __launch_bounds__(64) void zstd_sequences_kernel(const uint8_t *__restrict__ src, const wsx_zstd_frame *__restrict__ frames, uint8_t *__restrict__ dst,
                                                           uint8_t *lits_w, int32_t *__restrict__ status)
{
    __shared__ BlockRec blocks[Z_MAX_BLOCKS];
    __shared__ int n_blocks, frame_status;
    __shared__ int sh[8];                       // the sequences section: count, status, header bytes
    __shared__ FseTable fse[3];                 // LL, OF, ML (they persist from block to block: "repeat" mode)
    __shared__ int16_t freq_s[256];
    __shared__ uint16_t next_s[256];
    __shared__ uint32_t seq_ll[SEQ_CHUNK], seq_ml[SEQ_CHUNK], seq_of[SEQ_CHUNK];
    __shared__ uint8_t seq_stage[SEQ_STAGE];    // a block's sequences section, where it fits (a longer one is read where it lies)

    const int frame = blockIdx.x, lane = threadIdx.x;
    const wsx_zstd_frame F = frames[frame];
    const uint8_t *f = src + F.src_offset;
    uint8_t *out = dst + F.dst_offset;
    const uint8_t *lit = lits_w + F.dst_offset;
    const long long cap = F.dst_bytes;
    if (lane == 0) {
        int nb = 0;
        int st = walk_blocks(f, (int)F.src_bytes, cap, &nb, [&](int b, const BlockRec &B) { blocks[b] = B; });
        if (st == Z_OK && status && status[frame] != 0) st = status[frame];   // (a literal stream of the frame was found corrupt)
        n_blocks = nb;
        frame_status = st;
        fse[0].valid = fse[1].valid = fse[2].valid = 0;
    }
    wave_sync();
    int st = frame_status;
    long long o = 0;               // bytes written
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8;
    for (int b = 0; b < n_blocks && st == Z_OK; b++) {
        const BlockRec B = blocks[b];
        if (B.type == 0 || B.type == 1) {
            if (o + B.size > cap) { st = Z_CORRUPT; break; }
            const uint8_t *p = f + B.src;
            if (B.type == 0) wave_copy(out + o, p, B.size, lane);
            else
                for (int i = lane; i < B.size; i += 64) out[o + i] = p[0];
            o += B.size;
            __threadfence_block();
            continue;
        }
        const uint8_t *L = lit + B.lit_at;
        {   // raw and RLE literals are not in the literal area yet (zstd_literals_kernel does the Huffman-coded ones): put them there
            const uint8_t *lp = f + B.src;
            const int ltype = lp[0] & 3, sf = (lp[0] >> 2) & 3;
            if (ltype < 2) {
                const int hl = (sf == 0 || sf == 2) ? 1 : sf == 1 ? 2 : 3;
                uint8_t *Lw = lits_w + F.dst_offset + B.lit_at;
                if (ltype == 0) wave_copy(Lw, lp + hl, B.regen, lane);
                else
                    for (int i = lane; i < B.regen; i += 64) Lw[i] = lp[hl];
                __threadfence_block();
            }
        }
        // The sequences section is read by ONE lane, a few bits at a time, every read waiting for the one before: the wave puts the
        // section into LDS first, where such a read costs a seventh of what it costs from memory.  (VBZ chunks bring a dozen sequences
        // a block: 0.64 -> 0.62 ms for 2 048 of them; it is text-like content, thousands of sequences a block, that this is for.)
        const int ql = B.src + B.size - B.seq_at;
        const uint8_t *sq = f + B.seq_at;
        if (ql <= SEQ_STAGE) {
            wave_sync();   // (the block before has done with the bytes)
            for (int i = lane; i < ql; i += 64) seq_stage[i] = sq[i];
            wave_sync();
            sq = seq_stage;
        }
        // the section's header and its three tables (lane 0), then chunks of sequences: decoded by lane 0, executed by the wave
        if (lane == 0) {
            int s2 = Z_OK, used = 1, nseq = ql >= 1 ? sq[0] : 0;
            if (ql < 1) s2 = Z_CORRUPT;
            else if (nseq >= 128) {
                if (nseq < 255) { if (ql < 2) s2 = Z_CORRUPT; else { nseq = ((nseq - 128) << 8) + sq[1]; used = 2; } }
                else { if (ql < 3) s2 = Z_CORRUPT; else { nseq = sq[1] + (sq[2] << 8) + 0x7F00; used = 3; } }
            }
            if (s2 == Z_OK && nseq > 0) {
                if (ql < used + 1) s2 = Z_CORRUPT;
                else {
                    const int modes = sq[used++];
                    if (modes & 3) s2 = Z_CORRUPT;
                    for (int k = 0; k < 3 && s2 == Z_OK; k++) {
                        const int mode = (modes >> (6 - 2 * k)) & 3;
                        const int16_t *def = k == 0 ? LL_DEFAULT : k == 1 ? OF_DEFAULT : ML_DEFAULT;
                        const int ndef = k == 0 ? 36 : k == 1 ? 29 : 53, def_al = k == 1 ? 5 : 6, max_al = k == 1 ? 8 : 9, max_sym = k == 0 ? 35 : k == 1 ? 31 : 52;
                        if (mode == 0) {
                            for (int i = 0; i < ndef; i++) freq_s[i] = def[i];
                            s2 = fse_build(fse[k], freq_s, ndef, def_al, next_s);
                        } else if (mode == 1) {
                            if (ql < used + 1 || sq[used] > max_sym) s2 = Z_CORRUPT;
                            else {
                                fse[k].al = 0, fse[k].symbol[0] = sq[used], fse[k].nbits[0] = 0, fse[k].base[0] = 0, fse[k].valid = 1;
                                used++;
                            }
                        } else if (mode == 2) {
                            int h = 0;
                            s2 = fse_read(fse[k], sq + used, ql - used, max_al, max_sym, freq_s, next_s, &h);
                            used += h;
                        } else if (!fse[k].valid) s2 = Z_CORRUPT;   // repeat: the previous block's table
                    }
                }
            } else if (s2 == Z_OK && ql != used) s2 = Z_CORRUPT;
            sh[0] = nseq, sh[1] = s2, sh[2] = used;
        }
        wave_sync();
        const int nseq = sh[0];
        st = sh[1];
        if (st != Z_OK) break;
        int lit_at = 0;
        if (nseq > 0) {
            const uint8_t *bs = sq + sh[2];
            const int bl = ql - sh[2];
            int off = 0;                       // (lane 0's decoder state lives in its registers across the chunks)
            uint32_t sl_ = 0, so = 0, sm = 0;
            if (lane == 0) {
                if (bl < 1 || bs[bl - 1] == 0) sh[1] = Z_CORRUPT;
                else {
                    off = bl * 8 - (8 - hibit(bs[bl - 1]));
                    sl_ = (uint32_t)back_bits(bs, fse[0].al, off), so = (uint32_t)back_bits(bs, fse[1].al, off), sm = (uint32_t)back_bits(bs, fse[2].al, off);
                }
            }
            wave_sync();
            if (sh[1] != Z_OK) { st = sh[1]; break; }
            for (int c0 = 0; c0 < nseq && st == Z_OK; c0 += SEQ_CHUNK) {
                const int cn = nseq - c0 < SEQ_CHUNK ? nseq - c0 : SEQ_CHUNK;
                if (lane == 0) {
                    int s2 = Z_OK;
                    for (int i = 0; i < cn && s2 == Z_OK; i++) {
                        const int oc = fse[1].symbol[so], lc = fse[0].symbol[sl_], mc = fse[2].symbol[sm];
                        if (oc > 31 || lc > 35 || mc > 52) { s2 = Z_CORRUPT; break; }
                        const uint64_t ov = (1ull << oc) + back_bits(bs, oc, off);
                        const uint32_t mlen = ML_BASE[mc] + (uint32_t)back_bits(bs, ML_BITS[mc], off);
                        const uint32_t llen = LL_BASE[lc] + (uint32_t)back_bits(bs, LL_BITS[lc], off);
                        if (c0 + i + 1 < nseq) {   // the states move on in the order literal length, match length, offset
                            sl_ = fse[0].base[sl_] + (uint32_t)back_bits(bs, fse[0].nbits[sl_], off);
                            sm = fse[2].base[sm] + (uint32_t)back_bits(bs, fse[2].nbits[sm], off);
                            so = fse[1].base[so] + (uint32_t)back_bits(bs, fse[1].nbits[so], off);
                        }
                        if (off < 0) { s2 = Z_CORRUPT; break; }
                        uint64_t offset;   // 3.1.1.5: repeat offsets
                        if (ov > 3) {
                            offset = ov - 3;
                            rep2 = rep1, rep1 = rep0, rep0 = (uint32_t)offset;
                        } else {
                            uint32_t idx = (uint32_t)ov - 1;
                            if (llen == 0) idx++;
                            if (idx == 0) offset = rep0;
                            else {
                                offset = idx == 1 ? rep1 : idx == 2 ? rep2 : rep0 - 1;
                                if (idx > 1) rep2 = rep1;
                                rep1 = rep0, rep0 = (uint32_t)offset;
                            }
                        }
                        seq_ll[i] = llen, seq_ml[i] = mlen, seq_of[i] = (uint32_t)offset;
                        if (offset == 0 || offset > 0xFFFFFFFFull) s2 = Z_CORRUPT;
                    }
                    if (s2 == Z_OK && c0 + cn == nseq && off != 0) s2 = Z_CORRUPT;
                    sh[1] = s2;
                }
                wave_sync();
                if (sh[1] != Z_OK) { st = sh[1]; break; }
                for (int i = 0; i < cn; i++) {
                    const uint32_t llen = seq_ll[i], mlen = seq_ml[i], offset = seq_of[i];
                    if (lit_at + (long long)llen > B.regen || o + llen + mlen > cap || offset > o + llen) { st = Z_CORRUPT; break; }
                    wave_copy(out + o, L + lit_at, llen, lane);
                    o += llen;
                    lit_at += llen;
                    __threadfence_block();   // the match may read what this wave has just written
                    const uint8_t *m = out + o - offset;
                    if (offset >= mlen) {
                        wave_copy(out + o, m, mlen, lane);
                    } else {   // the match overlaps its own output: a pattern of `offset` bytes repeated
                        for (uint32_t k = lane; k < mlen; k += 64) out[o + k] = m[k % offset];
                    }
                    o += mlen;
                    __threadfence_block();
                }
                wave_sync();
            }
            if (st != Z_OK) break;
        }
        const int rest = B.regen - lit_at;
        if (o + rest > cap) { st = Z_CORRUPT; break; }
        wave_copy(out + o, L + lit_at, rest, lane);
        o += rest;
        __threadfence_block();
    }
    if (st == Z_OK && o != cap) st = Z_CORRUPT;   // the frame must bring exactly the content its header declares
    if (lane == 0 && status) status[frame] = st;
}

#define ZCHK(call)                                                                                                     \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            char m_[256];                                                                                              \
            snprintf(m_, sizeof m_, "%s: %s", #call, hipGetErrorString(e_));                                           \
            wsx_internal_set_error(m_);                                                                                \
            return WSX_ERR_HIP;                                                                                        \
        }                                                                                                              \
    } while (0)

} // namespace

extern "C" int wsx_zstd_decode(wsx_caller *c, const uint8_t *src, int64_t src_bytes, const wsx_zstd_frame *frames, int64_t n_frames,
                               uint8_t *dst, int64_t dst_bytes, uint8_t *scratch, int32_t *status)
try {
    if (!c || n_frames < 0 || src_bytes < 0 || dst_bytes < 0 || (n_frames > 0 && (!frames || !src || !dst || !scratch))) {
        wsx_internal_set_error("wsx_zstd_decode: null or negative argument");
        return WSX_ERR_INVALID;
    }
    if (n_frames > 0x7fffffff / Z_MAX_BLOCKS) {
        wsx_internal_set_error("wsx_zstd_decode: too many frames in one call");
        return WSX_ERR_INVALID;
    }
    for (int64_t i = 0; i < n_frames; i++) {
        const wsx_zstd_frame &F = frames[i];
        if (F.src_offset < 0 || F.src_bytes < 0 || F.dst_offset < 0 || F.dst_bytes < 0 || F.src_bytes > src_bytes || F.src_offset > src_bytes - F.src_bytes ||
            F.dst_bytes > dst_bytes || F.dst_offset > dst_bytes - F.dst_bytes) {
            wsx_internal_set_error("wsx_zstd_decode: a frame lies outside src or dst");
            return WSX_ERR_INVALID;
        }
        if (F.src_bytes > 0x7fffffff || F.dst_bytes > (int64_t)Z_MAX_BLOCKS << 17) {
            wsx_internal_set_error("wsx_zstd_decode: a frame of 2 GB or more, or of more content than 32 blocks hold (4 MB)");
            return WSX_ERR_INVALID;
        }
    }
    if (n_frames == 0) return WSX_SUCCESS;
    ZCHK(hipSetDevice(wsx_internal_device(c)));
    hipStream_t st = wsx_internal_stream(c);
    void *h = nullptr, *d = nullptr;
    hipEvent_t ev = nullptr;
    const size_t bytes = (size_t)n_frames * sizeof(wsx_zstd_frame);
    ZCHK(wsx_internal_vbz_slot(c, bytes, &h, &d, &ev));
    ZCHK(hipEventSynchronize(ev));
    memcpy(h, frames, bytes);
    ZCHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
    ZCHK(hipEventRecord(ev, st));
    // device scratch of the call: the status array (the index and literal kernels tell the sequence kernel of a corrupt frame through
    // it: it exists whether the caller wants it or not), the counter and the list of Huffman blocks
    const size_t st_bytes = ((size_t)n_frames * sizeof(int32_t) + 63) & ~(size_t)63;
    const size_t list_bytes = (size_t)n_frames * Z_MAX_BLOCKS * sizeof(HufWork);
    void *pscr = nullptr;
    ZCHK(wsx_internal_zstd_status(c, st_bytes + 128 + 2 * list_bytes, &pscr));
    int32_t *st_own = (int32_t *)pscr;
    int *counter = (int *)((char *)pscr + st_bytes);   // the blocks listed; how many of every size class; the ordering's cursors
    static_assert((1 + 2 * Z_CLASSES) * sizeof(int) <= 128, "the counters fit their place");
    HufWork *list = (HufWork *)((char *)pscr + st_bytes + 128);
    HufWork *ordered = (HufWork *)((char *)pscr + st_bytes + 128 + list_bytes);
    int32_t *st_dev = status ? status : st_own;
    ZCHK(hipMemsetAsync(st_dev, 0, (size_t)n_frames * sizeof(int32_t), st));
    ZCHK(hipMemsetAsync(counter, 0, 128, st));
    hipLaunchKernelGGL(zstd_index_kernel, dim3((unsigned)((n_frames + 63) / 64)), dim3(64), 0, st, src, (const wsx_zstd_frame *)d, (int)n_frames, list, counter, st_dev);
    ZCHK(hipGetLastError());
    // blocks a launch may bring: a block holds at most 128 KB of content; the grid strides over whatever the count turns out to be
    int64_t est = n_frames;
    for (int64_t i = 0; i < n_frames; i++) est += frames[i].dst_bytes >> 17;
    const unsigned grid = (unsigned)std::min<int64_t>(std::max<int64_t>((est + ZG - 1) / ZG, 1), 1 << 20);
    hipLaunchKernelGGL(zstd_order_kernel, dim3((unsigned)std::min<int64_t>((est + 255) / 256, 256)), dim3(256), 0, st, (const HufWork *)list, counter, ordered);
    ZCHK(hipGetLastError());
    hipLaunchKernelGGL(zstd_literals_kernel, dim3(grid), dim3(64), 0, st, src, (long long)src_bytes, (const wsx_zstd_frame *)d, (const HufWork *)ordered, (const int *)counter,
                       scratch, st_dev);
    ZCHK(hipGetLastError());
    hipLaunchKernelGGL(zstd_sequences_kernel, dim3((unsigned)n_frames), dim3(64), 0, st, src, (const wsx_zstd_frame *)d, dst, scratch, st_dev);
    ZCHK(hipGetLastError());
    return WSX_SUCCESS;
} catch (...) {
    return wsx_internal_on_exception();
}
