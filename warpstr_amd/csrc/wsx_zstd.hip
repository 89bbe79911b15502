// wsx_zstd.hip -- the Zstandard frames of VBZ chunks decoded on the GPU (the step in front of wsx_vbz.hip).
//
// Upstream reads `Raw/Signal` through h5py and the HDF5 filter plugin 32020 (Fast5.get_data_processed, src/schemas/fast5.py:50-52);
// the plugin hands every chunk to libzstd before it undoes StreamVByte.  Until round 6 that was the one part of a read's way from
// the file to the caller that stayed on the host: 0.06 ms of a reader process per read, more than libhdf5 and everything else a
// reader does together -- and the box's CPU share, not the GPU, decided how many reads per second a run could call from files.
// The format is RFC 8878.  What a VBZ chunk of nanopore samples holds is almost nothing but Huffman-coded literals (the value bytes
// of the StreamVByte block; the key bytes are the few matches): blocks of <= 128 KB, four interleaved-by-position streams each, a
// handful of sequences.  Hence the shape of the kernel:
//   * a workgroup per frame, a wavefront per block for the literals: lane 0 reads the tree description (direct weights, or FSE-coded
//     weights decoded with two states), the wave builds the 2^max_bits-entry decoding table in LDS, lanes 0..3 decode the four
//     streams side by side (a stream is a chain of dependent table look-ups: no parallelism inside it; the refill words are
//     fetched a refill ahead);
//   * then wave 0 alone walks the blocks in order: sequences section (predefined / RLE / FSE / repeat tables, built by lane 0 in
//     LDS), decoded by lane 0 a few hundred sequences at a time, executed by the whole wave (literal run, match -- overlapping
//     matches as a periodic copy), repeat offsets carried from block to block.
// Everything the format allows is decoded except dictionaries, treeless literals (a block that reuses the tree of the block
// before it: blocks are decoded side by side here) and frames of more than 32 blocks: the host never hands those over
// (warpstr_amd/_h5core.py / csrc/host_reader.cpp look at the headers and decompress such a frame themselves); a frame that turns out
// corrupt sets its status and leaves its output undefined.  Pinned against libzstd (tests/test_gpu_zstd.py) and oracle/zstd_oracle.c.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "../../include/warpstr_hip.h"

int wsx_internal_device(wsx_caller *c);
hipStream_t wsx_internal_stream(wsx_caller *c);
void wsx_internal_set_error(const char *msg);
extern "C" int wsx_internal_on_exception(void);
hipError_t wsx_internal_vbz_slot(wsx_caller *c, size_t bytes, void **host, void **dev, hipEvent_t *last_use);

namespace {

constexpr int ZW = 4;             // wavefronts of a workgroup = blocks of a frame whose literals are decoded side by side
constexpr int Z_MAX_BLOCKS = 32;  // blocks per frame (4 MB of content)
constexpr int HUF_MAX_BITS = 11;
constexpr int SEQ_CHUNK = 256;    // sequences decoded (lane 0) before the wave executes them

enum { Z_OK = 0, Z_UNSUPPORTED = 1, Z_CORRUPT = 2 };

struct FseTable {   // an FSE decoding table of up to 512 cells
    uint8_t symbol[512];
    uint8_t nbits[512];
    uint16_t base[512];
    int al;
    int valid;
};

struct BlockRec {
    int type, src, size;   // block type, first byte behind its header (relative to the frame), Block_Size
    int lit_at, regen;     // where its literals lie in the frame's literal area, how many there are
    int seq_at;            // first byte of its sequences section (relative to the frame); compressed blocks only
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
__device__ int fse_build(FseTable &t, const int16_t *freq, int nsym, int al, uint16_t *next)
{
    if (al > 9 || nsym > 256) return Z_CORRUPT;
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
__device__ int fse_read(FseTable &t, const uint8_t *src, int len, int max_al, int max_sym, int16_t *freq, uint16_t *next, int *took)
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

// the four bytes below `p` of a backward stream as a little-endian word -- bytes below `start` read as zero --, from the aligned
// word that holds p - 4 (`lo`) and the one above it (`hi`, the previous call's `lo`): one aligned load per refill
__device__ __forceinline__ uint32_t bytes_below(const uint8_t *p, const uint8_t *start, uint32_t lo, uint32_t hi)
{
    uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)((uintptr_t)(p - 4) & 3));
    const long long have = p - start;   // bytes of the stream below p
    if (have < 4) v = have <= 0 ? 0u : v & (~0u << (8 * (4 - (int)have)));
    return v;
}

// One Huffman stream (4.2.2): n_out symbols written to out.  table: (nbits << 8) | symbol per cell, max_bits wide.  `floor_`: no
// byte below it is touched.  Returns Z_OK if the stream ends exactly where its last symbol does.
__device__ int huf_stream(const uint16_t *table, int max_bits, const uint8_t *src, int len, uint8_t *out, int n_out, const uint8_t *floor_)
{
    if (len < 1 || src[len - 1] == 0) return Z_CORRUPT;
    const uint8_t *end = src + len;
    // the container: bit 63 = the stream's last bit; `used` bits of it are consumed (the padding and its marker first)
    uint64_t cont = 0;
    for (int i = 0; i < 8; i++) {
        const uint8_t *q = end - 8 + i;
        cont |= (uint64_t)(q >= src ? *q : 0) << (8 * i);
    }
    int used = 8 - hibit(src[len - 1]);
    const uint8_t *p = end - 8;                       // the container's lowest byte
    auto aligned_word = [&](const uint8_t *q) -> uint32_t {   // the aligned 4 bytes that hold q (never below floor_)
        const uint8_t *a = (const uint8_t *)((uintptr_t)q & ~(uintptr_t)3);
        if (a < floor_) {   // (the frame's first bytes: byte by byte)
            uint32_t v = 0;
            for (int i = 0; i < 4; i++)
                if (a + i >= floor_) v |= (uint32_t)a[i] << (8 * i);
            return v;
        }
        return *(const uint32_t *)a;
    };
    // words for the next refills, fetched ahead: hi holds the aligned word of p - 1 (or above), lo the one below it
    uint32_t hi = aligned_word(p - 1 >= floor_ ? p - 1 : floor_), lo = aligned_word(p - 4 >= floor_ ? p - 4 : floor_);
    if ((((uintptr_t)(p - 4)) & ~(uintptr_t)3) == (((uintptr_t)(p - 1)) & ~(uintptr_t)3)) hi = lo;   // p aligned: both in one word
    int total = len * 8 - used;                       // bits the symbols may take
    const int shift = 64 - max_bits;
    uint32_t acc = 0;
    int i = 0;
    for (; i < n_out; i++) {
        if (used >= 32) {   // refill: four more bytes from below
            const uint32_t w = bytes_below(p, src, lo, hi);
            cont = (cont << 32) | w;
            used -= 32;
            p -= 4;
            hi = lo;
            const uint8_t *nx = p - 4;
            lo = nx >= floor_ ? aligned_word(nx) : 0u;
        }
        const uint32_t cell = table[(uint32_t)((cont << used) >> shift)];
        used += cell >> 8;
        total -= (int)(cell >> 8);
        acc |= (cell & 255u) << (8 * (i & 3));
        if ((i & 3) == 3) {
            // (the destination of a stream need not be aligned: byte stores)
            out[i - 3] = (uint8_t)acc, out[i - 2] = (uint8_t)(acc >> 8), out[i - 1] = (uint8_t)(acc >> 16), out[i] = (uint8_t)(acc >> 24);
            acc = 0;
        }
    }
    for (int k = i & ~3; k < n_out; k++) out[k] = (uint8_t)(acc >> (8 * (k & 3)));
    return total == 0 ? Z_OK : Z_CORRUPT;
}

__global__ __launch_bounds__(64 * ZW) void zstd_decode_kernel(const uint8_t *__restrict__ src, const wsx_zstd_frame *__restrict__ frames,
                                                            uint8_t *__restrict__ dst, uint8_t *__restrict__ lits, int32_t *__restrict__ status)
{
    __shared__ BlockRec blocks[Z_MAX_BLOCKS];
    __shared__ int n_blocks, frame_status;
    __shared__ int sh[8];                       // wave 0's sequences section: count, status, header bytes
    __shared__ uint16_t huf[ZW][1 << HUF_MAX_BITS];
    __shared__ uint8_t huf_w[ZW][256];          // weights, then code lengths
    __shared__ int huf_info[ZW][8];             // per wave: max_bits, symbols, status, literals type, streams, stream sizes ...
    __shared__ int huf_start[ZW][260];          // first cell of every symbol
    __shared__ FseTable fse[3];                 // LL, OF, ML (they persist from block to block: "repeat" mode)
    __shared__ FseTable fse_w[ZW];              // the weights' table of each wave
    __shared__ int16_t freq_s[ZW][256];
    __shared__ uint16_t next_s[ZW][256];
    __shared__ uint32_t seq_ll[SEQ_CHUNK], seq_ml[SEQ_CHUNK], seq_of[SEQ_CHUNK];

    const wsx_zstd_frame F = frames[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t *f = src + F.src_offset;
    const int flen = (int)F.src_bytes;
    uint8_t *out = dst + F.dst_offset;
    uint8_t *lit = lits + F.dst_offset;
    const long long cap = F.dst_bytes;

    // ---- frame header and the walk over the block headers (one lane) ---------------------------------------------------------
    if (tid == 0) {
        int st = Z_OK, nb = 0, pos = 0;
        if (flen < 5 || f[0] != 0x28 || f[1] != 0xB5 || f[2] != 0x2F || f[3] != 0xFD) st = Z_CORRUPT;
        else {
            const int fhd = f[4], flag = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
            if (fhd & 8) st = Z_CORRUPT;
            else if (did) st = Z_UNSUPPORTED;
            pos = 5 + (single ? 0 : 1) + (flag == 0 ? (single ? 1 : 0) : flag == 1 ? 2 : flag == 2 ? 4 : 8);
        }
        int lit_at = 0;
        while (st == Z_OK) {
            if (pos + 3 > flen) { st = Z_CORRUPT; break; }
            const uint32_t bh = f[pos] | (f[pos + 1] << 8) | ((uint32_t)f[pos + 2] << 16);
            pos += 3;
            const int last = bh & 1, type = (bh >> 1) & 3, size = (int)(bh >> 3);
            if (nb >= Z_MAX_BLOCKS) { st = Z_UNSUPPORTED; break; }
            if (type == 3 || pos + (type == 1 ? 1 : size) > flen) { st = Z_CORRUPT; break; }
            BlockRec &B = blocks[nb++];
            B.type = type, B.src = pos, B.size = size, B.lit_at = lit_at, B.regen = 0, B.seq_at = 0;
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
                    if (ltype == 3) { st = Z_UNSUPPORTED; break; }
                }
                if (hl + comp > size || regen > (1 << 17) || lit_at + regen > cap) { st = Z_CORRUPT; break; }
                B.regen = regen;
                B.seq_at = pos + hl + comp;
                lit_at += regen;
            }
            pos += type == 1 ? 1 : size;
            if (last) break;
        }
        n_blocks = nb;
        frame_status = st;
    }
    __syncthreads();
    if (frame_status != Z_OK) {
        if (tid == 0 && status) status[blockIdx.x] = frame_status;
        return;
    }

    // ---- the literals of every compressed block, a wave per block -------------------------------------------------------------
    for (int b = wave; b < n_blocks; b += ZW) {
        const BlockRec B = blocks[b];
        if (B.type != 2) continue;
        const uint8_t *p = f + B.src;
        const int b0 = p[0], ltype = b0 & 3, sf = (b0 >> 2) & 3;
        const int hl = ltype < 2 ? ((sf == 0 || sf == 2) ? 1 : sf == 1 ? 2 : 3) : (sf < 2 ? 3 : sf + 2);
        const int regen = B.regen, comp = B.seq_at - B.src - hl;
        uint8_t *L = lit + B.lit_at;
        if (ltype == 0) {
            for (int i = lane; i < regen; i += 64) L[i] = p[hl + i];
            continue;
        }
        if (ltype == 1) {
            const uint8_t v = p[hl];
            for (int i = lane; i < regen; i += 64) L[i] = v;
            continue;
        }
        const int streams = sf == 0 ? 1 : 4;
        const uint8_t *lp = p + hl;
        int *info = huf_info[wave];
        uint8_t *w = huf_w[wave];
        if (lane == 0) {   // the tree description: weights
            int st = Z_OK, n = 0, took = 0;
            const int hb = comp >= 1 ? lp[0] : 0;
            if (comp < 1) st = Z_CORRUPT;
            else if (hb >= 128) {
                n = hb - 127;
                const int bytes = (n + 1) / 2;
                if (1 + bytes > comp) st = Z_CORRUPT;
                else
                    for (int i = 0; i < n; i++) w[i] = (i & 1) ? lp[1 + i / 2] & 15 : lp[1 + i / 2] >> 4;
                took = 1 + bytes;
            } else {
                int h = 0;
                if (hb == 0 || 1 + hb > comp) st = Z_CORRUPT;
                else st = fse_read(fse_w[wave], lp + 1, hb, 6, 255, freq_s[wave], next_s[wave], &h);
                if (st == Z_OK) {
                    const FseTable &ft = fse_w[wave];
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
            // weights -> code lengths (the last weight completes a power of two), then the first cell of every symbol
            int max_bits = 0;
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
                                huf_start[wave][s] = w[s] ? (int)idx[w[s]] : 0;
                                if (w[s]) idx[w[s]] += 1u << (max_bits - w[s]);
                            }
                    }
                }
            }
            info[0] = max_bits, info[1] = n, info[2] = st, info[3] = took;
        }
        wave_sync();
        if (info[2] != Z_OK) {
            if (lane == 0) atomicMax(&frame_status, info[2]);
            continue;
        }
        const int max_bits = info[0], nsym = info[1], took = info[3];
        for (int s = 0; s < nsym; s++) {   // the table: symbol s fills 2^(max_bits - length) cells
            const int len_bits = w[s];
            if (!len_bits) continue;
            const int cells = 1 << (max_bits - len_bits), c0 = huf_start[wave][s];
            const uint16_t cell = (uint16_t)((len_bits << 8) | s);
            for (int i = lane; i < cells; i += 64) huf[wave][c0 + i] = cell;
        }
        wave_sync();
        const uint8_t *sp = lp + took;
        const int sl = comp - took;
        int st = Z_OK;
        if (streams == 1) {
            if (lane == 0) st = huf_stream(huf[wave], max_bits, sp, sl, L, regen, f);
        } else if (sl < 6) {
            st = Z_CORRUPT;
        } else if (lane < 4) {
            const int s1 = sp[0] | (sp[1] << 8), s2 = sp[2] | (sp[3] << 8), s3 = sp[4] | (sp[5] << 8), s4 = sl - 6 - s1 - s2 - s3;
            const int per = (regen + 3) / 4, lastn = regen - 3 * per;
            if (s4 < 1 || lastn < 0) st = Z_CORRUPT;
            else {
                const int at = lane == 0 ? 0 : lane == 1 ? s1 : lane == 2 ? s1 + s2 : s1 + s2 + s3;
                const int sz = lane == 0 ? s1 : lane == 1 ? s2 : lane == 2 ? s3 : s4;
                st = huf_stream(huf[wave], max_bits, sp + 6 + at, sz, L + lane * per, lane < 3 ? per : lastn, f);
            }
        }
        if (st != Z_OK) atomicMax(&frame_status, st);
        wave_sync();
    }
    __threadfence_block();
    __syncthreads();
    if (frame_status != Z_OK || wave != 0) {
        if (tid == 0 && status) status[blockIdx.x] = frame_status;
        return;
    }

    // ---- wave 0: the blocks in order -- raw, RLE, or sequences executed over the literals --------------------------------------
    long long o = 0;               // bytes written
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8;
    int st = Z_OK;
    if (lane == 0) fse[0].valid = fse[1].valid = fse[2].valid = 0;
    wave_sync();
    for (int b = 0; b < n_blocks && st == Z_OK; b++) {
        const BlockRec B = blocks[b];
        if (B.type == 0 || B.type == 1) {
            if (o + B.size > cap) { st = Z_CORRUPT; break; }
            const uint8_t *p = f + B.src;
            for (int i = lane; i < B.size; i += 64) out[o + i] = B.type == 0 ? p[i] : p[0];
            o += B.size;
            continue;
        }
        const uint8_t *L = lit + B.lit_at;
        const uint8_t *sq = f + B.seq_at;
        const int ql = B.src + B.size - B.seq_at;
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
                            for (int i = 0; i < ndef; i++) freq_s[0][i] = def[i];
                            s2 = fse_build(fse[k], freq_s[0], ndef, def_al, next_s[0]);
                        } else if (mode == 1) {
                            if (ql < used + 1 || sq[used] > max_sym) s2 = Z_CORRUPT;
                            else {
                                fse[k].al = 0, fse[k].symbol[0] = sq[used], fse[k].nbits[0] = 0, fse[k].base[0] = 0, fse[k].valid = 1;
                                used++;
                            }
                        } else if (mode == 2) {
                            int h = 0;
                            s2 = fse_read(fse[k], sq + used, ql - used, max_al, max_sym, freq_s[0], next_s[0], &h);
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
            // lane 0's decoder state lives in its registers across the chunks
            int off = 0;
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
                        if (c0 + i + 1 < nseq) {
                            sl_ = fse[0].base[sl_] + (uint32_t)back_bits(bs, fse[0].nbits[sl_], off);
                            sm = fse[2].base[sm] + (uint32_t)back_bits(bs, fse[2].nbits[sm], off);
                            so = fse[1].base[so] + (uint32_t)back_bits(bs, fse[1].nbits[so], off);
                        }
                        if (off < 0) { s2 = Z_CORRUPT; break; }
                        uint64_t offset;
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
                    for (uint32_t k = lane; k < llen; k += 64) out[o + k] = L[lit_at + k];
                    o += llen;
                    lit_at += llen;
                    __threadfence_block();   // the match may read what this wave has just written
                    const uint8_t *m = out + o - offset;
                    if (offset >= mlen) {
                        for (uint32_t k = lane; k < mlen; k += 64) out[o + k] = m[k];
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
        for (int k = lane; k < rest; k += 64) out[o + k] = L[lit_at + k];
        o += rest;
        __threadfence_block();
    }
    if (st == Z_OK && o != cap) st = Z_CORRUPT;   // the frame must bring exactly the content its header declares
    if (lane == 0 && status) status[blockIdx.x] = st;
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
    if (n_frames > 0x7fffffff) {
        wsx_internal_set_error("wsx_zstd_decode: more than 2^31 - 1 frames in one call");
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
    if (status) ZCHK(hipMemsetAsync(status, 0, (size_t)n_frames * sizeof(int32_t), st));
    hipLaunchKernelGGL(zstd_decode_kernel, dim3((unsigned)n_frames), dim3(64 * ZW), 0, st, src, (const wsx_zstd_frame *)d, dst, scratch, status);
    ZCHK(hipGetLastError());
    return WSX_SUCCESS;
} catch (...) {
    return wsx_internal_on_exception();
}
