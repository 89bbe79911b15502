/* zstd_oracle.c -- CPU restatement of a Zstandard frame decoder.  TEST INFRASTRUCTURE ONLY (tests/, bench.py's checker): the product
 * never links or loads this file.
 *
 * What it restates: upstream reads a read's samples with h5py, and HDF5's filter plugin 32020 (ont-vbz-hdf-plugin, a third-party
 * dependency that is not in the upstream tree: src/schemas/fast5.py:50-52) hands every chunk's bytes to libzstd before it
 * undoes StreamVByte.  libzstd is a third-party dependency too (absent from /root/reference); its format is published: RFC 8878,
 * "Zstandard Compression and the application/zstd Media Type".  This file follows the RFC section by section (frame header 3.1.1.1,
 * blocks 3.1.1.2, literals section 3.1.1.3.1, Huffman tree description 4.2.1, sequences section 3.1.1.3.2, FSE table description
 * 4.1.1, sequence execution 3.1.1.4, repeat offsets 3.1.1.5, default distributions 3.1.1.3.2.2), in the shape of the format's
 * educational decoder: one bit at a time where the format is described one bit at a time.  No dictionaries, one frame.
 *
 * Pinned by: libzstd itself, which IS installed here and on the GPU box -- tests/test_zstd_oracle.py decodes the frames of the
 * upstream test file and frames libzstd makes at several levels from random, repetitive, text-like and StreamVByte-like input
 * (raw, RLE and compressed blocks; raw, RLE, Huffman and treeless literals; predefined, RLE, FSE and repeat sequence tables)
 * with both and compares every byte.  The device decoder (csrc/wsx_zstd.hip) is then held against libzstd directly and against
 * this file where a test wants to know WHICH part of a frame differs.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ERR_TRUNCATED (-1)   /* the input ends inside a field */
#define ERR_MAGIC (-2)       /* not a Zstandard frame */
#define ERR_UNSUPPORTED (-3) /* a dictionary id, a reserved block type */
#define ERR_CORRUPT (-4)     /* a field holds a value the format forbids */
#define ERR_ROOM (-5)        /* the output does not fit */

#define HUF_MAX_BITS 11
#define HUF_MAX_SYMBS 256
#define FSE_MAX_AL 9
#define FSE_MAX_SYMBS 256

static int highest_set_bit(uint64_t x)
{
    int r = -1;
    while (x) {
        x >>= 1;
        r++;
    }
    return r;
}

/* ---- forward bit reader (FSE table descriptions) ---------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *p;
    int64_t len;   /* bytes */
    int64_t bit;   /* bits consumed */
    int bad;
} fwd_t;

static uint32_t fwd_bits(fwd_t *s, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) {
        const int64_t b = s->bit + i;
        if (b >= 8 * s->len) {
            s->bad = 1;
            break;
        }
        v |= (uint32_t)((s->p[b >> 3] >> (b & 7)) & 1) << i;
    }
    s->bit += n;
    return v;
}

/* ---- backward bit reader (Huffman and FSE streams): `off` = bits not yet read; bits before the start read as zero -------------- */
static uint64_t back_bits(const uint8_t *src, int n, int64_t *off)
{
    *off -= n;
    int64_t at = *off, take = n;
    if (at < 0) {
        take += at;
        at = 0;
    }
    uint64_t v = 0;
    for (int64_t i = 0; i < take; i++) v |= (uint64_t)((src[(at + i) >> 3] >> ((at + i) & 7)) & 1) << i;
    if (*off < 0) v = -*off >= 64 ? 0 : v << -*off;
    return v;
}

/* ---- FSE decoding tables ------------------------------------------------------------------------------------------------------ */
typedef struct {
    uint8_t symbol[1 << FSE_MAX_AL];
    uint8_t nbits[1 << FSE_MAX_AL];
    uint16_t base[1 << FSE_MAX_AL];
    int al;
} fse_t;

static int fse_build(fse_t *t, const int16_t *freq, int nsym, int al)
{
    if (al > FSE_MAX_AL || nsym > FSE_MAX_SYMBS) return ERR_CORRUPT;
    const int size = 1 << al;
    uint16_t next[FSE_MAX_SYMBS];
    int high = size;
    t->al = al;
    for (int s = 0; s < nsym; s++)
        if (freq[s] == -1) {   /* "less than 1": one cell, at the end of the table */
            t->symbol[--high] = (uint8_t)s;
            next[s] = 1;
        }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        if (freq[s] <= 0) continue;
        next[s] = (uint16_t)freq[s];
        for (int i = 0; i < freq[s]; i++) {
            t->symbol[pos] = (uint8_t)s;
            do pos = (pos + step) & mask;
            while (pos >= high);
        }
    }
    if (pos != 0) return ERR_CORRUPT;
    for (int i = 0; i < size; i++) {
        const uint16_t x = next[t->symbol[i]]++;
        t->nbits[i] = (uint8_t)(al - highest_set_bit(x));
        t->base[i] = (uint16_t)(((uint32_t)x << t->nbits[i]) - size);
    }
    return 0;
}

static void fse_rle(fse_t *t, uint8_t symbol)
{
    t->al = 0;
    t->symbol[0] = symbol;
    t->nbits[0] = 0;
    t->base[0] = 0;
}

/* 4.1.1: the table description; returns the bytes it took (the stream is byte aligned behind it) */
static int64_t fse_read(fse_t *t, const uint8_t *src, int64_t len, int max_al, int max_sym)
{
    fwd_t in = {src, len, 0, 0};
    const int al = 5 + (int)fwd_bits(&in, 4);
    if (al > max_al) return ERR_CORRUPT;
    int32_t remaining = 1 << al;
    int16_t freq[FSE_MAX_SYMBS];
    int nsym = 0;
    while (remaining > 0 && nsym < FSE_MAX_SYMBS) {
        const int bits = highest_set_bit((uint64_t)remaining + 1) + 1;
        uint32_t val = fwd_bits(&in, bits);
        const uint32_t lower = (1u << (bits - 1)) - 1, thresh = (1u << bits) - 1 - (uint32_t)(remaining + 1);
        if ((val & lower) < thresh) {
            in.bit -= 1;
            val &= lower;
        } else if (val > lower) {
            val -= thresh;
        }
        const int16_t p = (int16_t)val - 1;
        remaining -= p < 0 ? -p : p;
        freq[nsym++] = p;
        if (p == 0) {
            uint32_t rep = fwd_bits(&in, 2);
            for (;;) {
                for (uint32_t i = 0; i < rep && nsym < FSE_MAX_SYMBS; i++) freq[nsym++] = 0;
                if (rep != 3) break;
                rep = fwd_bits(&in, 2);
            }
        }
        if (in.bad) return ERR_TRUNCATED;
    }
    if (remaining != 0 || nsym > max_sym + 1 || in.bad) return ERR_CORRUPT;
    const int rc = fse_build(t, freq, nsym, al);
    if (rc) return rc;
    return (in.bit + 7) >> 3;
}

/* ---- Huffman ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
    uint8_t symbol[1 << HUF_MAX_BITS];
    uint8_t nbits[1 << HUF_MAX_BITS];
    int max_bits;
} huf_t;

static int huf_build(huf_t *t, const uint8_t *bits, int nsym)
{
    uint32_t count[HUF_MAX_BITS + 2] = {0}, idx[HUF_MAX_BITS + 2];
    int max_bits = 0;
    for (int s = 0; s < nsym; s++) {
        if (bits[s] > HUF_MAX_BITS) return ERR_CORRUPT;
        if (bits[s] > max_bits) max_bits = bits[s];
        count[bits[s]]++;
    }
    if (max_bits == 0) return ERR_CORRUPT;
    t->max_bits = max_bits;
    idx[max_bits] = 0;
    for (int i = max_bits; i >= 1; i--) {   /* the longest codes come first in the table */
        idx[i - 1] = idx[i] + count[i] * (1u << (max_bits - i));
        memset(t->nbits + idx[i], i, idx[i - 1] - idx[i]);
    }
    if (idx[0] != (1u << max_bits)) return ERR_CORRUPT;
    for (int s = 0; s < nsym; s++)
        if (bits[s]) {
            const uint32_t len = 1u << (max_bits - bits[s]);
            memset(t->symbol + idx[bits[s]], s, len);
            idx[bits[s]] += len;
        }
    return 0;
}

static int huf_from_weights(huf_t *t, uint8_t *w, int n)
{
    uint64_t sum = 0;
    for (int i = 0; i < n; i++) {
        if (w[i] > HUF_MAX_BITS) return ERR_CORRUPT;
        sum += w[i] ? (uint64_t)1 << (w[i] - 1) : 0;
    }
    if (sum == 0) return ERR_CORRUPT;
    const int max_bits = highest_set_bit(sum) + 1;
    const uint64_t left = ((uint64_t)1 << max_bits) - sum;
    if (left & (left - 1)) return ERR_CORRUPT;   /* the last weight completes a power of two */
    const int last = highest_set_bit(left) + 1;
    uint8_t bits[HUF_MAX_SYMBS];
    if (n + 1 > HUF_MAX_SYMBS) return ERR_CORRUPT;
    for (int i = 0; i < n; i++) bits[i] = w[i] ? (uint8_t)(max_bits + 1 - w[i]) : 0;
    bits[n] = (uint8_t)(max_bits + 1 - last);
    return huf_build(t, bits, n + 1);
}

/* 4.2.1: the tree description; returns the bytes it took */
static int64_t huf_read(huf_t *t, const uint8_t *src, int64_t len)
{
    if (len < 1) return ERR_TRUNCATED;
    const int hb = src[0];
    uint8_t w[HUF_MAX_SYMBS];
    int n;
    int64_t took;
    if (hb >= 128) {   /* direct: 4 bits a weight */
        n = hb - 127;
        const int bytes = (n + 1) / 2;
        if (1 + bytes > len) return ERR_TRUNCATED;
        for (int i = 0; i < n; i++) w[i] = (i & 1) ? src[1 + i / 2] & 15 : src[1 + i / 2] >> 4;
        took = 1 + bytes;
    } else {           /* FSE-compressed weights, two interleaved states */
        if (hb == 0 || 1 + hb > len) return ERR_TRUNCATED;
        static fse_t ft;   /* (single-threaded test code) */
        const int64_t h = fse_read(&ft, src + 1, hb, 6, 255);
        if (h < 0) return h;
        const uint8_t *bs = src + 1 + h;
        const int64_t bl = hb - h;
        if (bl < 1 || bs[bl - 1] == 0) return ERR_CORRUPT;
        int64_t off = bl * 8 - (8 - highest_set_bit(bs[bl - 1]));
        uint32_t s1 = (uint32_t)back_bits(bs, ft.al, &off), s2 = (uint32_t)back_bits(bs, ft.al, &off);
        n = 0;
        for (;;) {
            if (n >= HUF_MAX_SYMBS - 1) return ERR_CORRUPT;
            w[n++] = ft.symbol[s1];
            s1 = ft.base[s1] + (uint32_t)back_bits(bs, ft.nbits[s1], &off);
            if (off < 0) {
                w[n++] = ft.symbol[s2];
                break;
            }
            if (n >= HUF_MAX_SYMBS - 1) return ERR_CORRUPT;
            w[n++] = ft.symbol[s2];
            s2 = ft.base[s2] + (uint32_t)back_bits(bs, ft.nbits[s2], &off);
            if (off < 0) {
                w[n++] = ft.symbol[s1];
                break;
            }
        }
        took = 1 + hb;
    }
    const int rc = huf_from_weights(t, w, n);
    return rc ? rc : took;
}

static int huf_stream(const huf_t *t, const uint8_t *src, int64_t len, uint8_t *dst, int64_t n_out)
{
    if (len < 1 || src[len - 1] == 0) return ERR_CORRUPT;
    int64_t off = len * 8 - (8 - highest_set_bit(src[len - 1]));
    const uint32_t mask = (1u << t->max_bits) - 1;
    uint32_t state = (uint32_t)back_bits(src, t->max_bits, &off);
    int64_t n = 0;
    while (off > -t->max_bits) {
        if (n >= n_out) return ERR_CORRUPT;
        dst[n++] = t->symbol[state];
        const int b = t->nbits[state];
        state = ((state << b) + (uint32_t)back_bits(src, b, &off)) & mask;
    }
    return (off == -t->max_bits && n == n_out) ? 0 : ERR_CORRUPT;
}

/* ---- sequences ---------------------------------------------------------------------------------------------------------------- */
static const int16_t LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static const uint32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512,
                                     1024, 2048, 4096, 8192, 16384, 32768, 65536};
static const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
                                     34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
static const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                    0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

typedef struct {
    huf_t huf;
    int have_huf;
    fse_t ll, of, ml;
    int have_seq[3];
    uint32_t rep[3];
    uint8_t *lit; /* literals of the current block */
} ctx_t;

static int64_t seq_table(fse_t *t, int *have, int mode, const uint8_t *src, int64_t len, const int16_t *def, int ndef, int def_al, int max_al,
                         int max_sym)
{
    if (mode == 0) {
        *have = 1;
        const int rc = fse_build(t, def, ndef, def_al);
        return rc ? rc : 0;
    }
    if (mode == 1) {
        if (len < 1) return ERR_TRUNCATED;
        if (src[0] > max_sym) return ERR_CORRUPT;
        fse_rle(t, src[0]);
        *have = 1;
        return 1;
    }
    if (mode == 2) {
        const int64_t h = fse_read(t, src, len, max_al, max_sym);
        if (h >= 0) *have = 1;
        return h;
    }
    return *have ? 0 : ERR_CORRUPT;   /* repeat: the previous block's table */
}

static int64_t block_compressed(ctx_t *c, const uint8_t *src, int64_t len, uint8_t *dst, int64_t at, int64_t cap)
{
    /* ---- literals section (3.1.1.3.1) ---- */
    if (len < 1) return ERR_TRUNCATED;
    const int ltype = src[0] & 3, sf = (src[0] >> 2) & 3;
    int64_t regen, comp, hl;
    int streams = 1;
    if (ltype < 2) {
        if (sf == 0 || sf == 2) { regen = src[0] >> 3; hl = 1; }
        else if (sf == 1) { if (len < 2) return ERR_TRUNCATED; regen = (src[0] >> 4) + ((int64_t)src[1] << 4); hl = 2; }
        else { if (len < 3) return ERR_TRUNCATED; regen = (src[0] >> 4) + ((int64_t)src[1] << 4) + ((int64_t)src[2] << 12); hl = 3; }
        comp = ltype == 0 ? regen : 1;
    } else {
        hl = sf < 2 ? 3 : sf + 2;
        if (len < hl) return ERR_TRUNCATED;
        uint64_t v = 0;
        for (int i = 0; i < hl; i++) v |= (uint64_t)src[i] << (8 * i);
        const int w = sf < 2 ? 10 : sf == 2 ? 14 : 18;
        regen = (int64_t)((v >> 4) & ((1u << w) - 1));
        comp = (int64_t)((v >> (4 + w)) & ((1u << w) - 1));
        streams = sf == 0 ? 1 : 4;
    }
    if (hl + comp > len || regen > (1 << 17)) return ERR_CORRUPT;
    const uint8_t *lp = src + hl;
    if (ltype == 0) memcpy(c->lit, lp, (size_t)regen);
    else if (ltype == 1) memset(c->lit, lp[0], (size_t)regen);
    else {
        int64_t took = 0;
        if (ltype == 2) {
            took = huf_read(&c->huf, lp, comp);
            if (took < 0) return took;
            c->have_huf = 1;
        } else if (!c->have_huf) return ERR_CORRUPT;   /* treeless: the previous block's tree */
        const uint8_t *sp = lp + took;
        const int64_t sl = comp - took;
        if (streams == 1) {
            const int rc = huf_stream(&c->huf, sp, sl, c->lit, regen);
            if (rc) return rc;
        } else {
            if (sl < 6) return ERR_TRUNCATED;
            const int64_t s1 = sp[0] | (sp[1] << 8), s2 = sp[2] | (sp[3] << 8), s3 = sp[4] | (sp[5] << 8), s4 = sl - 6 - s1 - s2 - s3;
            if (s4 < 1) return ERR_CORRUPT;
            const int64_t per = (regen + 3) / 4, lastn = regen - 3 * per;
            if (lastn < 0) return ERR_CORRUPT;
            const int64_t sz[4] = {s1, s2, s3, s4};
            const uint8_t *q = sp + 6;
            for (int i = 0; i < 4; i++) {
                const int rc = huf_stream(&c->huf, q, sz[i], c->lit + i * per, i < 3 ? per : lastn);
                if (rc) return rc;
                q += sz[i];
            }
        }
    }
    /* ---- sequences section (3.1.1.3.2) ---- */
    const uint8_t *sq = src + hl + comp;
    int64_t ql = len - hl - comp;
    if (ql < 1) return ERR_TRUNCATED;
    int64_t nseq = sq[0], used = 1;
    if (nseq >= 128) {
        if (nseq < 255) { if (ql < 2) return ERR_TRUNCATED; nseq = ((nseq - 128) << 8) + sq[1]; used = 2; }
        else { if (ql < 3) return ERR_TRUNCATED; nseq = sq[1] + ((int64_t)sq[2] << 8) + 0x7F00; used = 3; }
    }
    int64_t out = at, lit_at = 0;
    if (nseq > 0) {
        if (ql < used + 1) return ERR_TRUNCATED;
        const int modes = sq[used++];
        if (modes & 3) return ERR_CORRUPT;
        int64_t h = seq_table(&c->ll, &c->have_seq[0], modes >> 6, sq + used, ql - used, LL_DEFAULT, 36, 6, 9, 35);
        if (h < 0) return h;
        used += h;
        h = seq_table(&c->of, &c->have_seq[1], (modes >> 4) & 3, sq + used, ql - used, OF_DEFAULT, 29, 5, 8, 31);
        if (h < 0) return h;
        used += h;
        h = seq_table(&c->ml, &c->have_seq[2], (modes >> 2) & 3, sq + used, ql - used, ML_DEFAULT, 53, 6, 9, 52);
        if (h < 0) return h;
        used += h;
        const uint8_t *bs = sq + used;
        const int64_t bl = ql - used;
        if (bl < 1 || bs[bl - 1] == 0) return ERR_CORRUPT;
        int64_t off = bl * 8 - (8 - highest_set_bit(bs[bl - 1]));
        uint32_t sl_ = (uint32_t)back_bits(bs, c->ll.al, &off), so = (uint32_t)back_bits(bs, c->of.al, &off), sm = (uint32_t)back_bits(bs, c->ml.al, &off);
        for (int64_t i = 0; i < nseq; i++) {
            const int oc = c->of.symbol[so], lc = c->ll.symbol[sl_], mc = c->ml.symbol[sm];
            if (oc > 31 || lc > 35 || mc > 52) return ERR_CORRUPT;
            const uint64_t ov = ((uint64_t)1 << oc) + back_bits(bs, oc, &off);
            const uint32_t mlen = ML_BASE[mc] + (uint32_t)back_bits(bs, ML_BITS[mc], &off);
            const uint32_t llen = LL_BASE[lc] + (uint32_t)back_bits(bs, LL_BITS[lc], &off);
            if (i + 1 < nseq) {   /* the states move on in the order literal length, match length, offset */
                sl_ = c->ll.base[sl_] + (uint32_t)back_bits(bs, c->ll.nbits[sl_], &off);
                sm = c->ml.base[sm] + (uint32_t)back_bits(bs, c->ml.nbits[sm], &off);
                so = c->of.base[so] + (uint32_t)back_bits(bs, c->of.nbits[so], &off);
            }
            if (off < 0) return ERR_CORRUPT;
            uint64_t offset;   /* 3.1.1.5: repeat offsets */
            if (ov > 3) {
                offset = ov - 3;
                c->rep[2] = c->rep[1];
                c->rep[1] = c->rep[0];
                c->rep[0] = (uint32_t)offset;
            } else {
                uint32_t idx = (uint32_t)ov - 1;
                if (llen == 0) idx++;
                if (idx == 0) offset = c->rep[0];
                else {
                    offset = idx < 3 ? c->rep[idx] : c->rep[0] - 1;
                    if (idx > 1) c->rep[2] = c->rep[1];
                    c->rep[1] = c->rep[0];
                    c->rep[0] = (uint32_t)offset;
                }
            }
            if (lit_at + llen > regen) return ERR_CORRUPT;
            if (out + llen + mlen > cap) return ERR_ROOM;
            memcpy(dst + out, c->lit + lit_at, llen);
            out += llen;
            lit_at += llen;
            if (offset == 0 || offset > (uint64_t)out) return ERR_CORRUPT;
            for (uint32_t k = 0; k < mlen; k++, out++) dst[out] = dst[out - offset];   /* (may overlap itself) */
        }
        if (off != 0) return ERR_CORRUPT;
    } else if (ql != used) return ERR_CORRUPT;
    if (out + (regen - lit_at) > cap) return ERR_ROOM;
    memcpy(dst + out, c->lit + lit_at, (size_t)(regen - lit_at));
    return out + (regen - lit_at);
}

/* The content size a frame declares, or -1 (none declared), ERR_* (< -1 never: errors are -2 ...). */
int64_t wso_zstd_content_size(const uint8_t *src, int64_t n)
{
    if (n < 5) return ERR_TRUNCATED - 10;
    if (src[0] != 0x28 || src[1] != 0xB5 || src[2] != 0x2F || src[3] != 0xFD) return ERR_MAGIC - 10;
    const int fhd = src[4], flag = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
    int64_t pos = 5 + (single ? 0 : 1) + (did == 3 ? 4 : did);
    const int fcs = flag == 0 ? (single ? 1 : 0) : flag == 1 ? 2 : flag == 2 ? 4 : 8;
    if (pos + fcs > n) return ERR_TRUNCATED - 10;
    if (fcs == 0) return -1;
    uint64_t v = 0;
    for (int i = 0; i < fcs; i++) v |= (uint64_t)src[pos + i] << (8 * i);
    return (int64_t)(fcs == 2 ? v + 256 : v);
}

/* One frame -> its content; returns the bytes written or ERR_*.  *n_blocks (may be NULL) receives the number of blocks. */
int64_t wso_zstd_decode(const uint8_t *src, int64_t n, uint8_t *dst, int64_t cap, int32_t *n_blocks)
{
    if (n < 5) return ERR_TRUNCATED;
    if (src[0] != 0x28 || src[1] != 0xB5 || src[2] != 0x2F || src[3] != 0xFD) return ERR_MAGIC;
    const int fhd = src[4], flag = fhd >> 6, single = (fhd >> 5) & 1, cksum = (fhd >> 2) & 1, did = fhd & 3;
    if (fhd & 8) return ERR_CORRUPT;   /* reserved bit */
    if (did) return ERR_UNSUPPORTED;   /* (a dictionary) */
    int64_t pos = 5 + (single ? 0 : 1);
    pos += flag == 0 ? (single ? 1 : 0) : flag == 1 ? 2 : flag == 2 ? 4 : 8;
    if (pos > n) return ERR_TRUNCATED;
    ctx_t *c = (ctx_t *)calloc(1, sizeof(ctx_t));
    uint8_t *lit = (uint8_t *)malloc((1 << 17) + 64);
    if (!c || !lit) {
        free(c);
        free(lit);
        return ERR_ROOM;
    }
    c->lit = lit;
    c->rep[0] = 1;
    c->rep[1] = 4;
    c->rep[2] = 8;
    int64_t out = 0, rc = 0;
    int32_t blocks = 0;
    for (;;) {
        if (pos + 3 > n) { rc = ERR_TRUNCATED; break; }
        const uint32_t bh = src[pos] | (src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16);
        pos += 3;
        const int last = bh & 1, type = (bh >> 1) & 3;
        const int64_t size = bh >> 3;
        blocks++;
        if (type == 0) {
            if (pos + size > n) { rc = ERR_TRUNCATED; break; }
            if (out + size > cap) { rc = ERR_ROOM; break; }
            memcpy(dst + out, src + pos, (size_t)size);
            out += size;
            pos += size;
        } else if (type == 1) {
            if (pos + 1 > n) { rc = ERR_TRUNCATED; break; }
            if (out + size > cap) { rc = ERR_ROOM; break; }
            memset(dst + out, src[pos], (size_t)size);
            out += size;
            pos += 1;
        } else if (type == 2) {
            if (pos + size > n) { rc = ERR_TRUNCATED; break; }
            const int64_t r = block_compressed(c, src + pos, size, dst, out, cap);
            if (r < 0) { rc = r; break; }
            out = r;
            pos += size;
        } else { rc = ERR_UNSUPPORTED; break; }
        if (last) break;
    }
    if (rc == 0 && cksum && pos + 4 > n) rc = ERR_TRUNCATED;   /* (the checksum's value is not checked) */
    free(lit);
    free(c);
    if (n_blocks) *n_blocks = blocks;
    return rc ? rc : out;
}
