"""Fixture generator: approximate flanks of the upstream test locus from the upstream test BAM.

The upstream test case (README "Running the test case": 10 reads of locus Human_STR_1108232, chr4:183178378-183178421,
pattern (AAAT)) takes its flanks from GRCh38, which is not available offline.  The 10 basecalled reads in
`test/test_input/test_run1/mapping/mapping.bam` cover the locus, so a pile-up consensus of their alignments restates
the 110 reference bases either side of it well enough to call the reads (flanks only anchor the warping).  The BAM is
read with a few lines of struct/gzip code (no pysam offline).  Output: flanks.json next to this script.

Run (in the build container only):  python tests/golden/real/make_flanks.py
"""
import collections
import gzip
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
BAM = '/root/reference/test/test_input/test_run1/mapping/mapping.bam'
CHROM, START, END = 'chr4', 183178378, 183178421  # test/config_template.yaml (1-based, inclusive)
FLANK = 110
COMP = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', 'N': 'N'}


def read_bam(path):
    data = gzip.open(path, 'rb').read()  # BGZF is a multi-member gzip stream
    assert data[:4] == b'BAM\1'
    p = 4
    l_text, = struct.unpack_from('<i', data, p)
    p += 4 + l_text
    n_ref, = struct.unpack_from('<i', data, p)
    p += 4
    refs = []
    for _ in range(n_ref):
        ln, = struct.unpack_from('<i', data, p)
        refs.append(data[p + 4:p + 4 + ln - 1].decode())
        p += 4 + ln + 4
    while p < len(data):
        size, = struct.unpack_from('<i', data, p)
        p += 4
        ref_id, pos, l_name, _mapq, _bin, n_cig, flag, l_seq = struct.unpack_from('<iiBBHHHi', data, p)
        q = p + 32
        name = data[q:q + l_name - 1].decode()
        q += l_name
        cigar = [(c & 15, c >> 4) for c in struct.unpack_from('<%dI' % n_cig, data, q)]
        q += 4 * n_cig
        packed = data[q:q + (l_seq + 1) // 2]
        seq = ''.join('=ACMGRSVTWYHKDBN'[b >> 4] + '=ACMGRSVTWYHKDBN'[b & 15] for b in packed)[:l_seq]
        yield name, refs[ref_id] if ref_id >= 0 else None, pos, flag, cigar, seq
        p += size


def consensus(lo, hi):
    """Majority base (or deletion) per reference position lo..hi (1-based inclusive), plus majority insertions."""
    col = collections.defaultdict(collections.Counter)
    ins = collections.defaultdict(collections.Counter)
    depth = collections.Counter()
    for _name, chrom, pos0, _flag, cigar, seq in read_bam(BAM):
        if chrom != CHROM:
            continue
        r, q = pos0 + 1, 0
        for op, n in cigar:
            if op in (0, 7, 8):  # M = X
                for k in range(n):
                    if lo <= r + k <= hi:
                        col[r + k][seq[q + k]] += 1
                        depth[r + k] += 1
                r += n
                q += n
            elif op == 1:  # I: attached to the reference base before it
                if lo <= r - 1 <= hi:
                    ins[r - 1][seq[q:q + n]] += 1
                q += n
            elif op in (2, 3):  # D N
                for k in range(n):
                    if lo <= r + k <= hi:
                        col[r + k]['-'] += 1
                        depth[r + k] += 1
                r += n
            elif op == 4:  # S
                q += n
    out = []
    for r in range(lo, hi + 1):
        base, _ = col[r].most_common(1)[0]
        if base != '-':
            out.append(base)
        if ins[r]:
            s, c = ins[r].most_common(1)[0]
            if 2 * c > depth[r]:
                out.append(s)
    return ''.join(out), min(depth[r] for r in range(lo, hi + 1))


def revcomp(s):
    return ''.join(COMP[c] for c in reversed(s))


def main():
    left, d1 = consensus(START - FLANK, START - 1)
    right, d2 = consensus(END + 1, END + FLANK)
    left, right = left[-FLANK:].rjust(FLANK, 'N'), right[:FLANK].ljust(FLANK, 'N')
    middle, _ = consensus(START, END)
    doc = {'source': 'pile-up consensus of test/test_input/test_run1/mapping/mapping.bam (not GRCh38)',
           'coord': f'{CHROM}:{START}-{END}', 'flank_length': FLANK, 'sequence': '(AAAT)', 'min_depth': min(d1, d2),
           'left_template': left, 'right_template': right,
           'left_reverse': revcomp(right), 'right_reverse': revcomp(left), 'consensus_repeat': middle}
    with open(os.path.join(HERE, 'flanks.json'), 'w') as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == '__main__':
    main()
