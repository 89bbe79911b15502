"""How much of the flank localisation rests on a tie rule (the part Biopython's absence leaves unpinned, DESIGN.md section 8
item 4): upstream-shaped input -- 110-base flanks (flank_length default) with 8-12 % basecalling errors in windows of 10-13 k
basecalled bases (extract_tr: 5 % of the read +- 5000 around the mapped location) -- through wsx_locate_flanks; a hit with
n_best_cells == 1 and tie_steps == 0 is the UNIQUE optimal local alignment (what every correct aligner returns).  For the tied
hits an independent enumeration (tests/sw_enumerate.py) tells whether the alternatives change what upstream uses: the flank's
position (start, end).    Usage: exp_flank_ties.py [pairs] [enumerate_at_most]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from warpstr_amd import extractor
from tests.test_flank_oracle import mutate
from tests.sw_enumerate import hit_from_alignment, optimal_alignments

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n_enum = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(2026)
texts, pats = [], []
for k in range(n_pairs):
    n = int(rng.integers(10000, 13001))
    read = ''.join('ACGT'[q] for q in rng.integers(0, 4, size=n))
    a = int(rng.integers(0, n - 110))
    texts.append(read)
    pats.append(mutate(rng, read[a:a + 110], float(rng.uniform(0.08, 0.12)))[:256])
t0 = time.time()
hits, _ = extractor.locate(texts, pats)
dt = time.time() - t0
ok = hits['status'] == 0
ends = hits['n_best_cells'][ok] > 1
steps = hits['tie_steps'][ok] > 0
print(f'{n_pairs} pairs (110-base flanks, 8-12 % errors, 10-13 k-base windows), {int(ok.sum())} found, {dt * 1e3:.0f} ms')
print(f'unique optimal alignment (n_best_cells == 1 and tie_steps == 0): {int((~ends & ~steps).sum())} = {100.0 * (~ends & ~steps).mean():.2f} %')
print(f'several best end cells: {int(ends.sum())} = {100.0 * ends.mean():.2f} %;  ties on the traceback: {int(steps.sum())} = {100.0 * steps.mean():.2f} %')
tied = np.flatnonzero(ok)[ends | steps][:n_enum]
moved = changed = truncated = 0
for r in tied:
    best, cells, als = optimal_alignments(texts[r], pats[r], limit=512)
    truncated += len(als) >= 512
    outs = [hit_from_alignment(texts[r], pats[r], *a[:4], a[4], best) for a in als]
    pos = {(h['start'], h['end']) for h in outs}
    used = {(h['start'], h['end'], h['score'], h['matches'], h['span']) for h in outs}  # Alignment(score, identity, position)
    moved += len(pos) > 1
    changed += len(used) > 1
    assert (int(hits['start'][r]), int(hits['end'][r])) in pos
print(f'of the first {len(tied)} tied hits, every co-optimal alignment enumerated ({truncated} stopped at 512 alignments): '
      f'{moved} have alternatives that put the flank at a different (start, end), {changed} have alternatives that change anything '
      f'upstream uses (position, corrected score, identity); for the other {len(tied) - changed} the tie rule cannot matter')
