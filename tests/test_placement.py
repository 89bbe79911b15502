"""The state placement of the register-resident fill (warpstr_amd/csrc/wsx_place.h, host-only C++): compiled here with
g++ and checked for validity (a permutation; states with many predecessors in slot 0 / lanes 0..7; distinct LDS slots)
and for what it is for -- LDS bank conflicts per DP row, counted by an independent Python model."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from warpstr_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp('place') / 'libplace.so')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-Wall', '-Werror', '-shared', '-fPIC', '-o', so,
                           os.path.join(ROOT, 'tests', 'native', 'placement_shim.cpp')])
    return C.CDLL(so)


def place(lib, t, K, F, FL, low8):
    S = t.n_states
    pp, pi = np.ascontiguousarray(t.pred_ptr, np.int32), np.ascontiguousarray(t.pred_idx, np.int32)
    pos, sa, ws = np.zeros(S, np.uint16), np.zeros(K * 64, np.uint16), np.zeros(K * 64, np.uint16)
    l8, idn, plain = C.c_int(), C.c_int(), C.c_int()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    c = lib.wsx_test_place(S, p(pp), p(pi), K, F, FL, int(low8), p(pos), p(sa), p(ws), C.byref(l8), C.byref(idn), C.byref(plain))
    return c, pos, sa, ws, bool(l8.value), plain.value


def model_conflicts(t, K, F, FL, pos, sa, ws):
    """Extra LDS passes per row under the rule measured on gfx950 (scripts/exp_ldsbank.hip, profiles/r02_lds_bank_rule.log):
    ds_read_b64 -- per slot, candidate and half-wave the largest number of distinct export slots that share a bank pair
    (slot mod 32), minus one; ds_write_b64 -- the same per 16 lanes with slots taken modulo 16 (idle lanes write too)."""
    pp, pi = t.pred_ptr, t.pred_idx
    total = 0
    for k in range(K):
        for q in range(4):
            on = {}
            for lane in range(q * 16, q * 16 + 16):
                slot = int(ws[k * 64 + lane])
                on.setdefault(slot & 15, set()).add(slot)
            total += max(len(v) for v in on.values()) - 1
        for g in range(2):
            for f in range(F if k == 0 else FL):
                on = {}
                for lane in range(g * 32, g * 32 + 32):
                    j = int(sa[k * 64 + lane])
                    if j != 0xFFFF and pp[j + 1] - pp[j] > f:
                        slot = int(ws[pos[pi[pp[j] + f]]])
                        on.setdefault(slot & 31, set()).add(slot)
                total += max([len(v) for v in on.values()] or [1]) - 1
    return total


CASES = [('(AAAT)', 110, 1, None), ('(AGC)AACAGCCGCCAC(CGC)', 19, 2024, 64), ('(AGC)', 16, 11, None), ('(GGCCCC)', 61, 13, None),
         ('((CAGG){CAGM})(CAGA)(CA)', 40, 11, None), ('(CAG)CAACAG(CCG)', 54, 12, None), ('(NGC)', 24, 1, None),
         ('(CAG)', 30, 5, None), ('(CTG)CTA(CTG)', 18, 9, None), ('(GAA)', 61, 17, None)]


@pytest.mark.parametrize('pattern,fl,seed,max_states', CASES)
def test_placement_is_valid_and_not_worse_than_the_natural_order(shim, pattern, fl, seed, max_states):
    locus = synth.make_locus(pattern, fl, seed, max_states=max_states)
    for t in (locus.template, locus.reverse):
        S = t.n_states
        K = (S + 63) // 64
        fan = np.diff(t.pred_ptr)
        F = max(2, int(fan.max()))
        FL = F
        if K > 1:  # as wsx_caller_create chooses it
            FL = 1 if (fan >= 2).sum() <= 64 else (2 if F > 2 and (fan > 2).sum() <= 64 else F)
        c, pos, sa, ws, low8, plain = place(shim, t, K, F, FL, K == 1 and F == 2)
        assert len(set(pos.tolist())) == S and all(sa[pos[j]] == j for j in range(S))
        assert (sa != 0xFFFF).sum() == S
        assert len(set(ws.tolist())) == K * 64 and ws.max() < K * 64          # every lane owns an export slot
        if K > 1:
            assert all(ws[q] == q for q in range(K * 64))                       # several slots: export slot = position
            assert all(pos[j] < 64 for j in range(S) if fan[j] > FL)            # many predecessors: slot 0
        if low8:
            assert all(pos[j] < 8 for j in range(S) if fan[j] >= 2)             # packed rows: their bits form one byte
        assert c == model_conflicts(t, K, F, FL, pos, sa, ws)
        assert model_conflicts(t, K, F, FL, pos, sa, ws) <= 3
        if FL == F:
            assert c <= plain


def test_headline_and_upstream_test_case_are_conflict_free(shim):
    """configs[2] (packed rows) and the (AAAT) flank-110 automaton of the upstream test case: no bank conflict at all."""
    head = synth.make_locus('(AGC)AACAGCCGCCAC(CGC)', 19, 2024, max_states=64)
    for t in (head.template, head.reverse):
        c, pos, sa, ws, low8, _ = place(shim, t, 1, 2, 2, True)
        assert c == 0 and low8
    aaat = synth.make_locus('(AAAT)', 110, 1)
    for t in (aaat.template, aaat.reverse):
        c, *_ = place(shim, t, 4, 2, 1, False)
        assert c == 0


def test_random_loci_keep_the_invariants(shim):
    """Seeded random locus patterns (nested loops, optional blocks, IUPAC codes, interruptions, flanks 12..150): whatever
    the conflict count, the result is a valid placement and the reported count is the model's."""
    rng = np.random.default_rng(0)
    units = ['AGC', 'AAAT', 'GGCCCC', 'CAG', 'CTG', 'CCTG', 'NGC', 'RY', 'CAGM', 'AAGGG', 'GAA', 'TTTTA', 'GCN', 'CGG']
    placed, free = 0, 0
    for _ in range(40):
        pat = ''
        for _u in range(int(rng.integers(1, 4))):
            unit = units[int(rng.integers(len(units)))]
            if rng.random() < 0.2:
                unit = '(' + unit + '){' + units[int(rng.integers(len(units)))] + '}'
            pat += '(' + unit + ')'
            if rng.random() < 0.4:
                pat += ''.join('ACGT'[i] for i in rng.integers(0, 4, size=int(rng.integers(1, 14))))
        locus = synth.make_locus(pat, int(rng.integers(12, 150)), int(rng.integers(1_000_000)))
        for t in (locus.template, locus.reverse):
            S = t.n_states
            K = (S + 63) // 64
            fan = np.diff(t.pred_ptr)
            F = max(2, int(fan.max()))
            if K > 5 or F > 4:
                continue
            FL = F
            if K > 1:
                FL = 1 if (fan >= 2).sum() <= 64 else (2 if F > 2 and (fan > 2).sum() <= 64 else F)
            c, pos, sa, ws, low8, _ = place(shim, t, K, F, FL, K == 1 and F == 2)
            assert len(set(pos.tolist())) == S and all(sa[pos[j]] == j for j in range(S)), pat
            assert len(set(ws.tolist())) == K * 64 and ws.max() < K * 64, pat
            assert K == 1 or all(pos[j] < 64 for j in range(S) if fan[j] > FL), pat
            assert not low8 or all(pos[j] < 8 for j in range(S) if fan[j] >= 2), pat
            assert c == model_conflicts(t, K, F, FL, pos, sa, ws), pat
            placed += 1
            free += c == 0
    assert placed >= 40 and free >= placed // 2


def place_lane_major(lib, t, K):
    S = t.n_states
    pp, pi = np.ascontiguousarray(t.pred_ptr, np.int32), np.ascontiguousarray(t.pred_idx, np.int32)
    pos, sa = np.zeros(S, np.uint16), np.zeros(K * 64, np.uint16)
    lanes = C.c_int()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lm = lib.wsx_test_place_lane_major(S, p(pp), p(pi), K, p(pos), p(sa), C.byref(lanes))
    return lm, pos, sa, lanes.value


def check_lane_major(t, K, lm, pos, sa):
    """The contract dp_row<.., LM> relies on (dtw_kernels.hip): a permutation; a state above slot 0 has exactly one predecessor
    and it sits right below it in the same lane; what a slot-0 state reads through LDS sits in an exporting slot."""
    S = t.n_states
    assert len(set(int(x) for x in pos)) == S and all(int(sa[int(pos[j])]) == j for j in range(S))
    assert sum(1 for q in sa if q != 0xFFFF) == S
    for j in range(S):
        slot, lane = int(pos[j]) // 64, int(pos[j]) % 64
        inc = t.incoming(j)
        if slot > 0:
            assert len(inc) == 1 and int(pos[inc[0]]) == (slot - 1) * 64 + lane, (j, slot, lane, inc)
        elif lm in (1, 3):
            allowed = (0, K - 1) if lm == 1 else (0, 1, K - 1)
            assert all(int(pos[p]) // 64 in allowed for p in inc), (j, inc)
    for lane in range(64):  # a lane is filled from slot 0 upwards
        filled = [sa[k * 64 + lane] != 0xFFFF for k in range(K)]
        assert filled == sorted(filled, reverse=True)


@pytest.mark.parametrize('pattern,fl,seed', [('(AAAT)', 110, 1), ('(AGC)', 110, 3), ('(GGCCCC)', 110, 2), ('(AGC)', 126, 5),
                                             ('(CCTG)(TCTG)', 110, 4), ('(AAGGG)(AAAGG)', 110, 6), ('(AGC)', 40, 1)])
def test_lane_major_placement_of_simple_loci(shim, pattern, fl, seed):
    """Flank-110 automata of simple repeats fit the lane-major layout (chains along the slots of a lane)."""
    locus = synth.make_locus(pattern, fl, seed)
    for t in (locus.template, locus.reverse):
        K = (t.n_states + 63) // 64
        lm, pos, sa, lanes = place_lane_major(shim, t, K)
        assert lm in (1, 2, 3) and lanes <= 64, (pattern, t.n_states, lm, lanes)
        check_lane_major(t, K, lm, pos, sa)


def test_lane_major_placement_on_random_loci(shim):
    """Whatever the automaton: either the layout is refused, or it keeps the contract."""
    rng = np.random.default_rng(5)
    units = ['AGC', 'AAAT', 'GGCCCC', 'CAG', 'CCTG', 'NGC', 'RY', 'CAGM', 'AAGGG', 'GAA']
    fits = 0
    for it in range(60):
        pat = ''.join('(' + units[int(rng.integers(len(units)))] + ')' + ''.join('ACGT'[i] for i in rng.integers(0, 4, size=int(rng.integers(0, 6))))
                      for _ in range(int(rng.integers(1, 4))))
        locus = synth.make_locus(pat, int(rng.integers(30, 150)), int(rng.integers(1 << 20)))
        for t in (locus.template, locus.reverse):
            K = (t.n_states + 63) // 64
            if K < 2 or K > 5:
                continue
            lm, pos, sa, lanes = place_lane_major(shim, t, K)
            if lm:
                fits += 1
                check_lane_major(t, K, lm, pos, sa)
    assert fits >= 20


def place_lane_stacked(lib, t, K):
    S = t.n_states
    pp, pi = np.ascontiguousarray(t.pred_ptr, np.int32), np.ascontiguousarray(t.pred_idx, np.int32)
    pos, sa = np.zeros(S, np.uint16), np.zeros(K * 64, np.uint16)
    lanes, mask = C.c_int(), C.c_uint64()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lm = lib.wsx_test_place_lane_stacked(S, p(pp), p(pi), K, p(pos), p(sa), C.byref(lanes), C.byref(mask))
    return lm, pos, sa, lanes.value, mask.value


def check_lane_stacked(t, K, pos, sa, mask):
    """The contract dp_row<.., LM = 4> relies on: a permutation; a state above slot 0 has exactly one predecessor -- right
    below it in the same lane, or anywhere if it sits in the stack slot (2) of a lane of the stack mask; slot-0 states may
    have any predecessors (every slot exports)."""
    S = t.n_states
    assert len(set(int(x) for x in pos)) == S and all(int(sa[int(pos[j])]) == j for j in range(S))
    assert sum(1 for q in sa if q != 0xFFFF) == S
    for j in range(S):
        slot, lane = int(pos[j]) // 64, int(pos[j]) % 64
        inc = t.incoming(j)
        if slot == 2 and (mask >> lane) & 1:
            assert len(inc) == 1, (j, inc)
        elif slot > 0:
            assert len(inc) == 1 and int(pos[inc[0]]) == (slot - 1) * 64 + lane, (j, slot, lane, inc)
    for lane in range(64):
        if (mask >> lane) & 1:
            assert sa[2 * 64 + lane] != 0xFFFF  # a stacked lane has a state in the stack slot


@pytest.mark.parametrize('pattern,name', [('(AGC)AACAGCCGCCAC(CGC)', 'HD'), ('((CAGG){CAGM})(CAGA)(CA)', 'DM2')])
def test_stacked_lane_major_takes_the_complex_example_loci(shim, pattern, name):
    """HD and DM2 (upstream's example configuration) at flank 110: many short chains and little room -- the plain lane-major
    layout refuses at least one strand of each; two pieces to a lane fit both strands of HD in four slots and DM2's
    266-state strand in five.  (The caller uses the stacked layout for five slots only: with four it measured slower than
    the slot-major kernel, profiles/r03_stacked_ab.log.)"""
    locus = synth.make_locus(pattern, 110, 7)
    plain_refused = fitted = 0
    for t in (locus.template, locus.reverse):
        K = (t.n_states + 63) // 64
        lm, _, _, _ = place_lane_major(shim, t, K)
        plain_refused += lm == 0
        lm4, pos, sa, lanes, mask = place_lane_stacked(shim, t, K)
        if lm4:
            assert lm4 == 4 and lanes <= 64
            check_lane_stacked(t, K, pos, sa, mask)
            fitted += 1
    assert plain_refused >= 1 and fitted >= (2 if name == 'HD' else 1)


def test_stacked_lane_major_on_random_loci(shim):
    rng = np.random.default_rng(9)
    units = ['AGC', 'AAAT', 'GGCCCC', 'CAG', 'CCTG', 'NGC', 'RY', 'CAGM', 'AAGGG', 'GAA']
    fits = 0
    for it in range(80):
        pat = ''.join('(' + units[int(rng.integers(len(units)))] + ')' + ''.join('ACGT'[i] for i in rng.integers(0, 4, size=int(rng.integers(0, 6))))
                      for _ in range(int(rng.integers(1, 4))))
        locus = synth.make_locus(pat, int(rng.integers(60, 150)), int(rng.integers(1 << 20)))
        for t in (locus.template, locus.reverse):
            K = (t.n_states + 63) // 64
            if K < 4 or K > 5:
                continue
            lm, pos, sa, lanes, mask = place_lane_stacked(shim, t, K)
            if lm:
                fits += 1
                check_lane_stacked(t, K, pos, sa, mask)
    assert fits >= 20



def model_lane_read_conflicts(t, K, pos, sa, mask, stacked):
    """Extra ds_read_b64 passes per row of a lane-major placement, by the rule of model_conflicts: position (slot k, lane l)
    exports to LDS slot k*64 + l, so a read's bank pair is the predecessor's LANE modulo 32.  LDS is read by the slot-0
    states (one instruction per candidate) and, in the stacked layout, by the first state of an upper piece (slot 2 of the
    lanes of the stack mask: one more instruction); per instruction and half of the wavefront the largest number of distinct
    slots on one bank pair, minus one."""
    F = max(2, int(np.diff(t.pred_ptr).max()))
    total = 0
    for k, cands in [(0, range(F))] + ([(2, [0])] if stacked else []):
        for f in cands:
            for g in range(2):
                on = {}
                for lane in range(g * 32, g * 32 + 32):
                    j = int(sa[k * 64 + lane])
                    if j == 0xFFFF or (k == 2 and not (mask >> lane) & 1):
                        continue
                    inc = t.incoming(j)
                    if len(inc) > f:
                        p = int(pos[inc[f]])
                        on.setdefault(p & 31, set()).add(p)
                total += max([len(v) for v in on.values()] or [1]) - 1
    return total


@pytest.mark.parametrize('pattern', ['(AAAT)', '(AGC)AACAGCCGCCAC(CGC)', '((CAGG){CAGM})(CAGA)(CA)', '(AGC)', '(GGCCCC)', '(CCTG)(TCTG)',
                                     '(AAGGG)(AAAGG)'])
def test_lane_major_layouts_of_the_flank_110_loci_read_lds_without_bank_conflicts(shim, pattern):
    """Upstream's own loci at the default flank (HD and DM2 of example/config.yaml among them; their stacked layouts ran 3
    conflict cycles per row in round 3, profiles/r03_real_loci_pmc.log): whichever lane-major layout the caller takes -- the
    plain one where it fits, else the stacked one -- its lanes are chosen so that no LDS read has a bank conflict, by this
    file's own model; and the count the placement code reports is the model's."""
    locus = synth.make_locus(pattern, 110, 7)
    p = lambda a: np.ascontiguousarray(a, np.int32).ctypes.data_as(C.c_void_p)
    for t in (locus.template, locus.reverse):
        K = (t.n_states + 63) // 64
        lm, pos, sa, lanes = place_lane_major(shim, t, K)
        if lm:
            check_lane_major(t, K, lm, pos, sa)
            assert model_lane_read_conflicts(t, K, pos, sa, 0, False) == 0, (pattern, t.n_states)
            assert shim.wsx_test_lane_conflicts(t.n_states, p(t.pred_ptr), p(t.pred_idx), K, 0) == 0
        else:
            lm4, pos, sa, lanes, mask = place_lane_stacked(shim, t, K)
            if lm4 == 0:  # DM2's 254-state strand fits neither way in four slots: it keeps the slot-major kernel
                assert pattern.startswith('((CAGG)') and K == 4
                continue
            check_lane_stacked(t, K, pos, sa, mask)
            assert model_lane_read_conflicts(t, K, pos, sa, mask, True) == 0, (pattern, t.n_states)
            assert shim.wsx_test_lane_conflicts(t.n_states, p(t.pred_ptr), p(t.pred_idx), K, 1) == 0


def test_lane_conflict_count_is_the_models_on_random_loci(shim):
    rng = np.random.default_rng(21)
    units = ['AGC', 'AAAT', 'GGCCCC', 'CAG', 'CCTG', 'NGC', 'RY', 'CAGM', 'AAGGG', 'GAA']
    seen = free = 0
    p = lambda a: np.ascontiguousarray(a, np.int32).ctypes.data_as(C.c_void_p)
    for it in range(60):
        pat = ''.join('(' + units[int(rng.integers(len(units)))] + ')' + ''.join('ACGT'[i] for i in rng.integers(0, 4, size=int(rng.integers(0, 6))))
                      for _ in range(int(rng.integers(1, 4))))
        locus = synth.make_locus(pat, int(rng.integers(60, 150)), int(rng.integers(1 << 20)))
        for t in (locus.template, locus.reverse):
            K = (t.n_states + 63) // 64
            if K < 3 or K > 5:
                continue
            lm, pos, sa, lanes = place_lane_major(shim, t, K)
            if lm:
                c = model_lane_read_conflicts(t, K, pos, sa, 0, False)
                assert c == shim.wsx_test_lane_conflicts(t.n_states, p(t.pred_ptr), p(t.pred_idx), K, 0), pat
                seen += 1
                free += c == 0
            lm4, pos, sa, lanes, mask = place_lane_stacked(shim, t, K) if K >= 4 else (0, 0, 0, 0, 0)
            if lm4:
                c = model_lane_read_conflicts(t, K, pos, sa, mask, True)
                assert c == shim.wsx_test_lane_conflicts(t.n_states, p(t.pred_ptr), p(t.pred_idx), K, 1), pat
                seen += 1
                free += c == 0
    assert seen >= 40 and free >= seen * 3 // 4
