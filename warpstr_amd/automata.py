"""Locus pattern -> k-mer state automaton, as flat tables for the HIP caller.

Behavioural mirror of the reference's ``StateAutomata`` (src/caller/automata.py:36-226): same
state order, same ``incoming`` order, same ``repeat_mask``/``endstate``, so that state indices in
traces are interchangeable.  The construction here is array based (integer k-mer codes, CSR fan-in)
because its product is the table the device stages in LDS, not a Python object graph.

Pattern grammar (src/caller/automata.py:57-150):
  plain bases, IUPAC codes (parallel alternatives), ``( .. )`` one-or-more loop,
  ``{ .. }`` optional block (the base before ``{`` also links to the first plain base after ``}``).
"""
from typing import Dict, List, Optional, Tuple

import numpy as np

from .pore_model import PoreModel, default_pore_model, kmer_code

IUPAC: Dict[str, str] = {  # src/templates.py:32-44
    'R': 'AG', 'Y': 'CT', 'S': 'GC', 'W': 'AT', 'K': 'GT', 'M': 'AC',
    'B': 'CGT', 'D': 'AGT', 'H': 'ACT', 'V': 'ACG', 'N': 'ACGT',
}
STRAND_SWAP: Dict[str, str] = {  # src/templates.py:45-65
    '(': ')', ')': '(', '{': '}', '}': '{',
    'A': 'T', 'T': 'A', 'G': 'C', 'C': 'G', 'M': 'K', 'K': 'M', 'N': 'N', 'W': 'W', 'S': 'S',
    'R': 'Y', 'Y': 'R', 'B': 'V', 'D': 'H', 'H': 'D', 'V': 'B',
}
MAX_FANIN = 8


def reverse_pattern(sequence: str) -> str:
    """Pattern of the reverse strand (src/caller/wrapper.py:78-84)."""
    return ''.join(STRAND_SWAP[c] for c in reversed(sequence))


def _nucleotide_graph(pattern: str) -> Tuple[List[str], List[List[int]], int, int]:
    """One node per (expanded) base with successor lists; also first '(' / last ')' positions.

    Node numbering and successor order follow src/caller/automata.py:57-150 exactly, including its
    corner cases (an IUPAC code that opens a loop makes every alternative a loop target; optional
    blocks are only closed by the next *plain* base).
    """
    base: List[str] = [pattern[0]]
    succ: List[List[int]] = [[]]
    tails: List[int] = [0]            # nodes whose successor is the next base
    loop_heads: List[object] = []     # per open '(' : node id, or list of ids for IUPAC heads
    opt_tails: List[int] = []         # per open '{' : node before it
    pending_skips: List[int] = []     # nodes that jump over a closed '{..}'
    rep_first, rep_last = -1, -1
    nxt = 1
    for ch in pattern[1:]:
        if ch == '(':
            loop_heads.append(nxt)
            if rep_first == -1:
                rep_first = nxt
        elif ch == ')':
            rep_last = nxt
            head = loop_heads.pop()
            targets = head if isinstance(head, list) else [head]
            for t in tails:
                succ[t].extend(targets)
        elif ch == '{':
            opt_tails.append(nxt - 1)
        elif ch == '}':
            pending_skips.append(opt_tails.pop())
        elif ch in IUPAC:
            opens_loop = bool(loop_heads) and not isinstance(loop_heads[-1], list) and loop_heads[-1] == nxt
            opens_loop = opens_loop or (bool(opt_tails) and opt_tails[-1] == nxt)
            created: List[int] = []
            for alt in IUPAC[ch]:
                base.append(alt)
                succ.append([])
                created.append(nxt)
                for t in tails:
                    succ[t].append(nxt)
                nxt += 1
            if opens_loop:
                loop_heads[-1] = list(created)
            tails = created
        else:
            base.append(ch)
            succ.append([])
            for t in tails:
                succ[t].append(nxt)
            tails = [nxt]
            for t in pending_skips:
                succ[t].append(nxt)
            pending_skips = []
            nxt += 1
    return base, succ, rep_first, rep_last


class AutomatonTable:
    """Flat automaton: what the C ABI takes (include/warpstr_hip.h: wsx_automaton)."""

    def __init__(self, n_states: int, endstate: int, value: np.ndarray, seq_idx: np.ndarray, pred_ptr: np.ndarray,
                 pred_idx: np.ndarray, repeat_mask: np.ndarray, last_base: np.ndarray, kmers: Optional[List[str]] = None,
                 succ: Optional[List[List[int]]] = None, repstart: int = -1, repend: int = -1,
                 kmer_codes: Optional[np.ndarray] = None, kmersize: int = 6):
        self.n_states = n_states
        self.endstate = endstate
        self.value = value              # f64[S]  expected level per state
        self.seq_idx = seq_idx          # i32[S]  position of the state's last base in the expanded pattern
        self.pred_ptr = pred_ptr        # i32[S+1] CSR offsets into pred_idx
        self.pred_idx = pred_idx        # i32[E]  predecessors in the reference's `incoming` order
        self.repeat_mask = repeat_mask  # u8[S]
        self.last_base = last_base      # u8[S]  ASCII of the k-mer's last base
        self.repstart, self.repend = repstart, repend
        self._kmers, self._succ = kmers, succ
        self.kmer_codes, self.kmersize = kmer_codes, kmersize   # (the native compiler returns codes; the strings are made on demand)

    @property
    def kmers(self) -> List[str]:
        if self._kmers is None:
            if self.kmer_codes is None:
                return []
            k = self.kmersize
            self._kmers = [''.join('ACGT'[(int(c) >> (2 * (k - 1 - i))) & 3] for i in range(k)) for c in self.kmer_codes]
        return self._kmers

    @property
    def succ(self) -> List[List[int]]:
        """Successor states per state, ordered by target index (sources of `incoming`, transposed)."""
        if self._succ is None:
            out: List[List[int]] = [[] for _ in range(self.n_states)]
            for j in range(self.n_states):
                for p in self.pred_idx[self.pred_ptr[j]:self.pred_ptr[j + 1]]:
                    out[int(p)].append(j)
            self._succ = out
        return self._succ

    @property
    def max_fanin(self) -> int:
        return int(np.max(np.diff(self.pred_ptr))) if self.n_states else 0

    def incoming(self, j: int) -> List[int]:
        return [int(p) for p in self.pred_idx[self.pred_ptr[j]:self.pred_ptr[j + 1]]]


def compile_automaton(pattern: str, pore_model: Optional[PoreModel] = None, native: bool = True) -> AutomatonTable:
    """Build the k-mer automaton of ``left flank + locus pattern + right flank``.  native=True: by the host library
    (csrc/host_loci.cpp, the same tables 50 times faster; tests/test_host_native.py) when it is there and takes the pattern."""
    pm = pore_model or default_pore_model()
    if native:
        from . import _hostlib
        table = _hostlib.compile_automaton(pattern, pm)
        if table is not None:
            return table
    k = pm.kmersize
    base, succ, rep_first, rep_last = _nucleotide_graph(pattern)
    n_nodes = len(base)
    if n_nodes < k:
        raise ValueError('pattern shorter than the pore model k-mer size')

    # k-mer states, bucketed by the node of their last base (src/caller/automata.py:152-195).
    # state record: [kmer, node, came_from_node, successors(list of (node, slot))]
    buckets: List[List[list]] = [[] for _ in range(n_nodes)]
    first = ''.join(base[:k])
    root = [first, k - 1, -1, []]
    buckets[k - 1].append(root)
    stack = [root]
    while stack:
        cur = stack.pop()
        tail = cur[0][1:]
        for node in succ[cur[1]]:
            kmer = tail + base[node]
            linked = False
            for slot, other in enumerate(buckets[node]):
                if other[0] == kmer and other[2] == cur[1]:
                    cur[3].append((node, slot))
                    linked = True
            if not linked:
                new = [kmer, node, cur[1], []]
                cur[3].append((node, len(buckets[node])))
                buckets[node].append(new)
                stack.append(new)

    # flatten in node order (src/caller/automata.py:197-226)
    base_of = np.zeros(n_nodes + 1, dtype=np.int64)
    for n in range(n_nodes):
        base_of[n + 1] = base_of[n] + len(buckets[n])
    n_states = int(base_of[-1])
    kmers: List[str] = []
    seq_idx = np.zeros(n_states, dtype=np.int32)
    value = np.zeros(n_states, dtype=np.float64)
    rmask = np.zeros(n_states, dtype=np.uint8)
    last_base = np.zeros(n_states, dtype=np.uint8)
    succ_states: List[List[int]] = []
    for n in range(n_nodes):
        for st in buckets[n]:
            j = len(kmers)
            kmers.append(st[0])
            seq_idx[j] = n
            value[j] = pm.level_norm[kmer_code(st[0])]
            rmask[j] = 1 if (rep_first - 1 <= n <= rep_last + 10) else 0
            last_base[j] = ord(st[0][-1])
            succ_states.append([int(base_of[node] + slot) for node, slot in st[3]])
    endstate = int(base_of[n_nodes]) - 1 if buckets[-1] else -1

    incoming: List[List[int]] = [[] for _ in range(n_states)]
    for j in range(n_states):
        for t in succ_states[j]:
            incoming[t].append(j)
    pred_ptr = np.zeros(n_states + 1, dtype=np.int32)
    for j in range(n_states):
        pred_ptr[j + 1] = pred_ptr[j] + len(incoming[j])
    pred_idx = np.array([p for lst in incoming for p in lst], dtype=np.int32)
    return AutomatonTable(n_states=n_states, endstate=endstate, value=value, seq_idx=seq_idx,
                          pred_ptr=pred_ptr, pred_idx=pred_idx, repeat_mask=rmask, last_base=last_base,
                          kmers=kmers, succ=succ_states, repstart=rep_first, repend=rep_last)


def locus_automata(left_t: str, right_t: str, left_r: str, right_r: str, sequence: str,
                   pore_model: Optional[PoreModel] = None) -> Tuple[AutomatonTable, AutomatonTable]:
    """Template and reverse automata of a locus (src/caller/wrapper.py:63-76)."""
    tmp = left_t + sequence + right_t
    rev = left_r + reverse_pattern(sequence) + right_r
    return compile_automaton(tmp, pore_model), compile_automaton(rev, pore_model)
