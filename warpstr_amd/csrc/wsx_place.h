// wsx_place.h -- where the states of an automaton live in the register-resident DP fill (dtw_kernels.hip): which
// (slot, lane) holds a state ("position" = slot*64 + lane) and which LDS export slot it writes.  Plain C++ (no HIP), so the
// CPU test suite can compile and check it (tests/test_placement.py).
//
// The fill exchanges predecessor values through LDS: state x writes its export to slot w(x), a successor reads it with a
// ds_read_b64.  Measured on gfx950 (scripts/exp_ldsbank.hip, profiles/r02_lds_bank_rule.log): a ds_read_b64 serves the lanes
// 0-31 and 32-63 of a wavefront in one pass each if no two lanes of a half hit the same bank pair (slot mod 32) at different
// slots; a ds_write_b64 works on 16 lanes at a time and wants the 16 slots of lanes 16q..16q+15 distinct modulo 16.  Every
// extra slot on a bank pair costs a pass, and the LDS pipe is as loaded as the vector ALU in this kernel.  Upstream numbers the states of a repeat unit in an order that
// is not the order of the transitions (src/caller/automata.py: loops of k-mers interleave), so "state j in lane j" has
// conflicts, and moving the states with several predecessors together (they must share slot 0, see the fill) adds more.
//
// Layout: follow the FIRST-predecessor links.  They form a forest; it is cut into chains (a state is followed by the
// child with the longest tail), and the chains are laid out one after the other in LDS: w(x) = w(pred0(x)) + 1 inside a
// chain, so the 32 lanes of a half read 32 consecutive slots.  A chain that starts at a side branch ("jump": its first
// state reads a slot that is not the one before it) is moved up to the next slot that is congruent to w(pred0) + 1 modulo
// 32 while unused slots last: its read then uses exactly the bank pair its lane's neighbours leave free.  The whole
// sequence is rotated so that the states with many predecessors fall into slot 0; the ones that still do not
// trade places with the slot-0 state of the same bank pair.  For single-slot automata the lane of a state is free inside
// its half of the wavefront (writes and reads go through per-lane slot tables), which lets the states with two
// predecessors sit in lanes 0..7 (their back-pointer bits then form one byte: packed mask rows).
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

struct WsxPlacement {
    std::vector<uint16_t> pos;      // state -> position (slot*64 + lane)
    std::vector<uint16_t> state_at; // position -> state (0xFFFF = none), K*64 entries
    std::vector<uint16_t> wslot;    // position -> LDS export slot (0 .. K*64-1), K*64 entries (unused positions: own index)
    int conflict_cycles = 0;        // extra LDS cycles per DP row (reads of real predecessors + writes), all groups
    bool identity = true;           // pos[j] == j and wslot[q] == q
    bool low8 = false;              // every state with >= 2 predecessors sits in lanes 0..7 of slot 0
};

namespace wsx_place_detail {

inline int fanin(const int32_t *pp, int j) { return pp[j + 1] - pp[j]; }

// extra LDS cycles per row of a placement: for every slot k, candidate f and half g the largest number of distinct slots
// on one bank pair, minus one; the same for the export writes
inline int conflict_cycles(int S, const int32_t *pp, const int32_t *pi, int K, int F, int FL, const std::vector<uint16_t> &pos,
                           const std::vector<uint16_t> &state_at, const std::vector<uint16_t> &wslot, int *surplus_out = nullptr,
                           int *culprit_out = nullptr)
{
    // (no heap in here: the repair pass of wsx_place_attempt calls this thousands of times per automaton, and a handle for all
    // loci of a run places thousands of automata)
    struct Banks {
        int slot[32][8];
        int n[32];
        void clear(int nb) { for (int b = 0; b < nb; b++) n[b] = 0; }
        void add(int bank, int s)
        {
            for (int q = 0; q < n[bank] && q < 8; q++)
                if (slot[bank][q] == s) return;
            if (n[bank] < 8) slot[bank][n[bank]] = s;
            n[bank]++;
        }
        int worst(int nb) const
        {
            int w = 1;
            for (int b = 0; b < nb; b++) w = n[b] > w ? n[b] : w;
            return w;
        }
        int surplus(int nb) const // slots beyond the first on their bank pair, over all bank pairs
        {
            int x = 0;
            for (int b = 0; b < nb; b++) x += n[b] > 1 ? n[b] - 1 : 0;
            return x;
        }
    } on;
    int total = 0, extra = 0, culprit = -1, seen = 0;
    for (int k = 0; k < K; k++) {
        for (int q = 0; q < 4; q++) { // writes: 16 lanes at a time, slots distinct modulo 16 (idle lanes write as well)
            on.clear(16);
            for (int l = q * 16; l < q * 16 + 16; l++) {
                const int slot = wslot[k * 64 + l];
                on.add(slot & 15, slot);
            }
            total += on.worst(16) - 1;
            extra += on.surplus(16);
        }
        for (int g = 0; g < 2; g++)
            for (int f = 0; f < (k == 0 ? F : FL); f++) { // reads: 32 lanes at a time, slots distinct modulo 32
                on.clear(32);
                for (int l = g * 32; l < g * 32 + 32; l++) {
                    const int j = state_at[k * 64 + l] == 0xFFFF ? -1 : state_at[k * 64 + l];
                    if (j < 0 || fanin(pp, j) <= f) continue;
                    const int slot = wslot[pos[pi[pp[j] + f]]];
                    on.add(slot & 31, slot);
                }
                total += on.worst(32) - 1;
                extra += on.surplus(32);
                if (culprit_out && on.surplus(32) > 0) // a position whose export shares its bank pair with another one read here
                    for (int l = g * 32; l < g * 32 + 32; l++) {
                        const int j = state_at[k * 64 + l] == 0xFFFF ? -1 : state_at[k * 64 + l];
                        if (j < 0 || fanin(pp, j) <= f) continue;
                        const int q = pos[pi[pp[j] + f]];
                        if (on.n[wslot[q] & 31] <= 1) continue;
                        seen++; // (one of them, each with the same chance, by a fixed hash of its rank: deterministic)
                        if ((((unsigned)seen * 2654435761u) >> 7) % (unsigned)seen == 0) culprit = (seen & 1) ? q : k * 64 + l;
                    }
            }
    }
    (void)S;
    if (surplus_out) *surplus_out = extra;
    if (culprit_out) *culprit_out = culprit;
    return total;
}

} // namespace wsx_place_detail

// K slots of 64 lanes; states with more than FL predecessors must lie in slot 0 (FL = F: no such constraint).
// want_low8 (K == 1 only): put the states with >= 2 predecessors into lanes 0..7 if there are at most 8 of them.
inline WsxPlacement wsx_place_attempt(int S, const int32_t *pp, const int32_t *pi, int K, int F, int FL, bool want_low8, bool use_gaps)
{
    using namespace wsx_place_detail;
    const int P = K * 64;
    WsxPlacement out;
    out.pos.resize(S);
    std::iota(out.pos.begin(), out.pos.end(), (uint16_t)0);
    out.state_at.assign(P, 0xFFFF);
    for (int j = 0; j < S; j++) out.state_at[j] = (uint16_t)j;
    out.wslot.resize(P);
    std::iota(out.wslot.begin(), out.wslot.end(), (uint16_t)0);
    auto finish_identity = [&]() {
        out.identity = true;
        out.conflict_cycles = conflict_cycles(S, pp, pi, K, F, FL, out.pos, out.state_at, out.wslot);
        return out;
    };
    if (S > P || S <= 0) return finish_identity();

    // ---- first-predecessor forest, heights, chain order ------------------------------------------------------------------
    std::vector<int> pred0(S, -1), depth(S, -1);
    for (int j = 0; j < S; j++)
        if (fanin(pp, j) > 0) pred0[j] = pi[pp[j]];
    for (int j = 0; j < S; j++) { // depth by walking up; a cycle of first-predecessor links (never seen) -> identity layout
        int steps = 0, x = j;
        while (x >= 0 && depth[x] < 0 && steps <= S) {
            x = pred0[x];
            steps++;
        }
        if (steps > S) return finish_identity();
        int d = (x >= 0 ? depth[x] : -1) + steps;
        for (int y = j; y >= 0 && depth[y] < 0; y = pred0[y]) depth[y] = d--;
    }
    std::vector<std::vector<int>> kids(S);
    std::vector<int> roots;
    for (int j = 0; j < S; j++) {
        if (pred0[j] >= 0) kids[pred0[j]].push_back(j);
        else roots.push_back(j);
    }
    std::vector<int> by_depth(S), height(S, 0);
    std::iota(by_depth.begin(), by_depth.end(), 0);
    std::stable_sort(by_depth.begin(), by_depth.end(), [&](int a, int b) { return depth[a] > depth[b]; });
    for (int x : by_depth)
        if (pred0[x] >= 0) height[pred0[x]] = std::max(height[pred0[x]], height[x] + 1);
    std::vector<int> seq;
    seq.reserve(S);
    {
        std::vector<int> stack(roots.rbegin(), roots.rend());
        while (!stack.empty()) {
            int x = stack.back();
            stack.pop_back();
            // walk the chain: the child with the longest tail follows directly, the others wait on the stack
            while (true) {
                seq.push_back(x);
                auto &c = kids[x];
                if (c.empty()) break;
                int heavy = c[0];
                for (int y : c)
                    if (height[y] > height[heavy]) heavy = y;
                for (auto it = c.rbegin(); it != c.rend(); ++it)
                    if (*it != heavy) stack.push_back(*it);
                x = heavy;
            }
        }
    }
    if ((int)seq.size() != S) return finish_identity();

    // ---- LDS slots: consecutive inside a chain, jumps aligned modulo 32 while unused slots last --------------------------
    std::vector<int> w(S, -1);
    int cur = 0, slack = use_gaps ? P - S : 0;
    for (int i = 0; i < S; i++) {
        const int x = seq[i];
        const bool jump = pred0[x] >= 0 && (i == 0 || seq[i - 1] != pred0[x]);
        if (jump) {
            const int gap = (((w[pred0[x]] + 1 - cur) % 32) + 32) % 32;
            if (gap <= slack) {
                cur += gap;
                slack -= gap;
            }
        }
        w[x] = cur++;
    }
    // rotation: as many of the states that need slot 0 as possible into slots 0..63 (K > 1), or into 0..31 (low8)
    std::vector<int> need;
    for (int j = 0; j < S; j++)
        if ((K > 1 && fanin(pp, j) > FL) || (want_low8 && K == 1 && fanin(pp, j) >= 2)) need.push_back(j);
    if (!need.empty() && K > 1) {
        int best_r = 0, best_n = -1;
        for (int r = 0; r < P; r++) {
            int n = 0;
            for (int j : need) n += ((w[j] + r) % P) < 64;
            if (n > best_n) {
                best_n = n;
                best_r = r;
            }
        }
        for (int j = 0; j < S; j++) w[j] = (w[j] + best_r) % P;
    }

    std::vector<int> at_slot(P, -1); // slot -> state
    for (int j = 0; j < S; j++) at_slot[w[j]] = j;
    if (K > 1) {
        // position = slot.  A state that needs slot 0 and still lies outside trades places with the slot-0 tenant of the
        // same bank pair (both keep the banks their new neighbours leave free); failing that with any free slot-0 state.
        std::vector<char> fixed(64, 0);
        for (int j : need) {
            if (w[j] < 64) {
                fixed[w[j]] = 1;
                continue;
            }
        }
        for (int j : need) {
            if (w[j] < 64) continue;
            auto ok = [&](int c) { return !fixed[c] && (at_slot[c] < 0 || fanin(pp, at_slot[c]) <= FL); };
            int dst = -1;
            for (int c : {w[j] % 32, w[j] % 32 + 32})
                if (dst < 0 && ok(c)) dst = c;
            for (int c = 0; c < 64 && dst < 0; c++)
                if (ok(c)) dst = c;
            if (dst < 0) return finish_identity(); // more than 64 such states: the caller does not ask for this
            const int other = at_slot[dst], src = w[j];
            at_slot[dst] = j;
            at_slot[src] = other;
            w[j] = dst;
            if (other >= 0) w[other] = src;
            fixed[dst] = 1;
        }
        for (int q = 0; q < P; q++) {
            out.state_at[q] = at_slot[q] < 0 ? 0xFFFF : (uint16_t)at_slot[q];
            out.wslot[q] = (uint16_t)q;
        }
        for (int j = 0; j < S; j++) out.pos[j] = (uint16_t)w[j];
    } else {
        // one slot: LDS slot w(x) as computed; the lane is free inside a quarter of the wavefront (16 lanes: the unit of a
        // ds_write_b64).  Quarter q = slots 16q..16q+15, except that a state that has to be in the low lanes trades quarters
        // with the quarter-0 state of the same slot modulo 16 (its write keeps a bank pair of its own there).
        std::vector<int> quarter(S);
        for (int j = 0; j < S; j++) quarter[j] = w[j] / 16;
        auto is_need = [&](int j) { return std::find(need.begin(), need.end(), j) != need.end(); };
        bool low_ok = want_low8 && (int)need.size() <= 8;
        if (low_ok) {
            // two steps, each an exchange that leaves every bank rule intact: into the low HALF by trading places with the
            // state 32 slots away (reads: both halves keep their sets of bank pairs), then into quarter 0 by trading with
            // the state 16 slots away in the same half (writes: both quarters keep their sets modulo 16)
            auto tenant = [&](int qt, int mod, int val, int not_j) {
                for (int y = 0; y < S; y++)
                    if (y != not_j && quarter[y] == qt && (w[y] & (mod - 1)) == val) return y;
                return -1;
            };
            auto count = [&](int qt) { return (int)std::count(quarter.begin(), quarter.end(), qt); };
            for (int j : need) {
                if (quarter[j] >= 2) {
                    const int dst = quarter[j] - 2;
                    int u = tenant(dst, 32, w[j] & 31, j);
                    if (u < 0) u = tenant(dst ^ 1, 32, w[j] & 31, j);
                    if (u >= 0 && is_need(u)) low_ok = false;
                    if (u >= 0) std::swap(quarter[u], quarter[j]);
                    else if (count(dst) < 16) quarter[j] = dst;
                    else low_ok = false;
                }
                if (low_ok && quarter[j] == 1) {
                    const int t = tenant(0, 16, w[j] & 15, j);
                    if (t >= 0 && is_need(t)) low_ok = false;
                    if (t >= 0) std::swap(quarter[t], quarter[j]);
                    else if (count(0) < 16) quarter[j] = 0;
                    else low_ok = false;
                }
                if (!low_ok) break;
            }
        }
        if (!low_ok)
            for (int j = 0; j < S; j++) quarter[j] = w[j] / 16;
        std::vector<int> lanes[4];
        if (low_ok)
            for (int j : need) lanes[0].push_back(j);
        for (int s = 0; s < P; s++) {
            const int j = at_slot[s];
            if (j < 0 || (low_ok && is_need(j))) continue;
            lanes[quarter[j]].push_back(j);
        }
        for (auto &v : lanes)
            if (v.size() > 16) return finish_identity();
        out.state_at.assign(P, 0xFFFF);
        for (int q = 0; q < P; q++) out.wslot[q] = 0xFFFF;
        std::vector<char> slot_used(P, 0);
        for (int h = 0; h < 4; h++)
            for (size_t i = 0; i < lanes[h].size(); i++) {
                const int j = lanes[h][i], q = h * 16 + (int)i;
                out.pos[j] = (uint16_t)q;
                out.state_at[q] = (uint16_t)j;
                out.wslot[q] = (uint16_t)w[j];
                slot_used[w[j]] = 1;
            }
        // idle lanes write too (the row code has no branches): each gets a slot of its own, distinct modulo 16 in its quarter
        for (int h = 0; h < 4; h++) {
            std::vector<char> bank_used(16, 0);
            for (int l = 0; l < 16; l++)
                if (out.wslot[h * 16 + l] != 0xFFFF) bank_used[out.wslot[h * 16 + l] & 15] = 1;
            for (int l = 0; l < 16; l++) {
                const int q = h * 16 + l;
                if (out.wslot[q] != 0xFFFF) continue;
                int pick = -1;
                for (int s = 0; s < P && pick < 0; s++)
                    if (!slot_used[s] && !bank_used[s & 15]) pick = s;
                for (int s = 0; s < P && pick < 0; s++)
                    if (!slot_used[s]) pick = s;
                out.wslot[q] = (uint16_t)pick;
                slot_used[pick] = 1;
                bank_used[pick & 15] = 1;
            }
        }
        out.low8 = low_ok && !need.empty();
        if (need.empty() && want_low8) out.low8 = true; // nothing to place: trivially packed
    }
    // ---- repair: two states whose export slots share a bank pair may trade places (their writes stay conflict-free); keep
    // every trade that lowers the conflict count.  Covers what the construction above misses: jumps that found no slack,
    // states that changed halves, colliding reads of second and third predecessors.
    {
        auto pinned_ok = [&](int j, int q) { // may state j live at position q?
            if (j < 0) return true;
            if (K > 1) return fanin(pp, j) <= FL || q < 64;
            return !(out.low8 && fanin(pp, j) >= 2) || q < 8;
        };
        int cur_cost = conflict_cycles(S, pp, pi, K, F, FL, out.pos, out.state_at, out.wslot);
        for (int pass = 0; pass < 4 && cur_cost > 0; pass++) {
            bool improved = false;
            for (int a = 0; a < P && cur_cost > 0; a++)
                for (int b = a + 1; b < P && cur_cost > 0; b++) {
                    // a trade must not disturb the writes: equal slots modulo 32 (several slots: slot = position), or modulo
                    // 16 and a quarter of its own for each (one slot)
                    if (K > 1 ? ((a & 31) != (b & 31) || (a / 32) == (b / 32))
                              : ((out.wslot[a] & 15) != (out.wslot[b] & 15) || (a / 16) == (b / 16)))
                        continue;
                    const int ja = out.state_at[a] == 0xFFFF ? -1 : out.state_at[a], jb = out.state_at[b] == 0xFFFF ? -1 : out.state_at[b];
                    if ((ja < 0 && jb < 0) || !pinned_ok(ja, b) || !pinned_ok(jb, a)) continue;
                    auto trade = [&]() {
                        std::swap(out.state_at[a], out.state_at[b]);
                        if (K == 1) std::swap(out.wslot[a], out.wslot[b]);
                        if (out.state_at[a] != 0xFFFF) out.pos[out.state_at[a]] = (uint16_t)a;
                        if (out.state_at[b] != 0xFFFF) out.pos[out.state_at[b]] = (uint16_t)b;
                    };
                    trade();
                    const int c = conflict_cycles(S, pp, pi, K, F, FL, out.pos, out.state_at, out.wslot);
                    if (c < cur_cost) {
                        cur_cost = c;
                        improved = true;
                    } else {
                        trade();
                    }
                }
            if (!improved) break;
        }
        // Still above zero: a seeded walk over admissible trades that do not raise the count (the plateau is what stops the
        // descent above), descending again whenever one lowers it; deterministic.  (The count is cheap now -- no heap --, and
        // this only runs for the few automata the construction leaves with a conflict: DM2's 254-state strand had 1.0 cycles
        // per row, profiles/r04_real_loci_pmc.log.)
        if (cur_cost > 0 && K > 1) {
            // (round 5: ANY two positions may trade -- with several slots the export slot of a position is the position, the
            // writes never collide --, and among equal counts the walk prefers fewer surplus slots per bank pair: the count is the
            // worst bank pair of each read, so three colliding pairs in one read stay "1" until the last of them is resolved
            // and a walk that sees no difference between two and three of them does not get there.  configs[4]'s 127-state
            // four-candidate automaton: 1 cycle per row left by the walk before, 0 now after ~1 500 steps.)
            uint32_t rng = 0x51ED270Bu ^ (uint32_t)S;
            auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
            int cur_extra = 0, culprit = -1;
            (void)conflict_cycles(S, pp, pi, K, F, FL, out.pos, out.state_at, out.wslot, &cur_extra, &culprit);
            for (int step = 0; step < 6000 && cur_cost > 0; step++) {
                // (three steps in four move a position that is part of a collision -- a reader or the predecessor it reads, picked
                // by the count itself --, the fourth is blind: ~6 x fewer steps to zero than a blind walk, and the walk is what a
                // handle for thousands of loci spends its creation time in)
                const int a = (culprit >= 0 && (step & 3) != 3) ? culprit : (int)(next() % P), b = (int)(next() % P);
                if (a == b) continue;
                const int ja = out.state_at[a] == 0xFFFF ? -1 : out.state_at[a], jb = out.state_at[b] == 0xFFFF ? -1 : out.state_at[b];
                if ((ja < 0 && jb < 0) || !pinned_ok(ja, b) || !pinned_ok(jb, a)) continue;
                auto trade = [&]() {
                    std::swap(out.state_at[a], out.state_at[b]);
                    if (out.state_at[a] != 0xFFFF) out.pos[out.state_at[a]] = (uint16_t)a;
                    if (out.state_at[b] != 0xFFFF) out.pos[out.state_at[b]] = (uint16_t)b;
                };
                trade();
                int extra = 0, next_culprit = -1;
                const int c = conflict_cycles(S, pp, pi, K, F, FL, out.pos, out.state_at, out.wslot, &extra, &next_culprit);
                if (c < cur_cost || (c == cur_cost && extra <= cur_extra)) {
                    cur_cost = c;
                    cur_extra = extra;
                    culprit = next_culprit;
                } else {
                    trade();
                }
            }
        }
    }
    out.identity = true;
    for (int j = 0; j < S; j++) out.identity = out.identity && out.pos[j] == j;
    for (int q = 0; q < P; q++) out.identity = out.identity && out.wslot[q] == q;
    out.conflict_cycles = conflict_cycles(S, pp, pi, K, F, FL, out.pos, out.state_at, out.wslot);
    // never worse than the plain layout when that one is admissible (no slot-0 constraint to satisfy, no packing asked for)
    if ((K == 1 && !out.low8) || (K > 1 && FL >= F)) {
        WsxPlacement plain;
        plain.pos.resize(S);
        std::iota(plain.pos.begin(), plain.pos.end(), (uint16_t)0);
        plain.state_at.assign(P, 0xFFFF);
        for (int j = 0; j < S; j++) plain.state_at[j] = (uint16_t)j;
        plain.wslot.resize(P);
        std::iota(plain.wslot.begin(), plain.wslot.end(), (uint16_t)0);
        plain.conflict_cycles = conflict_cycles(S, pp, pi, K, F, FL, plain.pos, plain.state_at, plain.wslot);
        if (plain.conflict_cycles <= out.conflict_cycles) return plain;
    }
    return out;
}

// The layout with and without moving side chains up to aligned slots (the gaps can put two states that both have to be in
// the low lanes on one bank pair, or overfill a half), and the plain one where it is admissible: fewest conflict cycles
// first (the LDS pipe is what the fill runs out of), packed rows second.
inline WsxPlacement wsx_place_states(int S, const int32_t *pp, const int32_t *pi, int K, int F, int FL, bool want_low8)
{
    WsxPlacement best = wsx_place_attempt(S, pp, pi, K, F, FL, want_low8, true);
    auto better = [&](const WsxPlacement &x, const WsxPlacement &y) {
        if (x.conflict_cycles != y.conflict_cycles) return x.conflict_cycles < y.conflict_cycles;
        return want_low8 && x.low8 && !y.low8;
    };
    if (best.conflict_cycles == 0 && (best.low8 || !want_low8)) return best;
    WsxPlacement b = wsx_place_attempt(S, pp, pi, K, F, FL, want_low8, false);
    if (better(b, best)) best = b;
    if (want_low8) { // the unpacked layouts compete too
        for (bool gaps : {true, false}) {
            WsxPlacement c = wsx_place_attempt(S, pp, pi, K, F, FL, false, gaps);
            if (better(c, best)) best = c;
        }
    }
    return best;
}

// ------------------------------------------------------------------------------------------------------------------
// Lane-major placement (dtw_kernels.hip: dp_row with LM != 0).  The automata of real loci are almost chains: two long
// flanks and a few short loops.  Laid along the SLOTS of a lane -- position (slot r, lane l) holds the state after the one
// in (slot r-1, lane l) -- a state above slot 0 finds its one predecessor's value in a register of its own lane, and only
// the states in slot 0 exchange through LDS: heads of chains (states with no, or several, predecessors, and side branches),
// and the first state of every further K-state piece of a long chain, which reads the top slot of the lane before it.
//
// Chains: every state with exactly one predecessor may continue that predecessor's chain; a state keeps the child with the
// longest tail.  A chain is cut behind every state that has a successor outside the chain ("source": somebody reads it
// through LDS), so that such a state ends a piece.  A piece of L = q*K + rem states takes q full lanes; its last rem states
// either take one more lane (slots 0..rem-1: every slot may then have to export, LM = 2) or one lane EACH, in slot 0 -- then
// everything a slot-0 state reads sits in slot 0 or slot K-1 and only those two slots write to LDS (LM = 1).  The second
// form is used while 64 lanes suffice; otherwise remainders go two states to a lane (slot 1 exports too: LM = 3), and if
// that is still too many the pieces with the largest remainders change to the first form.
// Returns an empty placement when the automaton does not fit the K*64 positions this way.
// ------------------------------------------------------------------------------------------------------------------
struct WsxLanePlacement {
    WsxPlacement pl;   // pos, state_at, wslot (identity)
    int lm = 0;        // 0: does not fit; 1: slots 0 and K-1 export; 3: slots 0, 1 and K-1; 2: every slot exports;
                       // 4: stacked (wsx_place_lane_stacked): every slot exports, lanes in stack_mask read LDS in slot 2 too
    int lanes = 0;
    uint64_t stack_mask = 0; // lm = 4: lanes whose slot WSX_STACK_SLOT starts a piece of its own (one predecessor, through LDS)
};
constexpr int WSX_STACK_SLOT = 2;
inline int wsx_spread_lanes(int S, const int32_t *pp, const int32_t *pi, int K, WsxLanePlacement &lp); // bank-aware lanes, below

inline WsxLanePlacement wsx_place_lane_major(int S, const int32_t *pp, const int32_t *pi, int K)
{
    using namespace wsx_place_detail;
    WsxLanePlacement out;
    if (K < 2 || S > K * 64) return out;
    std::vector<std::vector<int>> succ(S);
    for (int j = 0; j < S; j++)
        for (int e = pp[j]; e < pp[j + 1]; e++) succ[pi[e]].push_back(j);
    // tail[j]: states in the longest run of single-predecessor states that starts at j
    std::vector<int> tail(S, 1);
    for (int it = 0; it <= S; it++) {
        bool changed = false;
        for (int j = S - 1; j >= 0; j--) {
            int best = 0;
            for (int c : succ[j])
                if (fanin(pp, c) == 1) best = std::max(best, tail[c]);
            if (1 + best != tail[j]) tail[j] = 1 + best, changed = true;
        }
        if (!changed) break;
        if (it == S) return out; // a loop of single-predecessor states: not an automaton this layout is for
    }
    std::vector<int> child(S, -1), parent(S, -1);
    for (int p = 0; p < S; p++) {
        for (int c : succ[p])
            if (fanin(pp, c) == 1 && (child[p] < 0 || tail[c] > tail[child[p]])) child[p] = c;
        if (child[p] >= 0) parent[child[p]] = p;
    }
    // pieces: maximal runs head -> child -> .. that end at a source or at the end of the chain
    std::vector<std::vector<int>> pieces;
    for (int h = 0; h < S; h++) {
        if (parent[h] >= 0) continue;
        std::vector<int> cur;
        for (int j = h; j >= 0; j = child[j]) {
            cur.push_back(j);
            bool source = false;
            for (int c : succ[j]) source |= c != child[j];
            if (source || child[j] < 0) {
                pieces.push_back(cur);
                cur.clear();
            }
            if ((int)pieces.size() > S) return out;
        }
    }
    // lanes: every piece q full lanes + rem single-state lanes (LM = 1).  Too many: the remainders go two states to a lane
    // (slots 0 and 1: slot 1 exports as well, LM = 3), largest remainders first; still too many: a remainder takes one lane
    // whatever its length (every slot may export, LM = 2).
    std::vector<char> packed(pieces.size(), 0); // 0: singles, 1: one lane for the whole remainder, 2: pairs (+ a single)
    int lanes = 0;
    for (auto &p : pieces) lanes += (int)p.size() / K + (int)p.size() % K;
    out.lm = 1;
    if (lanes > 64 && K >= 4) { // (K = 3: slots 0, 1, 2 are all there is)
        std::vector<char> pairs(pieces.size(), 0);
        int l3 = lanes;
        while (l3 > 64) {
            int best = -1;
            for (size_t q = 0; q < pieces.size(); q++)
                if (!pairs[q] && (int)pieces[q].size() % K >= 2 && (best < 0 || pieces[q].size() % K > pieces[best].size() % K))
                    best = (int)q;
            if (best < 0) break;
            pairs[best] = 2;
            l3 -= ((int)pieces[best].size() % K) / 2;
        }
        if (l3 <= 64) {
            packed = pairs;
            lanes = l3;
            out.lm = 3;
        }
    }
    while (lanes > 64) {
        int best = -1;
        for (size_t q = 0; q < pieces.size(); q++)
            if (packed[q] != 1 && (int)pieces[q].size() % K >= 2 &&
                (best < 0 || pieces[q].size() % K > pieces[best].size() % K))
                best = (int)q;
        if (best < 0) return WsxLanePlacement{};
        packed[best] = 1;
        lanes -= (int)pieces[best].size() % K - 1;
        out.lm = 2;
    }
    out.lanes = lanes;
    WsxPlacement &pl = out.pl;
    pl.pos.assign(S, 0);
    pl.state_at.assign((size_t)K * 64, 0xFFFF);
    pl.wslot.resize((size_t)K * 64);
    std::iota(pl.wslot.begin(), pl.wslot.end(), (uint16_t)0);
    int lane = 0;
    auto put = [&](int state, int slot, int l) {
        pl.pos[state] = (uint16_t)(slot * 64 + l);
        pl.state_at[slot * 64 + l] = (uint16_t)state;
    };
    for (size_t q = 0; q < pieces.size(); q++) {
        const auto &p = pieces[q];
        const int full = (int)p.size() / K, rem = (int)p.size() % K;
        for (int s = 0; s < full * K; s++) put(p[s], s % K, lane + s / K);
        lane += full;
        if (packed[q] == 1) {
            for (int s = 0; s < rem; s++) put(p[full * K + s], s, lane);
            lane += rem ? 1 : 0;
        } else if (packed[q] == 2) {
            for (int s = 0; s < rem; s++) put(p[full * K + s], s % 2, lane + s / 2); // pairs in slots 0, 1; an odd last state alone
            lane += (rem + 1) / 2;
        } else {
            for (int s = 0; s < rem; s++) put(p[full * K + s], 0, lane + s);
            lane += rem;
        }
    }
    pl.identity = false;
    pl.low8 = false;
    wsx_spread_lanes(S, pp, pi, K, out); // which lane a column takes: the one that keeps the LDS reads free of bank conflicts
    return out;
}

// ------------------------------------------------------------------------------------------------------------------
// Bank-aware lanes for the lane-major layouts.  Which LANE a column of states takes is free (a state above slot 0 finds its
// predecessor in its own lane; what a slot-0 state -- or the first state of an upper piece, LM = 4 -- reads comes through
// LDS by a per-lane address).  Position (slot k, lane l) exports to LDS slot k*64 + l, so the bank pair of a read is the
// LANE of the predecessor modulo 32, and a ds_read_b64 serves a half of the wavefront in one pass if the predecessors its 32
// lanes read sit in 32 different lanes modulo 32 (or in the very same slot).  Lanes handed out in order of the chains leave
// 1-3 extra passes per row on loci with several loops (HD, DM2 at flank 110: 3.0 measured, profiles/r03_real_loci_pmc.log);
// lane_major_read_conflicts counts them (the model of tests/test_placement.py), wsx_spread_lanes trades lanes -- whole
// columns, empty ones included -- while a trade lowers the count.
// ------------------------------------------------------------------------------------------------------------------
namespace wsx_place_detail {

struct LaneReads { // one entry per LDS read of a lane: which instruction (candidate f of slot 0, or F = the stack slot's), which position
    std::vector<std::vector<std::pair<int, int>>> of_lane; // [lane] -> (instruction, predecessor position in the UNPERMUTED layout)
    int n_instr = 0;
};

inline LaneReads lane_major_reads(int S, const int32_t *pp, const int32_t *pi, int K, const WsxPlacement &pl, uint64_t stack_mask,
                                  int stack_slot)
{
    LaneReads r;
    r.of_lane.resize(64);
    int F = 1;
    for (int j = 0; j < S; j++) F = std::max(F, fanin(pp, j));
    r.n_instr = F + 1;
    for (int l = 0; l < 64; l++) {
        const int j0 = pl.state_at[l] == 0xFFFF ? -1 : pl.state_at[l];
        if (j0 >= 0)
            for (int f = 0; f < fanin(pp, j0); f++) r.of_lane[l].push_back({f, pl.pos[pi[pp[j0] + f]]});
        if ((stack_mask >> l) & 1ull) {
            const int js = pl.state_at[stack_slot * 64 + l] == 0xFFFF ? -1 : pl.state_at[stack_slot * 64 + l];
            if (js >= 0 && fanin(pp, js) >= 1) r.of_lane[l].push_back({F, pl.pos[pi[pp[js]]]});
        }
    }
    (void)K;
    return r;
}

// extra passes per row with old lane l sitting in lane perm[l]
inline int lane_major_read_conflicts(const LaneReads &r, const int *perm, int *excess_out = nullptr)
{
    int total = 0;
    // slots[instr][half][bank]: up to a handful of distinct slots
    static thread_local std::vector<int> seen;
    seen.assign((size_t)r.n_instr * 2 * 32 * 8, -1);
    std::vector<int> cnt((size_t)r.n_instr * 2 * 32, 0);
    for (int l = 0; l < 64; l++) {
        const int half = perm[l] >> 5;
        for (auto &rd : r.of_lane[l]) {
            const int slot = (rd.second & ~63) | perm[rd.second & 63];
            const size_t cell = ((size_t)rd.first * 2 + half) * 32 + (slot & 31);
            int *sl = &seen[cell * 8];
            bool dup = false;
            for (int q = 0; q < cnt[cell] && q < 8; q++) dup = dup || sl[q] == slot;
            if (!dup) {
                if (cnt[cell] < 8) sl[cnt[cell]] = slot;
                cnt[cell]++;
            }
        }
    }
    int excess = 0; // slots beyond the first on any bank pair: what the search goes down on a plateau of `total`
    for (int i = 0; i < r.n_instr; i++)
        for (int h = 0; h < 2; h++) {
            int worst = 1;
            for (int b = 0; b < 32; b++) {
                const int n = cnt[((size_t)i * 2 + h) * 32 + b];
                worst = std::max(worst, n);
                excess += n > 1 ? n - 1 : 0;
            }
            total += worst - 1;
        }
    if (excess_out) *excess_out = excess;
    return total;
}

} // namespace wsx_place_detail

// Trades lanes of a lane-major placement (any lm) until no single trade lowers the modelled read conflicts; rewrites
// pl.pos / pl.state_at / stack_mask.  Returns the conflict cycles per row that remain (pl.conflict_cycles).
inline int wsx_spread_lanes(int S, const int32_t *pp, const int32_t *pi, int K, WsxLanePlacement &lp)
{
    using namespace wsx_place_detail;
    if (lp.lm == 0) return 0;
    WsxPlacement &pl = lp.pl;
    const LaneReads reads = lane_major_reads(S, pp, pi, K, pl, lp.stack_mask, WSX_STACK_SLOT);
    int perm[64];
    std::iota(perm, perm + 64, 0);
    int excess = 0;
    int cost = lane_major_read_conflicts(reads, perm, &excess);
    for (int pass = 0; pass < 12 && cost > 0; pass++) {
        bool improved = false;
        for (int a = 0; a < 64 && cost > 0; a++)
            for (int b = a + 1; b < 64 && cost > 0; b++) {
                std::swap(perm[a], perm[b]);
                int ex = 0;
                const int c = lane_major_read_conflicts(reads, perm, &ex);
                if (c < cost || (c == cost && ex < excess)) {
                    cost = c;
                    excess = ex;
                    improved = true;
                } else {
                    std::swap(perm[a], perm[b]);
                }
            }
        if (!improved) break;
    }
    // Stuck above zero: single trades no longer help (two sources in one column read from the same half want their readers
    // in different halves, which takes a trade that is neutral first).  A seeded walk over neutral trades, descending after
    // each; the best permutation seen is kept.  Deterministic: every process places an automaton the same way.
    if (cost > 0) {
        int best_perm[64], best_cost = cost, best_excess = excess;
        std::copy(perm, perm + 64, best_perm);
        uint32_t rng = 0x9E3779B9u ^ (uint32_t)S;
        auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
        for (int step = 0; step < 600 && best_cost > 0; step++) {
            const int a = (int)(next() % 64), b = (int)(next() % 64);
            if (a == b) continue;
            std::swap(perm[a], perm[b]);
            int ex = 0;
            int c = lane_major_read_conflicts(reads, perm, &ex);
            if (c > cost || (c == cost && ex > excess + 1)) { // worse: undo
                std::swap(perm[a], perm[b]);
                continue;
            }
            cost = c;
            excess = ex;
            for (int x = 0; x < 64 && cost > 0; x++) // one descent over the trades of the two lanes that moved
                for (int y : {a, b}) {
                    if (x == y) continue;
                    std::swap(perm[x], perm[y]);
                    int ex2 = 0;
                    const int c2 = lane_major_read_conflicts(reads, perm, &ex2);
                    if (c2 < cost || (c2 == cost && ex2 < excess)) cost = c2, excess = ex2;
                    else std::swap(perm[x], perm[y]);
                }
            if (cost < best_cost || (cost == best_cost && excess < best_excess)) {
                best_cost = cost;
                best_excess = excess;
                std::copy(perm, perm + 64, best_perm);
            }
        }
        std::copy(best_perm, best_perm + 64, perm);
        cost = best_cost;
    }
    bool moved = false;
    for (int l = 0; l < 64; l++) moved = moved || perm[l] != l;
    if (moved) {
        std::vector<uint16_t> at((size_t)K * 64, 0xFFFF);
        for (int k = 0; k < K; k++)
            for (int l = 0; l < 64; l++) at[k * 64 + perm[l]] = pl.state_at[k * 64 + l];
        pl.state_at = at;
        for (int q = 0; q < K * 64; q++)
            if (at[q] != 0xFFFF) pl.pos[at[q]] = (uint16_t)q;
        uint64_t m = 0;
        for (int l = 0; l < 64; l++)
            if ((lp.stack_mask >> l) & 1ull) m |= 1ull << perm[l];
        lp.stack_mask = m;
    }
    pl.conflict_cycles = cost;
    return cost;
}

// ------------------------------------------------------------------------------------------------------------------
// Stacked lane-major placement (LM = 4), for automata with many short chains and little room (HD, DM2 at flank 110): a
// lane may hold TWO pieces -- one in slots 0..1, one from slot WSX_STACK_SLOT (= 2) upwards.  The state that starts the upper
// piece has exactly one predecessor, somewhere else, and takes it through LDS (its own read and a per-lane select in
// dp_row: +1 LDS read, +2 vector instructions per row); every slot exports.  Chains are NOT cut behind sources here (any
// slot exports), only where the automaton branches; what is left of a chain after its full lanes (r < K states) becomes
// a lower part (<= 2 states, any head) and/or an upper part (<= K-2 states; a cut inside a chain always leaves a state
// with one predecessor).  Lanes = full lanes + max(lower parts, upper parts).
// ------------------------------------------------------------------------------------------------------------------
inline WsxLanePlacement wsx_place_lane_stacked(int S, const int32_t *pp, const int32_t *pi, int K)
{
    using namespace wsx_place_detail;
    WsxLanePlacement out;
    constexpr int SS = WSX_STACK_SLOT;
    if (K < SS + 1 || S > K * 64) return out;
    const int UP = K - SS; // capacity of an upper part
    std::vector<std::vector<int>> succ(S);
    for (int j = 0; j < S; j++)
        for (int e = pp[j]; e < pp[j + 1]; e++) succ[pi[e]].push_back(j);
    std::vector<int> tail(S, 1);
    for (int it = 0; it <= S; it++) {
        bool changed = false;
        for (int j = S - 1; j >= 0; j--) {
            int best = 0;
            for (int c : succ[j])
                if (fanin(pp, c) == 1) best = std::max(best, tail[c]);
            if (1 + best != tail[j]) tail[j] = 1 + best, changed = true;
        }
        if (!changed) break;
        if (it == S) return out;
    }
    std::vector<int> child(S, -1), parent(S, -1);
    for (int p = 0; p < S; p++) {
        for (int c : succ[p])
            if (fanin(pp, c) == 1 && (child[p] < 0 || tail[c] > tail[child[p]])) child[p] = c;
        if (child[p] >= 0) parent[child[p]] = p;
    }
    struct Part {
        std::vector<int> states;
        bool soft; // its first state has exactly one predecessor: may start at the stack slot
    };
    std::vector<std::vector<int>> full; // K states each: one lane
    std::vector<Part> lower, upper, flexible;
    for (int h = 0; h < S; h++) {
        if (parent[h] >= 0) continue;
        std::vector<int> chain;
        for (int j = h; j >= 0; j = child[j]) {
            chain.push_back(j);
            if ((int)chain.size() > S) return out;
        }
        const int q = (int)chain.size() / K, r = (int)chain.size() % K;
        for (int a = 0; a < q; a++) full.emplace_back(chain.begin() + a * K, chain.begin() + (a + 1) * K);
        if (r == 0) continue;
        std::vector<int> rem(chain.begin() + q * K, chain.end());
        const bool soft = fanin(pp, rem[0]) == 1;
        if (r <= SS && !(soft && r <= UP)) lower.push_back({rem, soft});
        else if (r <= SS || (soft && r <= UP)) {
            if (r <= SS) flexible.push_back({rem, soft}); // either half takes it
            else upper.push_back({rem, soft});
        } else { // longer than the lower half and not allowed (or too long) for the upper one: split behind slot 1
            lower.push_back({std::vector<int>(rem.begin(), rem.begin() + SS), soft});
            std::vector<int> rest(rem.begin() + SS, rem.end());
            if ((int)rest.size() > UP) return out; // (cannot happen: r <= K-1 = SS + UP - 1)
            upper.push_back({rest, true});
        }
    }
    for (auto &f : flexible) (lower.size() <= upper.size() ? lower : upper).push_back(f);
    const int lanes = (int)full.size() + (int)std::max(lower.size(), upper.size());
    if (lanes > 64) return out;
    out.lm = 4;
    out.lanes = lanes;
    WsxPlacement &pl = out.pl;
    pl.pos.assign(S, 0);
    pl.state_at.assign((size_t)K * 64, 0xFFFF);
    pl.wslot.resize((size_t)K * 64);
    std::iota(pl.wslot.begin(), pl.wslot.end(), (uint16_t)0);
    auto put = [&](int state, int slot, int l) {
        pl.pos[state] = (uint16_t)(slot * 64 + l);
        pl.state_at[slot * 64 + l] = (uint16_t)state;
    };
    int lane = 0;
    for (auto &f : full) {
        for (int s2 = 0; s2 < K; s2++) put(f[s2], s2, lane);
        lane++;
    }
    for (size_t q = 0; q < std::max(lower.size(), upper.size()); q++) {
        if (q < lower.size())
            for (size_t s2 = 0; s2 < lower[q].states.size(); s2++) put(lower[q].states[s2], (int)s2, lane);
        if (q < upper.size()) {
            for (size_t s2 = 0; s2 < upper[q].states.size(); s2++) put(upper[q].states[s2], SS + (int)s2, lane);
            out.stack_mask |= 1ull << lane;
        }
        lane++;
    }
    pl.identity = false;
    pl.low8 = false;
    wsx_spread_lanes(S, pp, pi, K, out); // which lane a column takes: the one that keeps the LDS reads free of bank conflicts
    return out;
}
