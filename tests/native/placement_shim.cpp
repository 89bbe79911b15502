// Test shim: exposes warpstr_amd/csrc/wsx_place.h (host-only C++) to the CPU test suite through a C function.
#include "../../warpstr_amd/csrc/wsx_place.h"

extern "C" int wsx_test_place(int S, const int32_t *pred_ptr, const int32_t *pred_idx, int K, int F, int FL, int want_low8,
                              uint16_t *pos, uint16_t *state_at, uint16_t *wslot, int *low8, int *identity, int *plain_conflicts)
{
    const WsxPlacement p = wsx_place_states(S, pred_ptr, pred_idx, K, F, FL, want_low8 != 0);
    for (int j = 0; j < S; j++) pos[j] = p.pos[j];
    for (int q = 0; q < K * 64; q++) {
        state_at[q] = p.state_at[q];
        wslot[q] = p.wslot[q];
    }
    *low8 = p.low8;
    *identity = p.identity;
    std::vector<uint16_t> ipos(S), iat(K * 64, 0xFFFF), iw(K * 64);
    for (int j = 0; j < S; j++) ipos[j] = iat[j] = (uint16_t)j;
    for (int q = 0; q < K * 64; q++) iw[q] = (uint16_t)q;
    *plain_conflicts = wsx_place_detail::conflict_cycles(S, pred_ptr, pred_idx, K, F, F, ipos, iat, iw);
    return p.conflict_cycles;
}

// lane-major placement: returns lm (0 = does not fit, 1 = slots 0 and K-1 export, 2 = every slot), fills pos / state_at
extern "C" int wsx_test_place_lane_major(int S, const int32_t *pred_ptr, const int32_t *pred_idx, int K, uint16_t *pos,
                                         uint16_t *state_at, int *lanes)
{
    const WsxLanePlacement p = wsx_place_lane_major(S, pred_ptr, pred_idx, K);
    *lanes = p.lanes;
    if (p.lm == 0) return 0;
    for (int j = 0; j < S; j++) pos[j] = p.pl.pos[j];
    for (int q = 0; q < K * 64; q++) state_at[q] = p.pl.state_at[q];
    return p.lm;
}

// the read-conflict cycles per row the placement code itself counts for its last lane-major result (0 = none left)
extern "C" int wsx_test_lane_conflicts(int S, const int32_t *pred_ptr, const int32_t *pred_idx, int K, int stacked)
{
    const WsxLanePlacement p = stacked ? wsx_place_lane_stacked(S, pred_ptr, pred_idx, K) : wsx_place_lane_major(S, pred_ptr, pred_idx, K);
    return p.lm == 0 ? -1 : p.pl.conflict_cycles;
}

extern "C" int wsx_test_place_lane_stacked(int S, const int32_t *pred_ptr, const int32_t *pred_idx, int K, uint16_t *pos,
                                           uint16_t *state_at, int *lanes, uint64_t *stack_mask)
{
    const WsxLanePlacement p = wsx_place_lane_stacked(S, pred_ptr, pred_idx, K);
    *lanes = p.lanes;
    *stack_mask = p.stack_mask;
    if (p.lm == 0) return 0;
    for (int j = 0; j < S; j++) pos[j] = p.pl.pos[j];
    for (int q = 0; q < K * 64; q++) state_at[q] = p.pl.state_at[q];
    return p.lm;
}
