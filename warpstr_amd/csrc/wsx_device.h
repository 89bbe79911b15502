// wsx_device.h -- device-side data layout shared by the HIP kernels and the C-ABI host code.
//
// HBM layout (one "chunk" = a contiguous range of reads of the batch, sized to the workspace limit):
//   per-sample arrays are indexed by   loff = offsets[r] - base_off      (signal, rescaled, trace, run lists,
//                                                                         alignment records)
//   per-read arrays by                 lr   = r - first_read
//   packed per-sample bit masks by     loff/32 + lr                       (32 samples per word, one spare word/read)
//   DP back-pointer scratch by         bp_off[lr]                         (64-bit words from the start of the chunk's region: the reads lie
//                                                                         back to back, each with the rows its kernel variant writes)
// so no prefix sums are needed besides the caller's own offsets[] and that one array.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WSX_WAVE 64
#define WSX_MAX_K 5      // states per lane in the register-resident DP kernel (S <= 320)
#define WSX_MAX_F 4
#define WSX_MAX_STREAMS 8 // chunks of a batch rotate over this many HIP streams      // fan-in handled by the register-resident DP kernel

// Launch-policy knobs of a handle (wsx_caller_set_tuning, include/warpstr_hip.h); the defaults are the measured ones.
struct WsxTuning {
    int32_t stream_traceback_min = 8192; // smallest single-slot launch that takes the thread-per-read traceback
    int32_t borders_wave_below = 8192;   // launches of fewer reads take the wave-per-read form of the borders stage
    int32_t segment_two_kernels = 0;     // 1: always the two-kernel segmentation (t-statistics through HBM)
    int32_t fill_blocks_per_cu = 0;      // > 0: cap the fill's workgroups per CU through its LDS request
};

// A/B switches of the experiments under scripts/ are read from the environment only by builds made with -DWSX_EXPERIMENT
// (scripts/build_exp.sh); the product library ignores them.
#ifdef WSX_EXPERIMENT
#include <cstdlib>
inline const char *wsx_exp_env(const char *name) { return getenv(name); }
#else
inline const char *wsx_exp_env(const char *) { return nullptr; }
#endif

#ifndef WSX_STACK_SLOT_DEFINED
#define WSX_STACK_SLOT_DEFINED
constexpr int WSX_DEV_STACK_SLOT = 2; // = WSX_STACK_SLOT of wsx_place.h (static_assert in wsx_api.hip)
#endif

struct DevAutomaton {
    int32_t n_states;
    int32_t endstate;
    int32_t flank_length;
    int32_t max_fanin;
    int32_t seq_idx_last;   // seq_idx[S-1]
    int32_t reverse;        // reverse-strand automaton: called sequences are reverse-complemented
    const double *value;
    const int32_t *seq_idx;
    const int32_t *pred_ptr;
    const int32_t *pred_idx;
    const uint8_t *repeat_mask;
    const uint8_t *last_base; // ASCII of the k-mer's last base, or NULL
    const uint16_t *paddr;    // register-resident fill: LDS export slot read by (slot k, predecessor f, lane), at
                              // [(k*WSX_MAX_F + f)*64 + lane]; absent predecessors read one of the 32 +inf slots K*64..
    const uint16_t *wslot;    // single-slot automata: the LDS export slot lane l writes (64 entries; wsx_place.h)
    const uint16_t *pos;      // state -> position (slot*64 + lane) in the register-resident fill; NULL = identity
    const uint16_t *state_at; // position -> state (0xFFFF = none); NULL = identity
    const uint64_t *pred4;  // per POSITION (slot*64 + lane; K*64 entries): the positions of its state's first four
                            // predecessors, 16 bits each (mask traceback)
    uint64_t stack_mask;    // stacked lane-major placement (wsx_place.h, LM = 4): lanes whose slot WSX_STACK_SLOT reads LDS
};

struct DevParams {
    int32_t m;                  // min_values_per_state
    int32_t states_in_segment;
    double threshold;
    double max_std;
    int32_t method_median;
    int32_t reps_as_one;
};

// Arguments of one DP pass (fill + traceback) over the reads listed in `order`.
struct PassArgs {
    const DevAutomaton *aut;
    const double *signal;      // per-sample, already offset to the chunk (index loff + i)
    const int64_t *offsets;    // global offsets[] (device copy), indexed by global read id
    const int32_t *aut_id;     // global
    const int32_t *order;      // read ids of this launch
    int32_t n_launch;
    int32_t first_read;
    int64_t base_off;
    const uint32_t *maskbits;  // packed mask (NULL = unmasked pass)
    uint32_t *bp;              // back-pointer scratch
    const int64_t *bp_off;     // per read of the chunk (index lr): where its back-pointer rows start, in 64-bit words
    uint16_t *run_state;       // runs, in reverse time order
    int32_t *run_start;
    int32_t *n_runs;           // per read
    uint16_t *trace;           // optional per-sample state ids
    double *end_cost;          // optional per read
    double *last_row;          // optional per read, stride last_row_stride
    int32_t last_row_stride;
    int32_t *status;           // per read: written by pass 1, read (skip if != 0) by pass 2
    int32_t check_status;      // 1: skip reads whose status is already non-zero and do not write status
    int32_t m;
    int32_t lane_major;        // the launch group's automata are placed lane-major (wsx_place.h): traceback shortcut
};

// Per-read record produced by borders_kernel for the segmentation kernels.
struct MidRec {
    int32_t start;   // first repeat transition (index into the run list)
    int32_t nsel;    // number of selected chunk boundaries
    int32_t p_lo;    // first sample any chunk reads (sel(0) - 3)
    int32_t p_hi;    // one past the last sample any chunk reads
    int32_t n_good;
    int32_t pad;
};

// Arguments of the alignment-statistics / masking stage.
struct MidArgs {
    const DevAutomaton *aut;
    DevParams prm;
    const double *signal;      // signal the trace was computed on
    const int64_t *offsets;
    const int32_t *aut_id;
    int32_t n_reads;           // reads in chunk
    int32_t first_read;
    int64_t base_off;
    const uint16_t *run_state;
    const int32_t *run_start;
    const int32_t *n_runs;
    int32_t pass;              // 1 or 2
    int32_t tcap;              // doubles of LDS reserved for the staged signal (set by the launcher)
    // alignment records (forward order), per-sample capacity
    double *al_value;
    double *al_expected;
    double *al_cost;
    uint8_t *al_good;
    // sorted filtered pairs for the rescaling fit (pass 1)
    double *fit_x;
    double *fit_y;
    int32_t *fit_m;            // per read
    MidRec *rec;               // per read
    int32_t *n_align;          // per read: record count in reps_as_one mode, else NULL
    int32_t *state_scratch;    // reps_as_one: 2*max_states ints per read
    int32_t max_states;
    double *scr0, *scr1, *scr2; // per-sample scratch (window statistics, t-statistics, compaction)
    uint32_t *maskbits;        // out (pass 1)
    uint8_t *badmask_bytes;    // optional out (pass 1), per sample
    int32_t *status;           // per read (in/out)
    const double *end_cost;    // per read, from the DP pass
    uint8_t *seq_out;          // optional: called sequence of this pass as ASCII (per-sample layout), or NULL
    // results
    void *results;             // wsx_result[] (per read)
};

struct FitArgs {
    const int64_t *offsets;
    int32_t n_reads;
    int32_t first_read;
    int64_t base_off;
    const double *fit_x;
    const double *fit_y;
    const int32_t *fit_m;
    double *coef;              // per read: xb, xe, c0..c3 (6 doubles)
    int32_t *status;
    // handles with rescaling.threshold > 1 (else NULL): reads whose least-squares cubic fails fpcurf's test queue up for
    // fit_smooth_kernel -- smooth_count[0] reads (list), smooth_count[1] the largest number of points among them
    int32_t *smooth_count;
    int32_t *smooth_list;
    int32_t *smooth_slot;      // per read: its slot of the smoothing workspace, -1 = the cubic in `coef`
};

struct EvalArgs {
    const int64_t *offsets;
    int32_t n_reads;
    int32_t first_read;
    int64_t base_off;
    const double *signal;
    const double *coef;
    const int32_t *status;
    double *out;               // rescaled signal
    double *out_user;          // optional second copy (user-visible), may be NULL
    const int32_t *smooth_slot; // see FitArgs; NULL unless rescaling.threshold > 1
    const double *smooth_ws;    // smoothing workspace of the chunk's work set (slots of smooth_nest knots)
    int32_t smooth_nest;
};

// Host-side launchers (defined next to the kernels).
// pk: packed mask rows (single-slot automata whose states with two predecessors sit in lanes 0..7): per 16 rows
// 16 x 8 bytes of first-candidate masks + 2 x 8 bytes holding the second candidate's byte of every row
// lm: lane-major placement (0 = no; 1 = slots 0 and K-1 export to LDS; 3 = slots 0, 1 and K-1; 2 = every slot), dtw_kernels.hip: dp_row
hipError_t wsx_launch_fill(const PassArgs &a, int m, int K, int F, int FL, bool pk, int lm, bool generic, const WsxTuning &tun, hipStream_t s);
hipError_t wsx_launch_traceback(const PassArgs &a, int K, int F, int FL, bool pk, bool generic, int n_aut, const WsxTuning &tun, hipStream_t s);
hipError_t wsx_launch_expand_trace(const PassArgs &a, hipStream_t s);
hipError_t wsx_launch_mid(const MidArgs &a, int max_T, const WsxTuning &tun, hipStream_t s);
hipError_t wsx_launch_fit(const FitArgs &a, hipStream_t s);
size_t wsx_smooth_workspace_bytes(int count, int max_m);
hipError_t wsx_launch_fit_smooth(const FitArgs &a, int count, int max_m, double *ws, hipStream_t s);
hipError_t wsx_launch_eval(const EvalArgs &a, int max_T, hipStream_t s);
bool wsx_fast_pass_supported(int m, int K, int F);
const char *wsx_pass_kernel_name(int m, int K, int F, int FL, bool pk, int lm, bool generic);
bool wsx_lane_major_supported(int m, int K);
bool wsx_split_supported(int m, int K);
