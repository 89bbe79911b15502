/*
 * warpstr_oracle.h -- interface of the CPU parity oracle (TEST INFRASTRUCTURE ONLY).
 * See warpstr_oracle.c.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product path never does.
 */
#ifndef WARPSTR_ORACLE_H
#define WARPSTR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    WSO_OK = 0,
    WSO_ERR_SHAPE = 1,         /* T <= m or S <= m (reference: IndexError, caller.py:206-208) */
    WSO_ERR_BACKTRACK = 2,     /* RuntimeError, caller.py:290-291 */
    WSO_ERR_FIT_POINTS = 3,    /* fewer than 4 states survive filter_alignment (splrep TypeError) */
    WSO_ERR_FIT_ORDER = 4,     /* degenerate abscissae (all equal) */
    WSO_ERR_FIT_SMOOTH = 5,    /* FITPACK's smoothing spline (rescaling.threshold > 1) came out with a coefficient that is not finite */
    WSO_ERR_NO_REPEAT = 6,     /* no repeat state on the path (trues[0] IndexError, caller.py:384) */
    WSO_ERR_SEGMENT_RANGE = 7, /* IndexError in find_event_borders/segment (caller.py:395-397, 361) */
};
#define WSO_SEGMENT_EMPTY (-1000000L)

typedef struct {
    int32_t n_states;
    int32_t endstate;
    int32_t flank_length;
    const double *value;        /* [S] */
    const int32_t *seq_idx;     /* [S] */
    const int32_t *pred_ptr;    /* [S+1] */
    const int32_t *pred_idx;    /* [E], reference `incoming` order */
    const uint8_t *repeat_mask; /* [S] */
} wso_automaton;

typedef struct {
    int32_t min_values_per_state; /* tr_calling_config.min_values_per_state (default 4) */
    int32_t states_in_segment;    /* tr_calling_config.states_in_segment (default 6) */
    double threshold;             /* rescaling.threshold (0.5) */
    double max_std;               /* rescaling.max_std (0.5) */
    int32_t method_median;        /* rescaling.method == 'median' */
    int32_t reps_as_one;          /* rescaling.reps_as_one */
} wso_params;

typedef struct {
    int32_t status;
    int32_t len1, len2;         /* len(seq), len(resc_seq): overview.csv `orig`, `results` */
    int32_t n_trans1, n_trans2; /* number of state transitions on each path */
    double cost1, cost2;        /* CallerResult.cost, .resc_cost */
    double dtw_end_cost1, dtw_end_cost2; /* D[T-1, endstate] of each pass */
} wso_result;

typedef struct { /* optional per-read intermediates; any pointer may be NULL */
    int32_t *trace1, *trace2;   /* [T] */
    double *rescaled, *rescaled2; /* [T] */
    uint8_t *badmask;           /* [T] */
    double *dlast1, *dlast2;    /* [S] last DP row of each pass */
    int64_t idx[4];             /* start, end, resc_start, resc_end */
    int64_t fit_knots;          /* knots of the first pass's spline: 8 = the least-squares cubic, more = FITPACK's smoothing branch */
} wso_debug;

double wso_np_mean(const double *a, long n);
double wso_np_std(const double *a, long n, double *tmp);
double wso_np_median(const double *a, long n, double *tmp);
int wso_dtw_fill(const wso_automaton *A, const double *sig, long T, const uint8_t *mask, int m, double *D);
int wso_backtrack(const wso_automaton *A, const double *D, const double *sig, long T, const uint8_t *mask, int m,
                  int32_t *trace);
long wso_transitions(const int32_t *trace, long T, int32_t *trans, int32_t *run_start);
long wso_create_alignment(const wso_automaton *A, const wso_params *P, const int32_t *trace, const double *sig,
                          long T, double *value, double *expected, uint8_t *good);
int wso_fit_cubic(const double *x, const double *y, long m, double t[8], double c[4], double *fp_out);
void wso_eval_cubic(const double t[8], const double c[4], const double *x, long n, double *out);
int wso_curfit(const double *x, const double *y, long m, double s, double *t_out, double *c_out, int *n_out, double *fp_out,
               int *ier_out);
void wso_splev(const double *t, int n, const double *c, const double *x, long cnt, double *out);
int wso_rescale_signal(const double *sig, long T, const double *value, const double *expected, const uint8_t *good,
                       long n_align, double *out, double tck_t[8], double tck_c[4]);
long wso_segment(const double *data, long n, int win);
int wso_mask_bad_repeats(const wso_automaton *A, const wso_params *P, const double *input_signal, long T,
                         const int32_t *trace, long *start_out, long *end_out, uint8_t *badmask);
void wso_sequence_span(const wso_automaton *A, const int32_t *trans, long ntr, long *lo, long *hi);
int wso_call_read(const wso_automaton *A, const wso_params *P, const double *sig, long T, wso_result *R,
                  wso_debug *dbg);

#ifdef __cplusplus
}
#endif
#endif
