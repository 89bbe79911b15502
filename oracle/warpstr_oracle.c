/*
 * warpstr_oracle.c -- CPU restatement of WarpSTR's per-read caller (step 3 of the pipeline).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP caller: it is built into
 * oracle/libwarpstr_oracle.so and may be loaded only by tests/, by __graft_entry__.smoke() and by
 * bench.py's cpu_baseline leg.  The product (warpstr_amd/) never links, loads or calls it.
 *
 * It restates, in plain scalar C (fp64, no FMA contraction: build with -ffp-contract=off), the
 * algorithm of the reference's Python caller.  Each function cites the reference lines it follows
 * (paths relative to the upstream repository root).  Deliberately literal: full T x S matrix,
 * the reference's loop order, its candidate re-computation in the traceback -- NOT the
 * data-parallel formulation the GPU uses -- so that agreement between the two is evidence.
 *
 * Third-party arithmetic on the path that is not in the upstream repository:
 *   - SciPy 1.6.3 (pinned, Pipfile:7-18; 1.15.3 in the dev container) interpolate.splrep/splev ->
 *     FITPACK (P. Dierckx) curfit/fpcurf/fpgivs/fprota/fpback/fpbspl/splev.  With s = m and the
 *     default rescaling.threshold <= 1 the smoothing loop exits at its first test with the
 *     weighted least-squares cubic polynomial (ier = -2, no interior knots); that published
 *     algorithm is restated here operation for operation (Givens row updates in data order).
 *   - NumPy pairwise summation in mean/std (numpy/core/src/umath/loops_utils.h: pairwise sum,
 *     blocks of 8 accumulators, recursion above 128 elements) and np.median.
 * Pinned by: the .npz fixtures under tests/golden, produced by running the upstream caller itself
 * (tests/golden/generate_golden.py); see tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "warpstr_oracle.h"

/* ------------------------------------------------------------------------------------------ */
/* NumPy reductions                                                                           */
/* ------------------------------------------------------------------------------------------ */

/* numpy pairwise_sum for contiguous doubles (published algorithm; see header comment). */
static double np_pairwise_sum(const double *a, long n)
{
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        long i;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

double wso_np_mean(const double *a, long n) { return np_pairwise_sum(a, n) / (double)n; }

/* np.std(a) (ddof = 0): sqrt(sum((a - mean)^2) / n), sums pairwise. tmp: n doubles scratch. */
double wso_np_std(const double *a, long n, double *tmp)
{
    double mean = np_pairwise_sum(a, n) / (double)n;
    for (long i = 0; i < n; i++) {
        double d = a[i] - mean;
        tmp[i] = d * d;
    }
    return sqrt(np_pairwise_sum(tmp, n) / (double)n);
}

static int cmp_double(const void *p, const void *q)
{
    double a = *(const double *)p, b = *(const double *)q;
    return (a > b) - (a < b);
}

/* np.median for finite data: mean of the middle one/two order statistics. */
double wso_np_median(const double *a, long n, double *tmp)
{
    memcpy(tmp, a, (size_t)n * sizeof(double));
    qsort(tmp, (size_t)n, sizeof(double), cmp_double);
    if (n % 2) return tmp[n / 2];
    return (tmp[n / 2 - 1] + tmp[n / 2]) / 2.0;
}

/* ------------------------------------------------------------------------------------------ */
/* DTW fill and traceback                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* src/caller/caller.py:198-245 (_calc_dtw_astates).  D is T*S row-major, caller allocated. */
int wso_dtw_fill(const wso_automaton *A, const double *sig, long T, const uint8_t *mask, int m, double *D)
{
    const int S = A->n_states;
    if (T <= m || S <= m) return WSO_ERR_SHAPE; /* reference would raise IndexError at 206-208 */
    for (long c = 0; c < T * (long)S; c++) D[c] = INFINITY;
    const double v0 = A->value[0];
    const double start_val = fabs(sig[0] - v0);
    D[0] = start_val;
    for (int i = 1; i <= m; i++) D[i] = start_val + fabs(sig[i] - v0); /* row 0, columns 1..m (sic) */

    const long boundary = A->flank_length - 10;
    const long after_repeat = A->seq_idx[S - 1] - boundary;
    const long first_threshold = 6 * boundary;
    const long second_threshold = T - 6 * boundary;

    for (long i = m; i < T; i++) {
        const double val = sig[i];
        const int back = (mask && mask[i]) ? m - 1 : m;
        double *row = D + i * S;
        const double *prow = D + (i - 1) * S;
        const double *brow = D + (i - back) * S;
        for (int j = 0; j < S; j++) {
            const long sj = A->seq_idx[j];
            if (i < first_threshold) {
                if (i < sj * 4 && i > sj * 15) continue;
            } else if (i > second_threshold && sj < after_repeat) {
                continue;
            }
            const double vj = A->value[j];
            if (prow[j] != INFINITY) {
                double cost = prow[j] + fabs(val - vj);
                if (cost < row[j]) row[j] = cost;
            }
            for (int e = A->pred_ptr[j]; e < A->pred_ptr[j + 1]; e++) {
                const int p = A->pred_idx[e];
                if (brow[p] == INFINITY) continue;
                const double vp = A->value[p];
                double skip_cost = brow[p];
                for (long q = i - back + 1; q < i; q++) skip_cost += fabs(sig[q] - vp);
                skip_cost += fabs(val - vj);
                if (skip_cost < row[j]) row[j] = skip_cost;
            }
        }
    }
    return WSO_OK;
}

/* src/caller/caller.py:247-301 (_backtracking).  trace: T ints (state per sample). */
int wso_backtrack(const wso_automaton *A, const double *D, const double *sig, long T, const uint8_t *mask, int m,
                  int32_t *trace)
{
    const int S = A->n_states;
    int newidx = A->endstate;
    long last = T - 1;
    long n = 0; /* trace is produced back to front */
    int skip_idx = -1;
    if (newidx < 0 || newidx >= S) return WSO_ERR_SHAPE;
    while (last != 0) {
        const double curr_dist = D[last * S + newidx];
        const double vj = A->value[newidx];
        double shift_delta;
        if (D[(last - 1) * S + newidx] == INFINITY) {
            shift_delta = INFINITY;
        } else {
            double shift = D[(last - 1) * S + newidx] + fabs(sig[last] - vj);
            shift_delta = fabs(shift - curr_dist);
        }
        double skip_delta = INFINITY;
        const int back = (mask && mask[last]) ? m - 1 : m;
        if (last - back >= 0) { /* rows < back hold inf cells only, the reference never transitions there */
            for (int e = A->pred_ptr[newidx]; e < A->pred_ptr[newidx + 1]; e++) {
                const int p = A->pred_idx[e];
                if (D[(last - back) * S + p] == INFINITY) continue;
                double skip_cost = D[(last - back) * S + p];
                for (long q = last - back + 1; q < last; q++) skip_cost += fabs(sig[q] - A->value[p]);
                skip_cost += fabs(sig[last] - vj);
                double delta = fabs(skip_cost - curr_dist);
                if (delta < skip_delta) {
                    skip_delta = delta;
                    skip_idx = p;
                }
            }
        }
        if (skip_delta < shift_delta) {
            if (skip_idx == -1) return WSO_ERR_BACKTRACK; /* RuntimeError at 290-291 */
            trace[T - 1 - n++] = newidx;
            for (int r = 0; r < back - 1; r++) trace[T - 1 - n++] = skip_idx;
            last -= back;
            newidx = skip_idx;
        } else {
            trace[T - 1 - n++] = newidx;
            last -= 1;
        }
    }
    trace[T - 1 - n++] = newidx;
    return n == T ? WSO_OK : WSO_ERR_BACKTRACK;
}

/* ------------------------------------------------------------------------------------------ */
/* Alignment statistics (run-length encoding of the trace)                                    */
/* ------------------------------------------------------------------------------------------ */

/* WarpResult.state_transitions, src/caller/caller.py:58-60: trace value at every run start. */
long wso_transitions(const int32_t *trace, long T, int32_t *trans, int32_t *run_start)
{
    long n = 0;
    for (long i = 0; i < T; i++) {
        if (i == 0 || trace[i] != trace[i - 1]) {
            trans[n] = trace[i];
            if (run_start) run_start[n] = (int32_t)i;
            n++;
        }
    }
    return n;
}

/*
 * WarpResult.create_alignment + StateAlignment, src/caller/caller.py:17-43,65-96,321-327.
 * Returns the number of alignment records; fills value/expected/good (caller sized >= T).
 */
long wso_create_alignment(const wso_automaton *A, const wso_params *P, const int32_t *trace, const double *sig, long T,
                          double *value, double *expected, uint8_t *good)
{
    double *raws = (double *)malloc(sizeof(double) * (size_t)T * 2);
    double *tmp = raws + T;
    long n = 0;
    if (P->reps_as_one) {
        /* np.unique(state_transitions): ascending distinct states; raws = all samples of the state (69-79) */
        uint8_t *seen = (uint8_t *)calloc((size_t)A->n_states, 1);
        for (long i = 0; i < T; i++) seen[trace[i]] = 1;
        for (int s = 0; s < A->n_states; s++) {
            if (!seen[s]) continue;
            long cnt = 0;
            for (long i = 0; i < T; i++)
                if (trace[i] == s) raws[cnt++] = sig[i];
            value[n] = P->method_median ? wso_np_median(raws, cnt, tmp) : wso_np_mean(raws, cnt);
            expected[n] = A->value[s];
            good[n] = (cnt >= P->min_values_per_state) && (wso_np_std(raws, cnt, tmp) < P->max_std) &&
                      (fabs(expected[n] - value[n]) <= P->threshold);
            n++;
        }
        free(seen);
    } else {
        long i = 0;
        while (i < T) {
            long j = i;
            while (j < T && trace[j] == trace[i]) j++;
            long cnt = j - i;
            const double *r = sig + i;
            value[n] = P->method_median ? wso_np_median(r, cnt, tmp) : wso_np_mean(r, cnt);
            expected[n] = A->value[trace[i]];
            good[n] = (cnt >= P->min_values_per_state) && (wso_np_std(r, cnt, tmp) < P->max_std) &&
                      (fabs(expected[n] - value[n]) <= P->threshold);
            n++;
            i = j;
        }
    }
    free(raws);
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* FITPACK: weighted least-squares cubic in B-spline form, and its evaluation                  */
/* ------------------------------------------------------------------------------------------ */

/* fpbspl (Dierckx): the k+1 non-zero B-splines of degree k at x, t(l) <= x < t(l+1). 1-based t. */
static void fpbspl(const double *t, int k, double x, int l, double *h /* [1..k+1] */)
{
    double hh[20];
    h[1] = 1.0;
    for (int j = 1; j <= k; j++) {
        for (int i = 1; i <= j; i++) hh[i] = h[i];
        h[1] = 0.0;
        for (int i = 1; i <= j; i++) {
            int li = l + i, lj = li - j;
            if (t[li] == t[lj]) {
                h[i + 1] = 0.0;
                continue;
            }
            double f = hh[i] / (t[li] - t[lj]);
            h[i] = h[i] + f * (t[li] - x);
            h[i + 1] = f * (x - t[lj]);
        }
    }
}

/* fpgivs: parameters of a Givens rotation. */
static void fpgivs(double piv, double *ww, double *c, double *s)
{
    double store = fabs(piv), dd;
    if (store >= *ww) {
        double r = *ww / piv;
        dd = store * sqrt(1.0 + r * r);
    } else {
        double r = piv / *ww;
        dd = *ww * sqrt(1.0 + r * r);
    }
    *c = *ww / dd;
    *s = piv / dd;
    *ww = dd;
}

/* fprota: apply a Givens rotation to a and b. */
static void fprota(double c, double s, double *a, double *b)
{
    double stor1 = *a, stor2 = *b;
    *b = c * stor2 + s * stor1;
    *a = c * stor1 - s * stor2;
}

/*
 * splrep(x, y, s=len(x)) for the case it always takes on this path (ier = -2): k = 3, unit
 * weights, knots t = [xb]*4 + [xe]*4, coefficients from fpcurf's Givens triangularisation followed
 * by fpback.  src/caller/caller.py:311.  x must be sorted ascending, m >= 4, x[0] < x[m-1].
 * Also returns fp (sum of squared residuals) so the caller can assert fp < s.
 */
int wso_fit_cubic(const double *x, const double *y, long m, double t[8], double c[4], double *fp_out)
{
    enum { K = 3, K1 = 4, NK1 = 4 };
    if (m < K1) return WSO_ERR_FIT_POINTS; /* splrep: TypeError m > k must hold */
    for (long i = 1; i < m; i++)
        if (x[i - 1] > x[i]) return WSO_ERR_FIT_ORDER;
    const double xb = x[0], xe = x[m - 1];
    if (!(xb < xe)) return WSO_ERR_FIT_ORDER;
    double tt[9]; /* 1-based */
    for (int j = 1; j <= K1; j++) {
        tt[j] = xb;
        tt[9 - j] = xe;
    }
    double a[5][5], z[5], h[6];
    memset(a, 0, sizeof(a));
    memset(z, 0, sizeof(z));
    double fp = 0.0;
    const int l = K1; /* single knot interval */
    for (long it = 0; it < m; it++) {
        double xi = x[it];
        double wi = 1.0;
        double yi = y[it] * wi;
        fpbspl(tt, K, xi, l, h);
        for (int i = 1; i <= K1; i++) h[i] = h[i] * wi;
        int j = l - K1;
        int i;
        for (i = 1; i <= K1; i++) {
            j = j + 1;
            double piv = h[i];
            if (piv == 0.0) continue;
            double cs, sn;
            fpgivs(piv, &a[j][1], &cs, &sn);
            fprota(cs, sn, &yi, &z[j]);
            if (i == K1) break;
            int i2 = 1;
            for (int i1 = i + 1; i1 <= K1; i1++) {
                i2 = i2 + 1;
                fprota(cs, sn, &h[i1], &a[j][i2]);
            }
        }
        fp = fp + yi * yi;
    }
    /* fpback: back substitution of the banded upper triangular system (n = NK1, bandwidth K1) */
    double cc[5];
    cc[NK1] = z[NK1] / a[NK1][1];
    int i = NK1 - 1;
    for (int j = 2; j <= NK1; j++) {
        double store = z[i];
        int i1 = K1 - 1;
        if (j <= K1 - 1) i1 = j - 1;
        int mm = i;
        for (int ll = 1; ll <= i1; ll++) {
            mm = mm + 1;
            store = store - cc[mm] * a[i][ll + 1];
        }
        cc[i] = store / a[i][1];
        i = i - 1;
    }
    for (int q = 0; q < 8; q++) t[q] = tt[q + 1];
    for (int q = 0; q < 4; q++) c[q] = cc[q + 1];
    if (fp_out) *fp_out = fp;
    return WSO_OK;
}

/* splev(x, tck) with ext = 0 (extrapolate) for the 8-knot cubic: src/caller/caller.py:312. */
void wso_eval_cubic(const double t[8], const double c[4], const double *x, long n, double *out)
{
    double tt[9], h[6];
    for (int q = 0; q < 8; q++) tt[q + 1] = t[q];
    for (long i = 0; i < n; i++) {
        fpbspl(tt, 3, x[i], 4, h);
        double sp = 0.0;
        for (int j = 1; j <= 4; j++) sp = sp + c[j - 1] * h[j];
        out[i] = sp;
    }
}

typedef struct {
    double x, y;
    long k;
} fit_pair;

static int cmp_pair(const void *p, const void *q)
{
    const fit_pair *a = (const fit_pair *)p, *b = (const fit_pair *)q;
    if (a->x < b->x) return -1;
    if (a->x > b->x) return 1;
    return (a->k > b->k) - (a->k < b->k); /* list.sort is stable */
}

/* rescale_signal + filter_alignment, src/caller/caller.py:304-318. */
int wso_rescale_signal(const double *sig, long T, const double *value, const double *expected, const uint8_t *good,
                       long n_align, double *out, double tck_t[8], double tck_c[4])
{
    fit_pair *pr = (fit_pair *)malloc(sizeof(fit_pair) * (size_t)(n_align > 0 ? n_align : 1));
    long m = 0;
    for (long i = 0; i < n_align; i++)
        if (good[i]) {
            pr[m].x = value[i];
            pr[m].y = expected[i];
            pr[m].k = m;
            m++;
        }
    qsort(pr, (size_t)m, sizeof(fit_pair), cmp_pair);
    double *x = (double *)malloc(sizeof(double) * (size_t)(2 * m + 2));
    double *y = x + m + 1;
    for (long i = 0; i < m; i++) {
        x[i] = pr[i].x;
        y[i] = pr[i].y;
    }
    double t[8], c[4], fp;
    int rc = wso_fit_cubic(x, y, m, t, c, &fp);
    free(pr);
    free(x);
    if (rc != WSO_OK) return rc;
    /* fpcurf: fpms = fp - s; |fpms| < acc (= 0.001 s) or fpms < 0 keeps the polynomial (ier = -2) */
    if (!(fp - (double)m < 0.001 * (double)m)) return WSO_ERR_FIT_SMOOTH; /* knots would be added: not restated */
    wso_eval_cubic(t, c, sig, T, out);
    if (tck_t) memcpy(tck_t, t, sizeof(t));
    if (tck_c) memcpy(tck_c, c, sizeof(c));
    return WSO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Bad-repeat masking                                                                         */
/* ------------------------------------------------------------------------------------------ */

/* calc_ttest, src/caller/caller.py:347-354.  np.std(..)**2 squares the rooted value (sic). */
static double calc_ttest(const double *a1, const double *a2, int win, double *tmp)
{
    double s1 = wso_np_std(a1, win, tmp), s2 = wso_np_std(a2, win, tmp);
    double sd = sqrt((s1 * s1 + s2 * s2) / (double)win);
    if (sd == 0.0) sd = sd + 0.0000001;
    return (wso_np_mean(a1, win) - wso_np_mean(a2, win)) / sd;
}

/* segment, src/caller/caller.py:357-378: number of detected events minus one in data[0..n). */
long wso_segment(const double *data, long n, int win)
{
    long nt = n - 2 * win + 1;
    if (nt <= 0) return WSO_SEGMENT_EMPTY; /* t_stats[0] would raise IndexError */
    double tmp[16];
    long borders = 0;
    int start = 0;
    double prev = 0.0;
    for (long q = 0; q < nt; q++) {
        long idx = q + win;
        double tq = calc_ttest(data + idx - win, data + idx, win, tmp);
        if (q == 0) prev = tq;
        if (tq > 3 || tq < -3) {
            if ((tq > 3 && tq >= prev) || (tq < -3 && tq <= prev)) {
                start = 1;
            } else {
                if (start) borders++;
                start = 0;
            }
        } else if (start) {
            borders++;
            start = 0;
        }
        prev = tq;
    }
    return borders - 1;
}

/*
 * mask_bad_repeats / find_event_borders / check_segments / mask_big_events,
 * src/caller/caller.py:330-344,381-421.  Outputs start,end (indices into the transition list)
 * and the per-sample mask (T bytes, may be NULL).
 */
int wso_mask_bad_repeats(const wso_automaton *A, const wso_params *P, const double *input_signal, long T,
                         const int32_t *trace, long *start_out, long *end_out, uint8_t *badmask)
{
    const int sis = P->states_in_segment;
    const int win = 3;
    int32_t *trans = (int32_t *)malloc(sizeof(int32_t) * (size_t)T);
    long ntr = wso_transitions(trace, T, trans, NULL);
    long start = -1, end = -1;
    for (long i = 0; i < ntr; i++)
        if (A->repeat_mask[trans[i]]) {
            if (start < 0) start = i;
            end = i;
        }
    int rc = WSO_OK;
    long *bounds = NULL;
    if (start < 0) {
        rc = WSO_ERR_NO_REPEAT; /* trues[0] IndexError */
        goto done;
    }
    long start_idx = -1, end_idx = -1, nb;
    bounds = (long *)malloc(sizeof(long) * (size_t)(T + 1));
    for (int round = 0; round < 2; round++) {
        int32_t start_state = trans[start], end_state = trans[end];
        start_idx = end_idx = -1;
        for (long i = 0; i < T; i++)
            if (trace[i] == start_state) {
                start_idx = i;
                break;
            }
        for (long i = T - 1; i >= 0; i--)
            if (trace[i] == end_state) {
                end_idx = i;
                break;
            }
        /* bounds = where(diff(trace[start_idx:end_idx+1]) != 0); empty slice when end_idx < start_idx */
        nb = 0;
        for (long i = start_idx; i < end_idx; i++)
            if (trace[i] != trace[i + 1]) bounds[nb++] = i - start_idx;
        if (round == 1) break;
        /* Python's % with a positive modulus is non-negative: (len(bounds)-1) % sis */
        long add = ((nb - 1) % sis + sis) % sis;
        if (add > 0) {
            end = end + (sis - add);
            if (end >= ntr) {
                rc = WSO_ERR_SEGMENT_RANGE; /* state_transitions[end] IndexError (395-397) */
                goto done;
            }
        } else {
            break;
        }
    }
    /* every sis-th boundary, as absolute sample indices */
    long nsel = 0;
    for (long i = 0; i < nb; i++)
        if (i % sis == 0) bounds[nsel++] = start_idx + bounds[i];
    if (nsel == 0) {
        rc = WSO_ERR_SEGMENT_RANGE; /* bounds[0] IndexError in mask_big_events */
        goto done;
    }
    *start_out = start;
    *end_out = end;
    if (badmask) memset(badmask, 0, (size_t)T);
    for (long k = 0; k + 1 < nsel; k++) {
        long lo = bounds[k] - win, hi = bounds[k + 1] + win;
        if (hi > T) hi = T; /* Python slices clamp at the end */
        if (lo < 0) {
            rc = WSO_ERR_SEGMENT_RANGE; /* a negative slice start wraps in Python: not restated */
            goto done;
        }
        long len = wso_segment(input_signal + lo, hi - lo, win);
        if (len == WSO_SEGMENT_EMPTY) {
            rc = WSO_ERR_SEGMENT_RANGE;
            goto done;
        }
        if (len >= sis + 1 && badmask) /* check_segments: >= states_in_segment + 1 */
            for (long i = bounds[k]; i < bounds[k + 1]; i++) badmask[i] = 1;
    }
done:
    free(trans);
    free(bounds);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Sequence / allele length                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* Python slice [a:b] bounds for a sequence of length n. */
static void py_slice(long a, long b, long n, long *lo, long *hi)
{
    if (a < 0) a += n;
    if (a < 0) a = 0;
    if (a > n) a = n;
    if (b < 0) b += n;
    if (b < 0) b = 0;
    if (b > n) b = n;
    *lo = a;
    *hi = b < a ? a : b;
}

/*
 * WarpSTR._get_sequence, src/caller/caller.py:178-187 (before the reverse-strand complement,
 * which does not change the length): indices [lo, hi) into the transition list that survive
 * flank stripping.  seq[flank-offset : -flank] with Python slice semantics.
 */
void wso_sequence_span(const wso_automaton *A, const int32_t *trans, long ntr, long *lo, long *hi)
{
    long offset = A->seq_idx[trans[0]];
    long fl = A->flank_length;
    /* "-flank_length" with flank_length == 0 is 0, i.e. an empty slice */
    py_slice(fl - offset, -fl, ntr, lo, hi);
}

/* mean of alignment costs over the Python slice [start:end]; NaN when empty (np.mean([])). */
static double mean_cost(const double *value, const double *expected, long n, long start, long end, double *tmp)
{
    long lo, hi;
    py_slice(start, end, n, &lo, &hi);
    if (hi <= lo) return NAN;
    for (long i = lo; i < hi; i++) tmp[i - lo] = fabs(value[i] - expected[i]);
    return wso_np_mean(tmp, hi - lo);
}

/* ------------------------------------------------------------------------------------------ */
/* Whole read: WarpSTR.run, src/caller/caller.py:117-149                                      */
/* ------------------------------------------------------------------------------------------ */

int wso_call_read(const wso_automaton *A, const wso_params *P, const double *sig, long T, wso_result *R,
                  wso_debug *dbg)
{
    const int S = A->n_states, m = P->min_values_per_state;
    memset(R, 0, sizeof(*R));
    R->cost1 = R->cost2 = NAN;
    if (T <= m || S <= m) return R->status = WSO_ERR_SHAPE;
    int rc = WSO_OK;
    double *D = (double *)malloc(sizeof(double) * (size_t)T * (size_t)S);
    int32_t *trace1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)T * 4);
    int32_t *trace2 = trace1 + T, *trans = trace2 + T;
    double *buf = (double *)malloc(sizeof(double) * (size_t)T * 8);
    double *resc = buf, *resc2 = buf + T, *v1 = buf + 2 * T, *e1 = buf + 3 * T, *v2 = buf + 4 * T, *e2 = buf + 5 * T,
           *tmp = buf + 6 * T;
    uint8_t *g1 = (uint8_t *)malloc((size_t)T * 3), *g2 = g1 + T, *bad = g1 + 2 * T;
    long n1 = 0, n2 = 0, start = 0, end = 0, rstart = 0, rend = 0, lo, hi, ntr;

    if ((rc = wso_dtw_fill(A, sig, T, NULL, m, D)) != WSO_OK) goto done;
    R->dtw_end_cost1 = D[(T - 1) * S + A->endstate];
    if (dbg && dbg->dlast1) memcpy(dbg->dlast1, D + (T - 1) * S, sizeof(double) * (size_t)S);
    if ((rc = wso_backtrack(A, D, sig, T, NULL, m, trace1)) != WSO_OK) goto done;
    if (dbg && dbg->trace1) memcpy(dbg->trace1, trace1, sizeof(int32_t) * (size_t)T);
    ntr = wso_transitions(trace1, T, trans, NULL);
    wso_sequence_span(A, trans, ntr, &lo, &hi);
    R->len1 = (int32_t)(hi - lo);
    R->n_trans1 = (int32_t)ntr;

    n1 = wso_create_alignment(A, P, trace1, sig, T, v1, e1, g1);
    if ((rc = wso_rescale_signal(sig, T, v1, e1, g1, n1, resc, NULL, NULL)) != WSO_OK) goto done;
    if (dbg && dbg->rescaled) memcpy(dbg->rescaled, resc, sizeof(double) * (size_t)T);
    if ((rc = wso_mask_bad_repeats(A, P, sig, T, trace1, &start, &end, bad)) != WSO_OK) goto done;
    if (dbg && dbg->badmask) memcpy(dbg->badmask, bad, (size_t)T);

    /* `if mask` at caller.py:190: a non-empty list is truthy, so the mask is always used */
    if ((rc = wso_dtw_fill(A, resc, T, bad, m, D)) != WSO_OK) goto done;
    R->dtw_end_cost2 = D[(T - 1) * S + A->endstate];
    if (dbg && dbg->dlast2) memcpy(dbg->dlast2, D + (T - 1) * S, sizeof(double) * (size_t)S);
    if ((rc = wso_backtrack(A, D, resc, T, bad, m, trace2)) != WSO_OK) goto done;
    if (dbg && dbg->trace2) memcpy(dbg->trace2, trace2, sizeof(int32_t) * (size_t)T);
    ntr = wso_transitions(trace2, T, trans, NULL);
    wso_sequence_span(A, trans, ntr, &lo, &hi);
    R->len2 = (int32_t)(hi - lo);
    R->n_trans2 = (int32_t)ntr;

    n2 = wso_create_alignment(A, P, trace2, resc, T, v2, e2, g2);
    /* caller.py:132-135: this second spline only feeds mask_bad_repeats, of which only the two indices are kept -- and those
     * come from the state path alone.  If FITPACK leaves its polynomial branch HERE (possible with rescaling.threshold > 1)
     * upstream goes on with a smoothing spline nobody looks at: not an error of the read. */
    rc = wso_rescale_signal(resc, T, v2, e2, g2, n2, resc2, NULL, NULL);
    if (rc == WSO_ERR_FIT_SMOOTH) {
        memcpy(resc2, resc, sizeof(double) * (size_t)T);
        rc = WSO_OK;
    }
    if (rc != WSO_OK) goto done;
    if (dbg && dbg->rescaled2) memcpy(dbg->rescaled2, resc2, sizeof(double) * (size_t)T);
    if ((rc = wso_mask_bad_repeats(A, P, resc2, T, trace2, &rstart, &rend, NULL)) != WSO_OK) goto done;

    R->cost1 = mean_cost(v1, e1, n1, start, end, tmp);
    R->cost2 = mean_cost(v2, e2, n2, rstart, rend, tmp);
    if (dbg) {
        dbg->idx[0] = start;
        dbg->idx[1] = end;
        dbg->idx[2] = rstart;
        dbg->idx[3] = rend;
    }
done:
    free(D);
    free(trace1);
    free(buf);
    free(g1);
    return R->status = rc;
}
