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

/*
 * splrep(x, y, s = s) in full: FITPACK curfit/fpcurf for k = 3, unit weights, iopt = 0, tol = 0.001, maxit = 20,
 * nest = max(m + 4, 9) (scipy.interpolate.splrep's choices) -- src/caller/caller.py:311 whatever rescaling.threshold is.
 * Part 1 (fpcurf's main loop): least-squares spline on the current knots by Givens rotations; while fp - s >= acc
 * knots are added where the residual is largest (fpknot), their number per round following the decrease of fp.
 * Part 2: the smoothing spline -- the rows of the third-derivative jumps (fpdisc) weighted 1/p are rotated into the
 * triangle and p is iterated by rational interpolation (fprati) until |f(p) - s| < acc.
 * Arrays are 1-based as in the published routines so that the index arithmetic reads the same.
 * t, c: room for m + 4 (at least 9) doubles each.  *ier: FITPACK's ier (-2 polynomial, -1 interpolating, 0 smoothing, 1..3).
 * Pinned bit for bit against SciPy's compiled FITPACK: tests/test_oracle_golden.py::test_fitpack_smoothing_bitwise.
 */
static void fpback_n(double *const *a, const double *z, int n, int k, double *c)
{
    const int k1 = k - 1;
    c[n] = z[n] / a[n][1];
    int i = n - 1;
    if (i == 0) return;
    for (int j = 2; j <= n; j++) {
        double store = z[i];
        int i1 = k1;
        if (j <= k1) i1 = j - 1;
        int mm = i;
        for (int l = 1; l <= i1; l++) {
            mm = mm + 1;
            store = store - c[mm] * a[i][l + 1];
        }
        c[i] = store / a[i][1];
        i = i - 1;
    }
}

/* fpdisc: discontinuity jumps of the k-th derivative of the B-splines at the interior knots */
static void fpdisc(const double *t, int n, int k2, double *const *b)
{
    double h[13];
    const int k1 = k2 - 1, k = k1 - 1, nk1 = n - k1, nrint = nk1 - k;
    const double an = (double)nrint;
    const double fac = an / (t[nk1 + 1] - t[k1]);
    for (int l = k2; l <= nk1; l++) {
        const int lmk = l - k1;
        for (int j = 1; j <= k1; j++) {
            const int ik = j + k1, lj = l + j, lk = lj - k2;
            h[j] = t[l] - t[lk];
            h[ik] = t[l] - t[lj];
        }
        int lp = lmk;
        for (int j = 1; j <= k2; j++) {
            int jk = j;
            double prod = h[j];
            for (int i = 1; i <= k; i++) {
                jk = jk + 1;
                prod = prod * h[jk] * fac;
            }
            const int lk = lp + k1;
            b[lmk][j] = (t[lk] - t[lp]) / prod;
            lp = lp + 1;
        }
    }
}

/* fpknot: one more knot in the interval with the largest residual that still holds data points */
static void fpknot(const double *x, double *t, int *n, double *fpint, int *nrdata, int *nrint, int istart)
{
    const int k = (*n - *nrint - 1) / 2;
    double fpmax = 0.0;
    int jbegin = istart, number = 0, maxpt = 0, maxbeg = 0;
    for (int j = 1; j <= *nrint; j++) {
        const int jpoint = nrdata[j];
        if (!(fpmax >= fpint[j] || jpoint == 0)) {
            fpmax = fpint[j];
            number = j;
            maxpt = jpoint;
            maxbeg = jbegin;
        }
        jbegin = jbegin + jpoint + 1;
    }
    if (number == 0) return; /* (SciPy's copy leaves here too: no interval can take a knot) */
    const int ihalf = maxpt / 2 + 1;
    const int nrx = maxbeg + ihalf;
    const int next = number + 1;
    if (next <= *nrint)
        for (int j = next; j <= *nrint; j++) {
            const int jj = next + *nrint - j;
            fpint[jj + 1] = fpint[jj];
            nrdata[jj + 1] = nrdata[jj];
            const int jk = jj + k;
            t[jk + 1] = t[jk];
        }
    nrdata[number] = ihalf - 1;
    nrdata[next] = maxpt - ihalf;
    const double am = (double)maxpt;
    double an = (double)nrdata[number];
    fpint[number] = fpmax * an / am;
    an = (double)nrdata[next];
    fpint[next] = fpmax * an / am;
    const int jk = next + k;
    t[jk] = x[nrx];
    *n = *n + 1;
    *nrint = *nrint + 1;
}

static double fprati(double *p1, double *f1, double p2, double f2, double *p3, double *f3)
{
    double p;
    if (*p3 > 0.0) {
        const double h1 = *f1 * (f2 - *f3), h2 = f2 * (*f3 - *f1), h3 = *f3 * (*f1 - f2);
        p = -(*p1 * p2 * h3 + p2 * *p3 * h1 + *p3 * *p1 * h2) / (*p1 * h1 + p2 * h2 + *p3 * h3);
    } else {
        p = (*p1 * (*f1 - *f3) * f2 - p2 * (f2 - *f3) * *f1) / ((*f1 - f2) * *f3);
    }
    if (f2 < 0.0) {
        *p3 = p2;
        *f3 = f2;
    } else {
        *p1 = p2;
        *f1 = f2;
    }
    return p;
}

int wso_curfit(const double *x0, const double *y0, long m_, double s, double *t_out, double *c_out, int *n_out, double *fp_out,
               int *ier_out)
{
    enum { K = 3, K1 = 4, K2 = 5, MAXIT = 20 };
    const double tol = 0.001, con1 = 0.1, con9 = 0.9, con4 = 0.04, half = 0.5;
    const int m = (int)m_;
    if (m < K1) return WSO_ERR_FIT_POINTS;
    for (int i = 1; i < m; i++)
        if (x0[i - 1] > x0[i]) return WSO_ERR_FIT_ORDER;
    if (!(x0[0] < x0[m - 1])) return WSO_ERR_FIT_ORDER;
    const int nest = m + K1 > 2 * K + 3 ? m + K1 : 2 * K + 3;
    const double *x = x0 - 1, *y = y0 - 1; /* 1-based views */
    const double xb = x[1], xe = x[m];
    /* work arrays */
    size_t nd = (size_t)nest + 2;
    double *t = (double *)calloc(nd, sizeof(double)), *c = (double *)calloc(nd, sizeof(double));
    double *z = (double *)calloc(nd, sizeof(double)), *fpint = (double *)calloc(nd, sizeof(double));
    int *nrdata = (int *)calloc(nd, sizeof(int));
    double *abuf = (double *)calloc(nd * (K1 + 1), sizeof(double)), *bbuf = (double *)calloc(nd * (K2 + 1), sizeof(double));
    double *gbuf = (double *)calloc(nd * (K2 + 1), sizeof(double)), *qbuf = (double *)calloc(((size_t)m + 1) * (K1 + 1), sizeof(double));
    double **a = (double **)malloc(nd * sizeof(double *)), **b = (double **)malloc(nd * sizeof(double *));
    double **g = (double **)malloc(nd * sizeof(double *)), **q = (double **)malloc(((size_t)m + 1) * sizeof(double *));
    for (size_t i = 0; i < nd; i++) {
        a[i] = abuf + i * (K1 + 1);
        b[i] = bbuf + i * (K2 + 1);
        g[i] = gbuf + i * (K2 + 1);
    }
    for (int i = 0; i <= m; i++) q[i] = qbuf + (size_t)i * (K1 + 1);
    double h[8];
    const int nmin = 2 * K1;
    const double acc = tol * s;
    const int nmax = m + K1;
    int n = nmin, ier = 0, nplus = 0, nrint = 0, nk1 = 0;
    double fp = 0.0, fp0 = 0.0, fpold = 0.0, fpms = 0.0;
    nrdata[1] = m - 2;
    int restart = 1;
    while (restart) { /* label 60: re-entered once from "go to 10" (knots of the interpolating spline) */
        restart = 0;
        for (int iter = 1; iter <= m; iter++) {
            if (n == nmin) ier = -2;
            nrint = n - nmin + 1;
            nk1 = n - K1;
            int i = n;
            for (int j = 1; j <= K1; j++) {
                t[j] = xb;
                t[i] = xe;
                i = i - 1;
            }
            fp = 0.0;
            for (i = 1; i <= nk1; i++) {
                z[i] = 0.0;
                for (int j = 1; j <= K1; j++) a[i][j] = 0.0;
            }
            int l = K1;
            for (int it = 1; it <= m; it++) {
                const double xi = x[it], wi = 1.0;
                double yi = y[it] * wi;
                while (!(xi < t[l + 1] || l == nk1)) l = l + 1;
                fpbspl(t, K, xi, l, h);
                for (i = 1; i <= K1; i++) {
                    q[it][i] = h[i];
                    h[i] = h[i] * wi;
                }
                int j = l - K1;
                for (i = 1; i <= K1; i++) {
                    j = j + 1;
                    const double piv = h[i];
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
            if (ier == -2) fp0 = fp;
            fpint[n] = fp0;
            fpint[n - 1] = fpold;
            nrdata[n] = nplus;
            fpback_n(a, z, nk1, K1, c);
            fpms = fp - s;
            if (fabs(fpms) < acc) goto done;
            if (fpms < 0.0) goto part2;
            if (n == nmax) {
                ier = -1;
                goto done;
            }
            if (n == nest) {
                ier = 1;
                goto done;
            }
            if (ier != 0) {
                nplus = 1;
                ier = 0;
            } else {
                int npl1 = nplus * 2;
                const double rn = (double)nplus;
                if (fpold - fp > acc) { /* (Fortran's real -> integer assignment; out of range: what x86 gives, INT_MIN) */
                    const double v = rn * fpms / (fpold - fp);
                    npl1 = (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (-2147483647 - 1);
                }
                int mx = npl1 > nplus / 2 ? npl1 : nplus / 2;
                if (mx < 1) mx = 1;
                nplus = nplus * 2 < mx ? nplus * 2 : mx;
            }
            fpold = fp;
            double fpart = 0.0;
            i = 1;
            l = K2;
            int nw = 0;
            for (int it = 1; it <= m; it++) {
                if (!(x[it] < t[l] || l > nk1)) {
                    nw = 1;
                    l = l + 1;
                }
                double term = 0.0;
                int l0 = l - K2;
                for (int j = 1; j <= K1; j++) {
                    l0 = l0 + 1;
                    term = term + c[l0] * q[it][j];
                }
                term = (1.0 * (term - y[it])) * (1.0 * (term - y[it]));
                fpart = fpart + term;
                if (nw == 0) continue;
                const double store = term * half;
                fpint[i] = fpart - store;
                i = i + 1;
                fpart = store;
                nw = 0;
            }
            fpint[nrint] = fpart;
            int to_interp = 0;
            for (l = 1; l <= nplus; l++) {
                fpknot(x, t, &n, fpint, nrdata, &nrint, 1);
                if (n == nmax) {
                    to_interp = 1;
                    break;
                }
                if (n == nest) break;
            }
            if (to_interp) { /* label 10: the knots of the interpolating spline, then the main loop from its start */
                const int mk1 = m - K1;
                int ii = K2, jj = K / 2 + 2;
                for (l = 1; l <= mk1; l++) {
                    t[ii] = x[jj];
                    ii = ii + 1;
                    jj = jj + 1;
                }
                restart = 1;
                break;
            }
        }
        if (restart) continue;
        break;
    }
part2:
    if (ier == -2) goto done;
    {
        fpdisc(t, n, K2, b);
        double p1 = 0.0, f1 = fp0 - s, p3 = -1.0, f3 = fpms, p = 0.0;
        for (int i = 1; i <= nk1; i++) p = p + a[i][1];
        const double rn = (double)nk1;
        p = rn / p;
        int ich1 = 0, ich3 = 0;
        const int n8 = n - nmin;
        int iter;
        for (iter = 1; iter <= MAXIT; iter++) {
            const double pinv = 1.0 / p;
            for (int i = 1; i <= nk1; i++) {
                c[i] = z[i];
                g[i][K2] = 0.0;
                for (int j = 1; j <= K1; j++) g[i][j] = a[i][j];
            }
            for (int it = 1; it <= n8; it++) {
                for (int i = 1; i <= K2; i++) h[i] = b[it][i] * pinv;
                double yi = 0.0;
                for (int j = it; j <= nk1; j++) {
                    const double piv = h[1];
                    double cs, sn;
                    fpgivs(piv, &g[j][1], &cs, &sn);
                    fprota(cs, sn, &yi, &c[j]);
                    if (j == nk1) break;
                    int i2 = K1;
                    if (j > n8) i2 = nk1 - j;
                    for (int i = 1; i <= i2; i++) {
                        const int i1 = i + 1;
                        fprota(cs, sn, &h[i1], &g[j][i1]);
                        h[i] = h[i1];
                    }
                    h[i2 + 1] = 0.0;
                }
            }
            fpback_n(g, c, nk1, K2, c);
            fp = 0.0;
            int l = K2;
            for (int it = 1; it <= m; it++) {
                if (!(x[it] < t[l] || l > nk1)) l = l + 1;
                int l0 = l - K2;
                double term = 0.0;
                for (int j = 1; j <= K1; j++) {
                    l0 = l0 + 1;
                    term = term + c[l0] * q[it][j];
                }
                fp = fp + (1.0 * (term - y[it])) * (1.0 * (term - y[it]));
            }
            fpms = fp - s;
            if (fabs(fpms) < acc) goto done;
            if (iter == MAXIT) {
                ier = 3;
                goto done;
            }
            const double p2 = p, f2 = fpms;
            if (ich3 == 0) {
                if (!(f2 - f3 > acc)) { /* the initial choice of p is too large */
                    p3 = p2;
                    f3 = f2;
                    p = p * con4;
                    if (p <= p1) p = p1 * con9 + p2 * con1;
                    continue;
                }
                if (f2 < 0.0) ich3 = 1;
            }
            if (ich1 == 0) {
                if (!(f1 - f2 > acc)) { /* the initial choice of p is too small */
                    p1 = p2;
                    f1 = f2;
                    p = p / con4;
                    if (p3 < 0.0) continue;
                    if (p >= p3) p = p2 * con1 + p3 * con9;
                    continue;
                }
                if (f2 > 0.0) ich1 = 1;
            }
            if (f2 >= f1 || f2 <= f3) {
                ier = 2;
                goto done;
            }
            p = fprati(&p1, &f1, p2, f2, &p3, &f3);
        }
    }
done:
    for (int i = 1; i <= n; i++) {
        t_out[i - 1] = t[i];
        c_out[i - 1] = i <= n - K1 ? c[i] : 0.0;
    }
    *n_out = n;
    if (fp_out) *fp_out = fp;
    if (ier_out) *ier_out = ier;
    free(t); free(c); free(z); free(fpint); free(nrdata); free(abuf); free(bbuf); free(gbuf); free(qbuf);
    free(a); free(b); free(g); free(q);
    return WSO_OK;
}

/* splev(x, (t, c, 3)) with ext = 0 for any knot vector: src/caller/caller.py:312. */
void wso_splev(const double *t0, int n, const double *c0, const double *x, long cnt, double *out)
{
    enum { K1 = 4, K2 = 5 };
    const double *t = t0 - 1, *c = c0 - 1;
    const int nk1 = n - K1;
    int l = K1, l1 = l + 1;
    double h[8];
    for (long i = 0; i < cnt; i++) {
        const double arg = x[i];
        while (!(arg >= t[l] || l1 == K2)) {
            l1 = l;
            l = l - 1;
        }
        while (!(arg < t[l1] || l == nk1)) {
            l = l1;
            l1 = l + 1;
        }
        fpbspl(t, 3, arg, l, h);
        double sp = 0.0;
        int ll = l - K1;
        for (int j = 1; j <= K1; j++) {
            ll = ll + 1;
            sp = sp + c[ll] * h[j];
        }
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
    if (rc != WSO_OK) {
        free(x);
        return rc;
    }
    /* fpcurf: fpms = fp - s; |fpms| < acc (= 0.001 s) or fpms < 0 keeps the polynomial (ier = -2) */
    if (!(fp - (double)m < 0.001 * (double)m)) {
        /* possible with rescaling.threshold > 1 only: knots are added and the spline is smoothed (wso_curfit); ier 1..3 are
         * warnings in splrep and upstream goes on.  A spline with a coefficient that is not finite (coinciding abscissae
         * can do that) is refused on both sides of the parity tests: the second pass has nothing to align to. */
        double *tk = (double *)malloc(sizeof(double) * (size_t)(2 * (m + 16)));
        double *ck = tk + m + 16;
        int n = 0, ier = 0, finite = 1;
        rc = wso_curfit(x, y, m, (double)m, tk, ck, &n, NULL, &ier);
        for (int i = 0; rc == WSO_OK && i < n - 4; i++) finite = finite && isfinite(ck[i]);
        if (rc == WSO_OK && !finite) rc = WSO_ERR_FIT_SMOOTH;
        if (rc == WSO_OK) wso_splev(tk, n, ck, sig, T, out);
        free(tk);
        free(x);
        if (rc == WSO_OK && tck_t) { /* (the caller's 8 + 4 doubles cannot hold it: NaN marks the smoothing spline) */
            for (int i = 0; i < 8; i++) tck_t[i] = NAN;
            for (int i = 0; i < 4; i++) tck_c[i] = NAN;
            tck_t[1] = (double)n; /* its number of knots */
        }
        return rc;
    }
    free(x);
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
    {
        double tk[8], ck[4];
        if ((rc = wso_rescale_signal(sig, T, v1, e1, g1, n1, resc, tk, ck)) != WSO_OK) goto done;
        if (dbg) dbg->fit_knots = isnan(tk[0]) ? (int64_t)tk[1] : 8;
    }
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
    if (rc == WSO_ERR_FIT_SMOOTH) { /* (a spline that is not finite: upstream goes on, nobody looks at it) */
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
