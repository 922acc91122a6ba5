/*
 * oracle/dff_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's 1-D layered travel-time forward
 * model (module raymod, /root/reference/subroutineR-quiet.f90).  It exists
 * only so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg can check / time the HIP path against the reference's arithmetic on a
 * box where /root/reference is absent.  Nothing under raytracerfortran_amd/
 * may link, import or call it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file
 *   - against the reference author's own gfortran outputs (notebook print of
 *     5 travel times, KAT-A; committed rays.dat ray geometry, KAT-B), and
 *   - against fixtures generated in the build container by running the
 *     reference source itself (compiled unmodified with amdflang into
 *     oracle/_ref/, recipe in oracle/Makefile; generator
 *     tests/golden/make_golden.py).
 *
 * Restatement rules (they decide the result, because the reference stops its
 * root-find at |f| < 0.1 m rather than at convergence, so the returned time
 * depends on the exact iteration sequence):
 *   - squares are x*x, the cube is (s*s)*s, the numerator is (H*V)*x;
 *   - SUM() is a plain left-to-right accumulation starting from the first
 *     element (gfortran's inline lowering);
 *   - no FMA contraction: build with -ffp-contract=off, no -ffast-math;
 *   - 1/maxval(V) is taken over the truncated stack V(1:nl);
 *   - constants 0.1, 15, 20, 1e-9, 1e-10, 1e-12 exactly as in the source.
 */
#include <math.h>
#include <stdlib.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/*
 * One source, two builds: the default is the reference's precision (double).  -DORACLE_F32 builds the same
 * restatement in IEEE single precision (every constant rounded to float, sqrtf, float divide) under the symbol
 * suffix _f32: the checker of the fp32 kernel path (BASELINE config 5).  The reference itself has no single
 * precision build of this module, so the fp32 oracle pins operation order, not a reference output.
 */
#ifdef ORACLE_F32
typedef float real_t;
#define R_SQRT sqrtf
#define R_FABS fabsf
#define R_ISNAN(x) isnan(x)
#define C(x) ((real_t)(x))
#define SYM(name) name##_f32
#else
typedef double real_t;
#define R_SQRT sqrt
#define R_FABS fabs
#define R_ISNAN(x) isnan(x)
#define C(x) (x)
#define SYM(name) name
#endif

#define ORACLE_MAX_STACK 4096 /* layers above one source; heap not needed below this */

/* Optional per-call branch statistics; layout mirrored in oracle/oracle.py. */
typedef struct {
    int64_t top_layer;    /* nl == 1, straight ray                          */
    int64_t p0_halvings;  /* trips of the p0 /= 2 loop          (:125-133)   */
    int64_t cf_negative;  /* branch 1: cf < 0 -> Newton         (:139-141)   */
    int64_t safe_jump;    /* branch 2: first jump safe -> Newton (:142-144)  */
    int64_t bisect;       /* branch 3: bisection then Newton    (:145-153)   */
    int64_t newton_iters; /* sum of Newton updates performed                 */
    int64_t bisect_trips; /* sum of bisection loop trips                     */
    int64_t clamps;       /* x > 1/max(V) clamp events          (:299-304)   */
    int64_t not_conv;     /* -999 results                       (:167-169)   */
    int64_t f_evals;      /* costFunc calls, as the reference makes them     */
    int64_t fp_evals;     /* costFunc_Prime calls                            */
} oracle_stats;

static oracle_stats *g_stats = NULL; /* not thread safe: test use only */

/* subroutineR-quiet.f90:9-32 whichLayer (returns 1-based layer index). */
static int which_layer(const real_t *depths, int nlayers, real_t dph)
{
    int in_n = 0;
    real_t diff = C(0.0);
    for (int i = 1; i <= nlayers; ++i) {
        in_n = i;
        diff = depths[i - 1] - dph;
        if (diff > C(0.0))
            break;
    }
    if (diff < C(0.0))
        return nlayers + 1;
    return in_n;
}

/* subroutineR-quiet.f90:184-202 costFunc */
static real_t cost_func(real_t x, const real_t *H, const real_t *V, int n, real_t R)
{
    real_t sum_all = C(0.0);
    for (int i = 0; i < n; ++i) {
        real_t term = ((H[i] * V[i]) * x) / R_SQRT(C(1.0) - (x * x) * (V[i] * V[i]));
        sum_all = (i == 0) ? term : sum_all + term;
    }
    if (g_stats)
        g_stats->f_evals++;
    return R - sum_all;
}

/* subroutineR-quiet.f90:206-221 costFunc_Prime */
static real_t cost_func_prime(real_t x, const real_t *H, const real_t *V, int n)
{
    real_t sum_all = C(0.0);
    for (int i = 0; i < n; ++i) {
        real_t s = R_SQRT(C(1.0) - (x * x) * (V[i] * V[i]));
        real_t term = (H[i] * V[i]) / ((s * s) * s);
        sum_all = (i == 0) ? term : sum_all + term;
    }
    if (g_stats)
        g_stats->fp_evals++;
    return -sum_all;
}

static real_t max_val(const real_t *V, int n)
{
    real_t m = V[0];
    for (int i = 1; i < n; ++i)
        if (V[i] > m)
            m = V[i];
    return m;
}

/* subroutineR-quiet.f90:226-332 solve (Newton, maxiter 15, tol 0.1).
 * Returns x; *conv follows the reference's inverted final test (:327-330). */
static real_t solve_newton(real_t x0, const real_t *H, const real_t *V, int n, real_t R, int *conv)
{
    const int maxiter = 15;
    const real_t tol = C(1.e-1);
    const real_t rr = C(1e-9);
    real_t x = x0, fx = C(0.0), fxprime = C(0.0);
    int k;
    *conv = 0;
    for (k = 1; k <= maxiter; ++k) {
        fx = cost_func(x, H, V, n, R);
        fxprime = cost_func_prime(x, H, V, n);
        if (R_FABS(fx) < tol) {
            *conv = 1;
            break;
        }
        real_t deltax = fx / fxprime;
        x = x - deltax;
        if (g_stats)
            g_stats->newton_iters++;
        if (x > C(1.0) / max_val(V, n)) {
            x = C(1.0) / max_val(V, n) - rr;
            if (g_stats)
                g_stats->clamps++;
        }
    }
    if (k > maxiter)
        fx = cost_func(x, H, V, n, R);
    if (R_FABS(fx) > tol)
        *conv = 1;
    return x;
}

/* subroutineR-quiet.f90:339-405 solvebst (guarded bisection, maxiter 20). */
static real_t solve_bisect(real_t x1, real_t x2, const real_t *H, const real_t *V, int n, real_t R)
{
    const int maxiter = 20;
    const real_t tol = C(1e-1);
    real_t x, dx, fmid, f, fx, fxprime, xmid, deltax, check_x;
    fmid = cost_func(x2, H, V, n, R); /* :353 value is never used */
    f = cost_func(x1, H, V, n, R);
    (void)fmid;
    if (f < C(0.0)) {
        x = x1;
        dx = x2 - x1;
    } else {
        x = x2;
        dx = x1 - x2;
    }
    for (int k = 1; k <= maxiter; ++k) {
        if (g_stats)
            g_stats->bisect_trips++;
        dx = dx * C(0.5);
        xmid = x + dx;
        fmid = cost_func(xmid, H, V, n, R);
        if (fmid < C(0.0))
            x = xmid;
        if (fmid == C(0.0))
            break;
        fx = cost_func(xmid, H, V, n, R);
        fxprime = cost_func_prime(xmid, H, V, n);
        deltax = fx / fxprime;
        check_x = x - deltax;
        if (check_x < C(1.0) / max_val(V, n) - (C(1e-10))) {
            x = xmid;
            break;
        }
        if (R_FABS(fmid) < tol)
            break;
    }
    return x;
}

/* subroutineR-quiet.f90:77-178 GetPTime for nl >= 2 (H, V are the stack above
 * the source, already differenced).  Writes the final ray parameter to *p_out. */
static real_t get_ptime_stack(real_t src_depth, real_t src_offset, const real_t *H, const real_t *V,
                              int n, real_t *p_out)
{
    int conv = 0;
    real_t s = C(0.0);
    for (int i = 0; i < n; ++i) {
        real_t q = H[i] / V[i];
        s = (i == 0) ? q : s + q;
    }
    real_t c_harmonic = src_depth / s;
    real_t cos_t = src_depth / R_SQRT(src_offset * src_offset + src_depth * src_depth);
    real_t weight = C(1.0);
    real_t p0 = cos_t / c_harmonic * weight;

    /* :125-133.  The reference loops forever when p0 is NaN/Inf; the oracle
     * (like the HIP path) gives up after 4096 halvings and reports -999. */
    int guard = 0;
    for (;;) {
        real_t acc = C(0.0);
        for (int i = 0; i < n; ++i) {
            real_t t = R_SQRT(C(1.0) - (p0 * p0) * ((V[i] + C(1.0)) * (V[i] + C(1.0))));
            acc = (i == 0) ? t : acc + t;
        }
        if (!isnan(acc))
            break;
        p0 = p0 / C(2.0);
        if (g_stats)
            g_stats->p0_halvings++;
        if (++guard > 4096) {
            if (p_out)
                *p_out = p0;
            if (g_stats)
                g_stats->not_conv++;
            return -C(999.0);
        }
    }

    real_t cf = cost_func(p0, H, V, n, src_offset);
    real_t cfp = cost_func_prime(p0, H, V, n);
    real_t check_x = p0 - cf / cfp;
    real_t p_final;
    if (cf < C(0.0)) {
        if (g_stats)
            g_stats->cf_negative++;
        p_final = solve_newton(p0, H, V, n, src_offset, &conv);
    } else if (check_x < C(1.0) / max_val(V, n) - (C(1e-10))) {
        if (g_stats)
            g_stats->safe_jump++;
        p_final = solve_newton(p0, H, V, n, src_offset, &conv);
    } else {
        if (g_stats)
            g_stats->bisect++;
        p_final = solve_bisect(C(1e-10), C(1.0) / max_val(V, n) - C(1e-12), H, V, n, src_offset);
        p0 = p_final;
        p_final = solve_newton(p0, H, V, n, src_offset, &conv);
    }
    real_t timeP = C(0.0);
    for (int i = 0; i < n; ++i) {
        real_t cosv = R_SQRT(C(1.0) - (p_final * p_final) * (V[i] * V[i]));
        real_t t_int = H[i] / (V[i] * cosv);
        timeP = (i == 0) ? t_int : timeP + t_int;
    }
    if (!conv) {
        timeP = -C(999.0);
        if (g_stats)
            g_stats->not_conv++;
    }
    if (p_out)
        *p_out = p_final;
    return timeP;
}

/*
 * subroutineR-quiet.f90:408-463 dofullforwardproblem ("dff_") and its twin
 * TraceRays (:467-520), without the rays.dat side effect.
 *   vels[nlayers+1], depths[nlayers] (depth TO each interface), 1-D contiguous.
 *   p_final / nl_out may be NULL; when given they receive the final ray
 *   parameter (NaN for straight rays) and the 1-based source layer, from which
 *   the rays.dat records (:157-164) can be rebuilt -- see oracle_ray_segments.
 */
void SYM(oracle_dff)(const real_t *vels, const real_t *depths, int nlayers, const real_t *src_offset,
                const real_t *src_depth, int64_t nsrc, real_t *timeP, real_t *p_final, int32_t *nl_out)
{
    real_t Hs[ORACLE_MAX_STACK], Vs[ORACLE_MAX_STACK];
    for (int64_t k = 0; k < nsrc; ++k)
        timeP[k] = -C(1.0); /* :430 */
    if (nlayers + 1 > ORACLE_MAX_STACK)
        return;
    for (int64_t k = 0; k < nsrc; ++k) {
        real_t dph = src_depth[k];
        int nl = which_layer(depths, nlayers, dph);
        if (nl_out)
            nl_out[k] = nl;
        if (nl == 1) {
            /* :94-97 straight ray in the top layer */
            timeP[k] = R_SQRT(dph * dph + src_offset[k] * src_offset[k]) / vels[0];
            if (p_final)
                p_final[k] = C(NAN);
            if (g_stats)
                g_stats->top_layer++;
            continue;
        }
        /* :38-69 InsertLayer: velocities 1..nl, interface depths differenced,
         * the last segment ends at the source depth. */
        for (int i = 0; i < nl; ++i)
            Vs[i] = vels[i];
        Hs[0] = depths[0];
        for (int i = 1; i < nl - 1; ++i)
            Hs[i] = depths[i] - depths[i - 1];
        Hs[nl - 1] = dph - depths[nl - 2];
        real_t p = C(0.0);
        timeP[k] = get_ptime_stack(dph, src_offset[k], Hs, Vs, nl, &p);
        if (p_final)
            p_final[k] = p;
    }
}

/* Same, collecting the branch statistics of SURVEY.md section 8(a). */
void SYM(oracle_dff_stats)(const real_t *vels, const real_t *depths, int nlayers, const real_t *src_offset,
                      const real_t *src_depth, int64_t nsrc, real_t *timeP, oracle_stats *stats)
{
    memset(stats, 0, sizeof(*stats));
    g_stats = stats;
    SYM(oracle_dff)(vels, depths, nlayers, src_offset, src_depth, nsrc, timeP, NULL, NULL);
    g_stats = NULL;
}

/*
 * The two rays.dat records the reference appends per ray when keep_delta > 0
 * (subroutineR-quiet.f90:157-164; top-layer variant :98-105):
 *   d[i] = sqrt((H[i]/cos_i)^2 - H[i]^2),  h[i] = H[i],  i < nl.
 * For nl == 1 the reference writes d = offset and h = depth.
 * Returns the number of segments written (nl), d and h sized >= nlayers+1.
 */
int SYM(oracle_ray_segments)(const real_t *vels, const real_t *depths, int nlayers, real_t src_offset,
                        real_t src_depth, real_t p_final, real_t *d, real_t *h)
{
    int nl = which_layer(depths, nlayers, src_depth);
    if (nl == 1) {
        d[0] = src_offset;
        h[0] = src_depth;
        return 1;
    }
    for (int i = 0; i < nl; ++i) {
        real_t Hi;
        if (i == 0)
            Hi = depths[0];
        else if (i < nl - 1)
            Hi = depths[i] - depths[i - 1];
        else
            Hi = src_depth - depths[nl - 2];
        real_t cosv = R_SQRT(C(1.0) - (p_final * p_final) * (vels[i] * vels[i]));
        real_t q = Hi / cosv;
        d[i] = R_SQRT(q * q - Hi * Hi);
        h[i] = Hi;
    }
    return nl;
}

#ifndef ORACLE_F32
#ifndef ORACLE_F32
/*
 * SUM(DresRT**2) of many models over one source set: what LOGLHOOD_RT leaves in SUM((DobsRT - DpredRT)**2) for each of
 * nmodels calls (loglhood.f90:128-146, :165-166, :195), in source order.  V / D rows of ld doubles, NL[m] interfaces.  Used
 * by the quadrature of the posterior tests (tests/golden/make_posterior_golden.py); test infrastructure like the rest.
 */
void oracle_ss_batch(const double *V, const double *D, const int32_t *NL, int ld, int64_t nmodels, const double *src_offset,
                     const double *src_depth, const double *obs, int64_t nsrc, double *ss)
{
    double *t = (double *)malloc((size_t)(nsrc > 0 ? nsrc : 1) * sizeof(double));
    if (!t)
        return;
    for (int64_t m = 0; m < nmodels; ++m) {
        oracle_dff(V + m * ld, D + m * ld, NL[m], src_offset, src_depth, nsrc, t, NULL, NULL);
        double s = 0.0;
        for (int64_t k = 0; k < nsrc; ++k) {
            const double r = obs[k] - t[k];
            s = s + r * r;
        }
        ss[m] = s;
    }
    free(t);
}
#endif

/*
 * Gaussian travel-time log-likelihood of one model, as LOGLHOOD_RT computes it
 * (/root/reference/ray_tracing_sampling/loglhood.f90:165-199, ICOV=1 branch):
 *   res = obs - pred
 *   logL = LOG(1/(2 pi)**(N/2)) - (SUM(res**2)/(2 sd**2) + N LOG(sd))
 * with one mode (NMODE = 1) and N = ndat.  The first term is evaluated the
 * reference's way, so like the reference it is -inf once (2 pi)**(N/2)
 * overflows (N > ~770); `stable` != 0 switches it to -(N/2) ln(2 pi), the form
 * the HIP path uses at every N (identical to ~1e-15 relative where both exist).
 */
double oracle_loglhood_rt(const double *obs, const double *pred, int64_t ndat, double sd, int stable)
{
    const double PI2 = 3.141592653589793238462643383279502884197; /* data_type.f90:5 */
    double ss = 0.0;
    for (int64_t i = 0; i < ndat; ++i) {
        double r = obs[i] - pred[i];
        double r2 = r * r;
        ss = (i == 0) ? r2 : ss + r2;
    }
    double norm = stable ? -((double)ndat / 2.0) * log(2.0 * PI2)
                         : log(1.0 / pow(2.0 * PI2, (double)ndat / 2.0));
    return norm - (ss / (2.0 * (sd * sd)) + (double)ndat * log(sd));
}

#endif
