/*
 * rtmi.h -- C ABI of librtmi.so: the MI355X (gfx950) implementation of the
 * 1-D layered-earth first-arrival travel-time forward model of
 * AntonBiryukovUofC/RayTracerFortran (module raymod).
 *
 * Every entry point names the reference interface it replaces (file:line under
 * /root/reference).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions shared by all entry points (they are the reference's):
 *   vels[nlayers+1]  P velocity of each layer, top -> down, last = half-space, m/s
 *   depths[nlayers]  depth TO each interface (not thicknesses), m
 *   src_offset[n]    horizontal source-receiver distance, m
 *   src_depth[n]     source depth, m
 *   timeP[n]         out: travel time, s; -999 where the reference's solver
 *                    reports non-convergence (subroutineR-quiet.f90:167-169)
 * All arithmetic is IEEE fp64 in the reference's operation order.
 */
#ifndef RTMI_H
#define RTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMI_VERSION 100      /* 0.1.0 */
#define RTMI_MAX_LAYERS 1024  /* interfaces per model the kernels accept */

/* Status codes of the rtmi_* entry points (0 = success). */
enum {
    RTMI_OK = 0,
    RTMI_ERR_NO_DEVICE = 1,   /* no usable gfx950 device / HIP runtime failure at init */
    RTMI_ERR_BAD_ARG = 2,     /* nlayers < 1, nlayers > RTMI_MAX_LAYERS, negative counts, NULL pointers */
    RTMI_ERR_HIP = 3,         /* a HIP call failed; text in rtmi_last_error() */
    RTMI_ERR_IO = 4           /* rays.dat could not be written */
};

/* ------------------------------------------------------------------------------------------
 * 1. Drop-in symbols: host pointers in, host pointers out.
 * ------------------------------------------------------------------------------------------ */

/*
 * Replaces `subroutine dofullforwardproblem(...) bind(C, name="dff_")`,
 * subroutineR-quiet.f90:408-463 -- the symbol R's
 *   .Fortran("dff", vels, depths, NLayers, src_offset, src_depth, NSrc, timeP, keep_delta)
 * resolves (rayTracerR.R:31-33).  All arguments by reference, as Fortran passes them.
 * keep_delta > 0 additionally writes ./rays.dat (two records per ray: horizontal
 * segment lengths, then segment thicknesses; subroutineR-quiet.f90:98-105,157-164).
 * No status is returned (the reference has none): on a device failure the call
 * prints the reason to stderr, fills timeP with NaN and aborts the process unless
 * RTMI_ABORT_ON_ERROR=0 is set in the environment.
 */
void dff_(const double *vels, const double *depths, const int *NLayers, double *src_offset,
          double *src_depth, const int *NSrc, double *timeP, const int *keep_delta);

/*
 * Replaces the body of `subroutine TraceRays(...)`, subroutineR-quiet.f90:467-520
 * (sampler copy ray_tracing_sampling/subroutineR-quiet.f90:459), called by
 * LOGLHOOD_RT (loglhood.f90:135,144).  TraceRays is a Fortran MODULE procedure,
 * so its link name is compiler specific; shim/raymod.f90 provides `module raymod`
 * whose TraceRays forwards here.  Same argument meaning as dff_.
 */
void rtmi_tracerays(const double *vels, const double *depths, const int *NLayers, double *src_offset,
                    double *src_depth, const int *NSrc, double *timeP, const int *keep_delta);

/*
 * 7-argument flavour: `dff_` of subroutineR.f90:405 (no keep_delta), the one
 * README.md:16-21 and adaptMCMC.Rmd:74-78 call.  Exported under its own name
 * here; librtmi_dff7.so re-exports it as `dff_` for callers that load that file.
 */
void rtmi_dff7(const double *vels, const double *depths, const int *NLayers, double *src_offset,
               double *src_depth, const int *NSrc, double *timeP);

/*
 * The likelihood as the sampler calls it: sources and observed times are read once (READDATA,
 * read_input.f90:302-387) and then LOGLHOOD is evaluated for model after model
 * (loglhood.f90:3-31 -> LOGLHOOD_RT :35-207; prjmh_temper_rf.f90:559,693,739).
 *   rtmi_set_sources     copies src_offset / src_depth / obs (observed travel times, may be NULL
 *                        if only forward calls follow) to HBM, where they stay.
 *   rtmi_loglhood_batch  HOST arrays in and out: nmodels models (row m of vels / depths holds
 *                        nlayers[m]+1 velocities / nlayers[m] interface depths, row stride
 *                        ld_model), sd[m] = data standard deviation; returns
 *                        logL[m] = -(N/2) ln 2pi - (SS/(2 sd^2) + N ln sd)   (loglhood.f90:194-196,
 *                        ICOV = 1, one mode) and, if ss != NULL, SS = SUM(DresRT**2).  One launch
 *                        for the whole batch; the travel times never leave the GPU.
 * The k = 1 model is passed the way LOGLHOOD_RT does: nlayers = 1, vels = (v, v), depths = (9999.9).
 */
int rtmi_set_sources(const double *src_offset, const double *src_depth, const double *obs, int64_t nsrc);
int rtmi_loglhood_batch(const double *vels, const double *depths, const int32_t *nlayers, int ld_model,
                        int nmodels, const double *sd, double *logL, double *ss);

/*
 * Native Metropolis-Hastings sweep of the reversible-jump sampler over the resident sources
 * (rtmi_set_sources with observations first).  One handle = the tempered chains of one process; every
 * proposal slot of a sweep -- birth/death, then each (node, parameter) perturbation -- is ONE
 * rtmi_loglhood_batch launch for all chains, decisions are taken on the host:
 *   EXPLORE_MH_NOVARPAR  prjmh_temper_rf.f90:637-761   BIRTH_FULL :997  DEATH_FULL :917  PROPOSAL :1387
 *   EXPLORE_MH           prjmh_temper_rf.f90:521-634   PROPOSAL_SDRT :1616   CHECKBOUNDS :1639
 * for the reference configuration NPL = 2, ENOS = 0, IPOIPR = 1, ICOV = 1, I_VARPAR = 0.  Chain state: k
 * nodes (depth, Vp), node 1 the surface node whose depth is not perturbed, and sigma.  The temperature
 * exchange (TEMPSWP_MH :1329) is left to the caller, which owns the collective: it reads logL with
 * rtmi_pt_get_chain and writes the new temperatures with rtmi_pt_set_beta.
 */
typedef struct rtmi_pt rtmi_pt;
typedef struct {
    int kmin, kmax;          /* NLMN, NLMX                                  read_input.f90:39-40  */
    double lambda;           /* Poisson prior on k                          read_input.f90:78-81  */
    double hmx, hmin;        /* max partition depth, min layer thickness    read_input.f90:46-47  */
    double vmin, vmax;       /* Vp bounds (1500, 10000)                     read_input.f90:207-208 */
    double sdmin, sdmax;     /* sigma bounds                                read_input.f90:234-236 */
    double pertsdsc;         /* Cauchy step = (max - min) / pertsdsc (30)   read_input.f90:169,214 */
    double pertsdsdsc;       /* sigma step = (sdmax - sdmin) / this (10)    read_input.f90:237-239 */
} rtmi_pt_prior;
/* ld_model >= kmax + 1; ndata = number of observations of the resident source set; (seed, stream_id)
 * select the random streams (one per chain; use the rank as stream_id).  NULL on bad arguments. */
rtmi_pt *rtmi_pt_create(int nchains, int ld_model, int64_t ndata, const rtmi_pt_prior *prior, uint64_t seed,
                        uint64_t stream_id);
void rtmi_pt_destroy(rtmi_pt *h);
int rtmi_pt_set_chain(rtmi_pt *h, int chain, int k, const double *depths, const double *vels, double sd,
                      double beta);
int rtmi_pt_set_beta(rtmi_pt *h, const double *beta);
int rtmi_pt_get_chain(const rtmi_pt *h, int chain, int *k, double *depths, double *vels, double *sd,
                      double *logL, double *ss);
int rtmi_pt_evaluate(rtmi_pt *h);             /* logL of the current states (after set_chain) */
int rtmi_pt_sweep(rtmi_pt *h, int nsweeps);   /* nsweeps x (EXPLORE_MH_NOVARPAR + EXPLORE_MH) for every chain */
/* out8: accepted/proposed birth-death, accepted/proposed perturbations, accepted/proposed sigma, launches, models */
int rtmi_pt_stats(const rtmi_pt *h, int64_t *out8);

/* ------------------------------------------------------------------------------------------
 * 2. Library state.
 * ------------------------------------------------------------------------------------------ */

/* Bind the calling process to HIP device `device` (default 0 if never called). */
int rtmi_init(int device);
/* Text of the last error of the calling thread ("" if none). */
const char *rtmi_last_error(void);
int rtmi_version(void);
/*
 * Sticky fault word of the device code: 0, or a bit mask if a kernel ever ran out of one of its bounded waits (the
 * hand-over ring of the fused kernel is deadlock-free by construction; the bounds exist so that a fault could never
 * hang the GPU).  The host-pointer entry points check it after their synchronise and fail; callers of the
 * asynchronous device entry points can poll it after their own synchronise.
 */
int rtmi_device_status(void);
/* Number of compute units of the bound device (256 on MI355X); 0 on failure. */
int rtmi_device_cus(void);
/*
 * Scheduling knobs of rtmi_trace_device (never change results, only speed); process-wide.
 *   "path"            0 auto (default), 1 always the plain kernel, 2 always the fused kernel
 *   "waves_per_simd"  resident waves per SIMD of the fused kernel, 1..8 (default 4)
 *   "refill_min"      idle lanes of a wave that trigger a refill, 1..64 (default 16)
 *   "tail_group"      end-game concentration: wave group g (4 waves each) stops taking new sources once
 *                     fewer than g x tail_group remain queued, 0..4096 (default 256, 0 = off)
 *   "promote_age"     waves 4.. of a workgroup hand a source over to the express waves 0..3 (one per SIMD) once it
 *                     has taken this many evaluation trips, 0..64 (default 10, 0 = off; applies to rounds of at
 *                     least two records per lane)
 *   "diag_fine"       with rtmi_debug_stamps: 1 (default) also takes the per-trip section clocks and counters, which
 *                     slow the solve loop down by about a quarter; 0 keeps to the phase and per-wave clocks
 *   "retire_max"      drain: waves 4.. hand their last <= retire_max live lanes over and leave, 0..64
 *                     (default 8, 0 = off)
 * Environment defaults: RTMI_FORCE_PATH=plain|fused, RTMI_SOLVE_WAVES, RTMI_REFILL_MIN, RTMI_TAIL_GROUP,
 * RTMI_RETIRE_MAX, RTMI_PROMOTE_AGE.
 */
int rtmi_set_option(const char *name, int value);

/* ------------------------------------------------------------------------------------------
 * 3. Device-resident entry points: every pointer is a DEVICE pointer, `stream`
 *    is a hipStream_t (NULL = default stream).  Nothing is allocated, copied or
 *    synchronised: the launch is asynchronous and graph-capturable.
 * ------------------------------------------------------------------------------------------ */

/*
 * One forward evaluation: the loop body of dff_/TraceRays
 * (subroutineR-quiet.f90:434-462: whichLayer, InsertLayer, GetPTime) for nsrc
 * sources of one model.  d_p_final may be NULL; when given it receives the final
 * ray parameter of each source (NaN for straight top-layer rays), from which the
 * rays.dat records are rebuilt on the host.
 * Source sets of RTMI_FUSED_MIN_NSRC or more take the fused large-N kernel (each workgroup sorts its
 * sources by layer in LDS and refills lanes as they finish), smaller ones the plain one-lane-per-
 * source kernel; the batch entry point switches at a quarter of that per model.  Results are
 * bit-identical either way.
 */
#define RTMI_FUSED_MIN_NSRC 8192
int rtmi_trace_device(const double *d_vels, const double *d_depths, int nlayers,
                      const double *d_src_offset, const double *d_src_depth, int64_t nsrc,
                      double *d_timeP, double *d_p_final, void *stream);

/*
 * The same forward evaluation in IEEE single precision (BASELINE config 5: "fp32 kernel path ...,
 * delta-t tolerance vs fp64 reported").  Same algorithm, operation order and constants (each rounded
 * to float) as the fp64 path; the reference has no single-precision build, so this entry point has
 * no reference output to match: tests/test_gpu_parity.py checks it bit-for-bit against the CPU
 * restatement built in single precision and reports its deviation from the fp64 result.
 */
int rtmi_trace_device_f32(const float *d_vels, const float *d_depths, int nlayers,
                          const float *d_src_offset, const float *d_src_depth, int64_t nsrc,
                          float *d_timeP, float *d_p_final, void *stream);

/*
 * Many models over one shared source set: what nmodels consecutive calls of
 * TraceRays from LOGLHOOD_RT compute (loglhood.f90:128-146), in one launch.
 *   d_vels   [nmodels][ld_model]   row m holds nlayers[m]+1 velocities
 *   d_depths [nmodels][ld_model]   row m holds nlayers[m] interface depths
 *   d_nlayers[nmodels]             int32, each in [1, RTMI_MAX_LAYERS]
 *   d_timeP  [nmodels][nsrc]       may be NULL when only logL is wanted
 * Optional fused Gaussian log-likelihood (loglhood.f90:165-199, ICOV = 1, one mode):
 *   d_obs[nsrc], d_sd[nmodels] -> d_logL[nmodels]; pass NULL for all three to skip.
 *   d_work: >= rtmi_trace_batch_work_bytes(nmodels, nsrc) bytes of scratch.
 * logL uses -(N/2) ln(2 pi) for the normalisation, which equals the reference's
 * LOG(1/(2 pi)**(N/2)) wherever that expression does not overflow (N < ~770).
 */
int rtmi_trace_batch_device(const double *d_vels, const double *d_depths, const int32_t *d_nlayers,
                            int ld_model, int nmodels, const double *d_src_offset,
                            const double *d_src_depth, int64_t nsrc, double *d_timeP,
                            const double *d_obs, const double *d_sd, double *d_logL, void *d_work,
                            void *stream);
int64_t rtmi_trace_batch_work_bytes(int nmodels, int64_t nsrc);

/*
 * Diagnostic used by the parity tests: the kernels evaluate sqrt and divide with the compiler's
 * expansions minus their exponent-range handling when all operands of a wave are inside a guarded
 * range.  This compares those forms bit for bit with the plain operators on caller-
 * supplied operands: d_mismatch[0] += sqrt mismatches over d_w, d_mismatch[1] += divide mismatches
 * over (d_num, d_den), d_mismatch[2] += mismatches of the shared-reciprocal layer term
 * (num/sqrt(w) and den/sqrt(w)^3 against sqrt and two plain divides).  d_mismatch (3 x uint64) must be
 * zeroed by the caller.  All device pointers.
 */
int rtmi_selftest_fastmath(const double *d_w, const double *d_num, const double *d_den, int64_t n,
                           unsigned long long *d_mismatch, void *stream);
/*
 * The same for the single-precision layer term (fp32 path): d_mismatch[0] += mismatches of num/sqrtf(w),
 * d_mismatch[1] += mismatches of den/sqrtf(w)^3 (2 x uint64, zeroed by the caller); guarded ranges
 * w in (1e-12, 1], num and den 0 or in (1e-18, 1e18).
 */
int rtmi_selftest_fastmath_f32(const float *d_w, const float *d_num, const float *d_den, int64_t n,
                               unsigned long long *d_mismatch, void *stream);

/*
 * Diagnostic: while d_stamps != NULL, every fused-kernel launch of rtmi_trace_device runs the instrumented build of
 * the kernel, which writes 200 words per workgroup (first round only), in shader-clock ticks:
 *   d_stamps[200*wg + 0..3]   clock at {start, solve order built, solved, times written}
 *   d_stamps[200*wg + 4..6]   clock at {bins counted, records in their slots, first evaluation done}
 *   d_stamps[200*wg + 8 + 12*w + 0..11], per wave w: {clock: queue seen empty, clock: last source finished,
 *       evaluation trips, sum of live lanes over trips, sum of layer-loop lengths over trips,
 *       useful layer terms | refill events << 40, and the ticks the wave spent in the six sections of a trip:
 *       tail check, hand-over, adoption, refill, wave maximum + exit test, evaluation}
 * (device buffer of 200 x 8 B x 2 x CUs).  NULL switches back to the production kernel, which carries no
 * instrumentation.
 */
int rtmi_debug_stamps(unsigned long long *d_stamps);

/* ------------------------------------------------------------------------------------------
 * 4. Host-side helpers of the drop-in path.
 * ------------------------------------------------------------------------------------------ */

/*
 * Write the reference's rays.dat (subroutineR-quiet.f90:98-105,157-164) for
 * sources whose final ray parameters are known.  One record per line, values
 * separated by blanks, as MiscFunctions.R:13-26 (read.rays) parses them.
 */
int rtmi_write_rays_dat(const char *path, const double *vels, const double *depths, int nlayers,
                        const double *src_offset, const double *src_depth, const double *p_final,
                        int64_t nsrc);

#ifdef __cplusplus
}
#endif
#endif /* RTMI_H */
