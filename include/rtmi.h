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

#define RTMI_VERSION 400      /* 0.4.0 */
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
 * prints the reason to stderr and fills timeP with NaN -- loud, never a plausible
 * number, and an R session that loaded the library survives it.
 * RTMI_ABORT_ON_ERROR=1 in the environment makes it abort the process instead.
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
/* Identifies the resident source set: bumped by every rtmi_set_sources (0 = none yet).  The set is ONE per process
 * (READDATA's data are module variables in the reference as well); a caller that caches anything derived from it can
 * detect that somebody else replaced it.  rtmi_pt handles do not depend on it after creation: they keep their own copy. */
uint64_t rtmi_sources_generation(void);
int rtmi_loglhood_batch(const double *vels, const double *depths, const int32_t *nlayers, int ld_model,
                        int nmodels, const double *sd, double *logL, double *ss);

/*
 * The Metropolis-Hastings sweep of the reversible-jump tempering sampler, RESIDENT ON THE GPU (csrc/rtmi_pt.hip).
 * One handle = the tempered chains of one process (one GPU).  Chain state -- k nodes (depth, Vp), node 1 the surface
 * node whose depth is not perturbed, sigma, logL, SS, beta, the position of the chain's counter-based random stream --
 * lives in HBM; every proposal slot of a sweep (birth/death, then each (node, parameter) perturbation) is a step
 * (decide the previous slot, propose the next) plus the likelihood of every chain's proposal over the handle's source
 * set; nothing of a sweep runs on the host.  A sweep is either ONE persistent cooperative launch in which the workgroups
 * of a chain meet among themselves slot by slot, so chains run ahead of each other (needs the fused kernel path and all
 * workgroups resident), or two launches per slot replayed as a captured hipGraph (RTMI_PT_GRAPH=0: launched one by one).
 * The engines give bit-identical chains; which one a handle uses is measured once: in the first rtmi_pt_run call of
 * >= 12 sweeps, as soon as the sweeps have their full length (the bound of k that sizes a sweep has reached kmax) and ten
 * more are to go, five sweeps run persistent and five as per-slot launches -- one warm sweep, then a block of four timed
 * as a whole, one event synchronisation in all -- and the persistent launch stays unless its block took more than 1.08x
 * the time per proposal slot (it does, several times over, when other processes share the GPU).  RTMI_PT_PERSISTENT=0 / 1
 * decides up front instead.
 * The likelihood of a proposal is INCREMENTAL: a birth, death or perturbation changes the model from some depth
 * downwards, and a source above that depth has, bit for bit, the travel time it has under the chain's current model --
 * its squared residual is taken from a per-chain cache in HBM (2 x nchains x ndata doubles) instead of being traced
 * again.  The sampler's sums of squared residuals are ORDER-INDEPENDENT (integer sums on a fixed 2^-64 s^2 grid -- addends below 2^32 s^2, so that the reference's -999 sentinel stays a finite residual --, per lane, wave, workgroup),
 * so their value does not depend on the order of the additions: logL and the chains are those of the full evaluation
 * (RTMI_PT_INCREMENTAL=0 evaluates every proposal over all sources), whatever the launch geometry, knobs or engine.
 *   EXPLORE_MH_NOVARPAR  prjmh_temper_rf.f90:637-761   BIRTH_FULL :997  DEATH_FULL :917  PROPOSAL :1387
 *   EXPLORE_MH           prjmh_temper_rf.f90:521-634   PROPOSAL_SDRT :1616   CHECKBOUNDS :1639
 *   TEMPSWP_MH           prjmh_temper_rf.f90:1329-1357 (replaces the pairing through rank 0, :331-349)
 * for the reference configuration NPL = 2, ENOS = 0, IPOIPR = 1, ICOV = 1, I_VARPAR = 0.  Chain-level parity with the
 * reference is statistical (its RANDOM_NUMBER streams are compiler specific); every logL the sampler acts on is the
 * bit-exact forward model.
 */
typedef struct rtmi_pt rtmi_pt;
typedef struct rtmi_comm rtmi_comm;
typedef struct {
    int kmin, kmax;          /* NLMN, NLMX (kmax <= 64)                     read_input.f90:39-40  */
    double lambda;           /* Poisson prior on k                          read_input.f90:78-81  */
    double hmx, hmin;        /* max partition depth, min layer thickness    read_input.f90:46-47  */
    double vmin, vmax;       /* Vp bounds (1500, 10000)                     read_input.f90:207-208 */
    double sdmin, sdmax;     /* sigma bounds                                read_input.f90:234-236 */
    double pertsdsc;         /* Cauchy step = (max - min) / pertsdsc (30)   read_input.f90:169,214 */
    double pertsdsdsc;       /* sigma step = (sdmax - sdmin) / this (10)    read_input.f90:237-239 */
} rtmi_pt_prior;
/* Creates the chains over the source set currently resident (rtmi_set_sources with observations first): the handle
 * takes its OWN device copy of that set, so a later rtmi_set_sources -- or a second handle over other data -- does
 * not retarget it.  ndata must equal the resident set's size; ld_model >= kmax + 1; (seed, stream_id, chain) select
 * the random streams (use the rank as stream_id).  NULL on bad arguments (text in rtmi_last_error). */
rtmi_pt *rtmi_pt_create(int nchains, int ld_model, int64_t ndata, const rtmi_pt_prior *prior, uint64_t seed,
                        uint64_t stream_id);
void rtmi_pt_destroy(rtmi_pt *h);
int rtmi_pt_set_chain(rtmi_pt *h, int chain, int k, const double *depths, const double *vels, double sd,
                      double beta);
int rtmi_pt_set_beta(rtmi_pt *h, const double *beta);
int rtmi_pt_get_chain(const rtmi_pt *h, int chain, int *k, double *depths, double *vels, double *sd,
                      double *logL, double *ss);
/* every chain in one read-back: k[nchains], depths/vels[nchains][kmax], sd/logL/ss/beta[nchains]; any may be NULL */
int rtmi_pt_get_all(rtmi_pt *h, int *k, double *depths, double *vels, double *sd, double *logL, double *ss,
                    double *beta);
int rtmi_pt_evaluate(rtmi_pt *h);             /* logL of the current states (after set_chain) */
int rtmi_pt_sweep(rtmi_pt *h, int nsweeps);   /* = rtmi_pt_run(h, nsweeps, 0, NULL, 0) */
/* out8: accepted/proposed birth-death, accepted/proposed perturbations, accepted/proposed sigma, likelihood launches,
 * models evaluated */
int rtmi_pt_stats(const rtmi_pt *h, int64_t *out8);
/* the same counters chain by chain (the reference keeps them per chain object: obj%iaccept_bd ..., prjmh_temper_rf.f90:1880-1883):
 * out[nchains][8] = accepted/proposed birth-death, accepted/proposed perturbations, accepted/proposed sigma, models of the
 * chain that went through the likelihood and were decided, speculated models evaluated and then discarded */
int rtmi_pt_chain_stats(const rtmi_pt *h, int64_t *out);
/* incremental likelihood: sources of proposals traced / taken from the residual cache so far */
int rtmi_pt_likelihood_counts(const rtmi_pt *h, int64_t *traced, int64_t *cached);
/* position of every chain's counter-based random stream (draws taken so far), ctr[nchains]: part of a chain's state --
 * two runs whose chains AND stream positions agree have taken the same draws for the same decisions */
int rtmi_pt_stream_positions(rtmi_pt *h, uint64_t *ctr);
/* The speculative step (RTMI_PT_SPECULATE = model rows per chain, at most 8; 0 = one proposal at a time; default: 6 where a
 * chain's workgroups hold few sources each and the prior has kmax <= 16, 2 for kmax <= 32, 0 where a pass is bound by
 * throughput).  The chain's random stream is counter-based, so what it does in the next slots can be laid out ahead of the
 * likelihoods along a predicted path: a proposal outside the prior is rejected; one that changes the model only below the
 * deepest source leaves every residual as it is, so its outcome is known when it is drawn; one that changes travel times gets
 * a model row and is assumed rejected.  The rows of a window are evaluated side by side in one likelihood pass and decided in
 * order with the draws the one-at-a-time sampler would take -- the first accepted one becomes the state, everything behind it
 * is discarded.  Chains, statistics and stream positions are bit-identical to RTMI_PT_SPECULATE=0 (the sums are exact).
 * Where every chain of the ladder is local and sits on a level of its own, rtmi_pt_run puts many sweeps into one launch with
 * the temperature exchange inside the kernel (a chain meets only its partner of the round; RTMI_PT_MULTI=0: one launch per
 * sweep) -- same chains, same exchanges.
 * out4: rows per chain (1 = off), models evaluated and then discarded, engine in use (-1 undecided, 0 per-slot launches,
 * 1 persistent launch), 1 if the handle's source set is kept sorted by depth. */
int rtmi_pt_speculation(rtmi_pt *h, int64_t *out4);

/*
 * Temperature ladder and exchange.  level[c] = temperature level of GLOBAL chain c (replicated on every rank),
 * beta = beta_of_level[level]; this handle holds global chains first_chain .. first_chain + nchains - 1.
 *   rtmi_pt_swap_round       one exchange: all-gather of the local logL over `comm` (RCCL; NULL when every chain is
 *                            local), then the deterministic adjacent-level pairing + Metropolis rule on the device,
 *                            identical on every rank, swapping LEVELS (temperatures) in place -- no model state moves.
 *                            Message size: 8 B per chain per round (256 chains: 2 KB).  Asynchronous on the handle's stream.
 *   rtmi_pt_swap_round_host  the same with the gather done by the caller (MPI_Allgather, gloo ...): logl_all[nchains_total].
 *   rtmi_pt_swap_decide      the pairing + acceptance rule alone, on the host, no GPU: what the device runs.
 *   rtmi_pt_run              nsweeps sweeps; a swap round after every swap_every-th sweep (0 = never); the chains at
 *                            T = 1 appended to the sample ring after every record_every-th sweep of this call (0 = never).  The
 *                            whole run is enqueued without host synchronisation; the call returns after ONE synchronise.
 */
int rtmi_pt_set_ladder(rtmi_pt *h, int nchains_total, int first_chain, const int *level, const double *beta_of_level,
                       int nlevels);
int rtmi_pt_get_ladder(rtmi_pt *h, int *level, int64_t *swap_stats2 /* accepted, proposed */);
/* New inverse temperatures for the ladder's levels, in place: the assignment of chains to levels, the exchange's counters
 * and its position in the random stream are kept.  For the burn-in adaptation of prjmh_temper_rf.f90:363-383 (dTlog x 1.02 /
 * x 0.98 on the rolling exchange acceptance), which raymod / tempering.py offer as an option. */
int rtmi_pt_set_ladder_betas(rtmi_pt *h, const double *beta_of_level, int nlevels);
int rtmi_pt_swap_round(rtmi_pt *h, rtmi_comm *comm);
int rtmi_pt_swap_round_host(rtmi_pt *h, const double *logl_all);
int rtmi_pt_swap_decide(int nchains_total, const double *logl_all, int *level, const double *beta_of_level, int nlevels,
                        uint64_t seed, uint64_t round, int64_t *proposed);
int rtmi_pt_run(rtmi_pt *h, int nsweeps, int swap_every, rtmi_comm *comm, int record_every);
/* sample ring of the T = 1 chains (device memory, `capacity` samples; 0 frees it) and its read-out: returns the number
 * of samples copied (<= max_samples) and empties the ring; depths/vels are [max_samples][kmax]; *dropped = samples
 * that did not fit since the last read-out.  Negative = -status. */
int rtmi_pt_sample_buffer(rtmi_pt *h, int capacity);
int rtmi_pt_samples(rtmi_pt *h, int max_samples, int64_t *sweep, int *chain, int *k, double *sd, double *logL,
                    double *depths, double *vels, int *dropped);

/*
 * The communicator of the exchange: RCCL over xGMI, one rank per GPU (the bound device of rtmi_init).  Rank 0 makes
 * the id, the host program broadcasts its RTMI_COMM_ID_BYTES bytes (MPI_Bcast in the reference's world), every rank
 * calls rtmi_comm_create.  Replaces the MPI_SEND/MPI_RECV pairs of prjmh_temper_rf.f90:331-349.
 */
#define RTMI_COMM_ID_BYTES 128
int rtmi_comm_unique_id(char id[RTMI_COMM_ID_BYTES]);
rtmi_comm *rtmi_comm_create(int nranks, int rank, const char id[RTMI_COMM_ID_BYTES]);
void rtmi_comm_destroy(rtmi_comm *comm);
int rtmi_comm_size(const rtmi_comm *comm);
/* which RCCL the communicator is: ncclGetVersion (-1 on failure) and the path of the shared object ncclAllGather resolves to
 * inside this library (a process that holds torch carries a second copy; load order decides) */
int rtmi_comm_rccl_version(void);
int rtmi_comm_rccl_library(char *path, int cap);
int rtmi_comm_rank(const rtmi_comm *comm);
/* all-gather of `count` doubles per rank, device pointers, asynchronous on `stream` */
int rtmi_comm_allgather_f64(rtmi_comm *comm, const double *d_send, double *d_recv, int64_t count, void *stream);

/* ------------------------------------------------------------------------------------------
 * 2. Library state.
 * ------------------------------------------------------------------------------------------ */

/* Bind the calling process to HIP device `device` (default 0 if never called). */
int rtmi_init(int device);
/* Text of the last error of the calling thread ("" if none). */
const char *rtmi_last_error(void);
int rtmi_version(void);
/*
 * Sticky fault word of the device code: 0, or a bit mask: bits 1, 2, 4 = a kernel ran out of one of its bounded waits
 * (the hand-over ring of the fused kernel is deadlock-free by construction; the bounds exist so that a fault could never
 * hang the GPU; bit 16: a wait between the workgroups of the persistent sweep kernel -- the sweep was abandoned, the
 * handle's chains are not to be trusted and its abort word stays set: DESTROY the handle, rtmi_clear_device_status does not
 * revive it); bit 8 = a row of rtmi_trace_batch_device had d_nlayers outside [1, ld_model - 1].  The host-pointer
 * entry points check it after their synchronise and fail; callers of the asynchronous device entry points can poll it
 * after their own synchronise.  The word is allocated by rtmi_init or by the first un-captured call of any entry
 * point; a FIRST call made inside a stream capture is refused with RTMI_ERR_BAD_ARG (call rtmi_init before capturing).
 */
int rtmi_device_status(void);
/* Resets the fault word (after the caller has dealt with what it reported -- for bit 16: after the sampler handle that
 * raised it was destroyed); returns the value it had. */
int rtmi_clear_device_status(void);
/* Number of compute units of the bound device (256 on MI355X); 0 on failure. */
int rtmi_device_cus(void);
/* Name of the kernel the calling thread's last forward / batch launch dispatched, spelled as rocprofv3 --kernel-trace prints it
 * ("" before the first launch): lets a measurement name the kernel it timed instead of assuming it. */
const char *rtmi_last_kernel_name(void);
/*
 * Scheduling knobs of rtmi_trace_device (never change results, only speed); process-wide.
 *   "path"            0 auto (default), 1 always the plain kernel, 2 always the fused kernel
 *   "waves_per_simd"  resident waves per SIMD of the fused kernel, 1..8 (default 4)
 *   "refill_min"      idle lanes of a wave that trigger a refill, 1..64 (default 32)
 *   "tail_group"      end-game concentration: wave group g (4 waves each) stops taking new sources once
 *                     fewer than g x tail_group remain queued, 0..4096 (default 128; 0 = off)
 *   "promote_age"     waves 4.. of a workgroup hand a source over to the express waves 0..3 (one per SIMD) once it
 *                     has taken this many evaluation trips, 0..64 (default 3, 0 = off; applies to rounds of at
 *                     least two records per lane)
 *   "diag_fine"       with rtmi_debug_stamps: 1 (default) also takes the per-trip section clocks and counters, which
 *                     slow the solve loop down by about a quarter; 0 keeps to the phase and per-wave clocks
 *   "retire_max"      drain: waves 4.. hand their last <= retire_max live lanes over and leave, 0..64
 *                     (default 12, 0 = off)
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
 * Single precision, FAST (opt-in; BASELINE.json configs[4]).  Not the reference's iteration: the same equations
 * (costFunc subroutineR-quiet.f90:184-202, the travel-time sum :156-166) over the same stack of layers (whichLayer :9-32,
 * InsertLayer :38-69), solved to single-precision roundoff by a monotone Newton iteration in s = tan(theta) of the stack's
 * fastest layer, where 1 - p^2 V^2 has no cancellation (csrc/rtmi_f32fast.hip).  Results differ from the reference by what
 * the reference's own stopping rule leaves (it stops 0.1 m short of the root: median 2e-7 s, 99.9 % below 4e-5 s; where
 * the reference gives up or accepts a non-converged iterate -- its -999 and its rare gross errors -- this entry point
 * returns the travel time of the actual ray), and from a converged double-precision solution by < 1e-6 s.  About five
 * times the throughput of the double-precision path.  Preconditions: at most 63 interfaces, in non-decreasing order from a
 * depth >= 0, velocities > 0; a model that violates them gets NaN outputs and bit 8 of rtmi_device_status().  -999 is
 * returned for a source whose ray does not exist (NaN / Inf data; a stack whose fastest layer has no thickness and cannot
 * reach the offset).  d_p_final (may be NULL) receives the ray parameter (NaN for a source in the top layer).
 */
int rtmi_trace_device_f32_fast(const float *d_vels, const float *d_depths, int nlayers,
                               const float *d_src_offset, const float *d_src_depth, int64_t nsrc,
                               float *d_timeP, float *d_p_final, void *stream);

/*
 * Many models over one shared source set: what nmodels consecutive calls of
 * TraceRays from LOGLHOOD_RT compute (loglhood.f90:128-146), in one launch.
 *   d_vels   [nmodels][ld_model]   row m holds nlayers[m]+1 velocities
 *   d_depths [nmodels][ld_model]   row m holds nlayers[m] interface depths
 *   d_nlayers[nmodels]             int32, each in [1, ld_model - 1]; the kernels check it: a row outside that range
 *                                  gets NaN travel times / NaN logL and sets bit 8 of rtmi_device_status()
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
 * Extended form for the constructed hard cases of tests/test_gpu_fastmath_hard.py: d_mismatch64[0..2] as above,
 * d_mismatch64[3] += layer-term mismatches counted WITHOUT regard to the term's own guard (the all-ones-divisor test of
 * what the kernels would return with that guard compiled out), d_mismatch64[4] += operands the
 * guard rejected (for which the kernels take the plain operators), d_mismatch64[5] += operands that mismatched
 * (guard ignored), whose first 58 indices are stored in d_mismatch64[6..63] (bit 63 set: the guard had accepted it).  64 x uint64, zeroed by the caller.
 */
int rtmi_selftest_fastmath_ex(const double *d_w, const double *d_num, const double *d_den, int64_t n,
                              unsigned long long *d_mismatch64, void *stream);
/*
 * The VALUES of the kernels' arithmetic building blocks, for comparison with an IEEE host (the tests use numpy; the
 * GPU's own `/` is not a valid checker: hipcc's fp64 divide is faithful but not always correctly rounded):
 *   d_div         num / den as the kernels divide outside their fast forms (compiler's quotient + one repair step)
 *   d_div_guarded the guarded divide of the fast paths; NaN where its guard rejects the divisor
 *   d_sqrt        the guarded square root of w
 *   d_tf, d_tp    the shared-reciprocal layer term num / sqrt(w), den / sqrt(w)^3; NaN where the term's guard rejects
 * Any output pointer may be NULL.  All device pointers.
 */
int rtmi_selftest_values(const double *d_w, const double *d_num, const double *d_den, int64_t n, double *d_div,
                         double *d_div_guarded, double *d_sqrt, double *d_tf, double *d_tp, void *stream);
int rtmi_selftest_values_f32(const float *d_w, const float *d_num, const float *d_den, int64_t n, float *d_div,
                             float *d_tf, float *d_tp, void *stream);
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
