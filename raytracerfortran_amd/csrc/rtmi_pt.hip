// rtmi_pt.hip -- the Metropolis-Hastings sweep of the reversible-jump tempering sampler, resident on the GPU.
//
// What the reference does per MCMC step and chain (one MPI rank each):
//   EXPLORE_MH_NOVARPAR   birth / death / perturbation of every (node, parameter)   prjmh_temper_rf.f90:637-761
//   EXPLORE_MH            sigma proposal                                            prjmh_temper_rf.f90:521-634
//   TEMPSWP_MH            temperature exchange between two chains                   prjmh_temper_rf.f90:1329-1357
// with LOGLHOOD -> TraceRays behind every decision (loglhood.f90:128-199).
//
// Here the chains of one process (one GPU) live on the GPU and NOTHING of a sweep runs on the host:
//   * chain state (k, node depths and velocities, sigma, logL, SS, beta, the position of the chain's random stream)
//     lives in HBM for the lifetime of the handle;
//   * a proposal slot is a step (rtmi_pt_dev.hpp::pt_step_group, one lane group per chain: accept/reject the previous
//     slot from the residual sums the likelihood left behind, then draw this slot's proposal, check CHECKBOUNDS and
//     write the proposed model as a row of the batch) and the likelihood of every proposed model over the handle's
//     resident sources (fused kernel body of rtmi_kernels.hip; rows of chains without a valid proposal are inactive);
//   * the likelihood is incremental: the step also tells it from which depth downwards the proposal changes the model
//     (rtmi_kernels.h::PtInc); sources above keep the squared residuals of the chain's current model, which live in
//     HBM next to the chain (r2cur / r2prop) -- bit-identical to evaluating every proposal over all sources;
//   * a sweep = birth/death slot + one slot per (node, parameter) + the sigma move.  Two engines, same chains: ONE
//     persistent cooperative launch whose workgroups meet chain by chain (rtmi_kernels.hip::pt_sweep_kernel), or two
//     launches per slot (pt_step_kernel + the batched likelihood launch) captured once per slot count as a hipGraph and
//     replayed; the first long rtmi_pt_run times both and keeps the faster (rtmi_pt_run below);
//   * the temperature exchange is one all-gather of logL (RCCL, rtmi_comm.cpp) plus pt_swap_kernel, and
//     pt_record_kernel appends the T = 1 chains to a sample ring -- so rtmi_pt_run enqueues whole runs without a host
//     synchronisation (but for the one of the engine trial, once per handle).
// The rules themselves (proposals, prior, bounds, acceptance, pairing) are in rtmi_pt_rules.h, shared with the host.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "rtmi_host.h"
#include "rtmi_pt_dev.hpp"
#include "rtmi_pt_rules.h"

using namespace rtmi_host;
using namespace rtmi_pt_dev;
namespace R = rtmi_pt_rules;

namespace {

__device__ __forceinline__ void load_nodes(const double *src, double *dst, int k)
{
    for (int i = 0; i < k; ++i)
        dst[i] = src[i];
}

// Working arrays of pt_pack_current_kernel (one thread per chain) live in LDS, not in private (scratch) memory: the
// rules index them dynamically.  Layout: thread t owns [t * stride, (t + 1) * stride).
#define PT_BLOCK 64
#ifndef RTMI_PT_SPECULATE_MAX_SHARE
#define RTMI_PT_SPECULATE_MAX_SHARE 4096  // sources per workgroup of a chain up to which the handle speculates by default (RTMI_PT_SPECULATE overrides)
#endif

// pt_step_kernel: the sampler's step (rtmi_pt_dev.hpp::pt_step_group) for every chain, one group of G lanes per chain.
template <int G>
__global__ __launch_bounds__(PT_STEP_THREADS) void pt_step_kernel(PtDev d, int decide, int sigma, int propose, int ivo, int ipar)
{
    constexpr int CPB = PT_STEP_THREADS / G;  // chains per block
    const int c = blockIdx.x * CPB + threadIdx.x / G;
    if (c >= d.nchains)
        return;  // whole groups leave together; there is no block-level barrier below
    pt_step_group<G>(d, c, decide, sigma, propose, ivo, ipar);
}

// The speculative step (rtmi_pt_dev.hpp::pt_step_spec): a window of up to d.nspec proposals per chain and pass.
template <int G>
__global__ __launch_bounds__(PT_STEP_THREADS) void pt_step_spec_kernel(PtDev d, int begin, int eval)
{
    const int c = blockIdx.x * (PT_STEP_THREADS / 64) + threadIdx.x / 64;  // one WAVE per chain: its lane groups take the window's slots side by side
    if (c >= d.nchains)
        return;
    pt_step_spec<G>(d, c, begin, eval, 0);
}

// The current state of every chain as batch rows (rtmi_pt_evaluate): row chain * nspec, evaluated in full.
__global__ __launch_bounds__(PT_BLOCK) void pt_pack_current_kernel(PtDev d)
{
    extern __shared__ __attribute__((aligned(16))) double pt_lds[];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nchains)
        return;
    const int k = d.k[c];
    double *qz = pt_lds + (size_t)threadIdx.x * d.K;
    double *qv = pt_lds + (size_t)PT_BLOCK * d.K + (size_t)threadIdx.x * d.K;
    int *ord = reinterpret_cast<int *>(pt_lds + (size_t)2 * PT_BLOCK * d.K) + (size_t)threadIdx.x * d.K;
    load_nodes(d.z + (size_t)c * d.K, qz, k);
    load_nodes(d.v + (size_t)c * d.K, qv, k);
    R::sorted_order(qz, k, ord);
    const int m = c * d.nspec;
    d.NL[m] = R::pack_model(qz, qv, k, ord, d.V + (size_t)m * d.ld_model, d.D + (size_t)m * d.ld_model);
    d.pactive[m] = 2;
    rtmi::PtInc *o = d.inc + m;
    o->zthr = -INFINITY;  // every source is evaluated (and the residual cache filled)
    o->dl1 = o->dl2 = 0.0;
    o->n0 = 0;
    if (d.nspec > 1 || d.spec_step) {
        // the window of the speculative step: one entry, the chain's state; the other rows are inactive
        for (int e = 1; e < d.nspec; ++e)
            d.NL[m + e] = 0;
        d.nwin[c] = 1;
        d.nact[c] = 1;
        d.wend[c] = 0;
        d.winv[c] = 0;
        d.pend_i[4 * c + 2] = 0;
        d.pslot[m] = 0;
        d.alist[m] = 0;
        d.rowwg[m] = d.gx_launch;
        o->pending = d.pend_i[4 * c + 0];
        o->pend_idx = d.pend_i[4 * c + 1];
        o->mover = 1;
        o->cur_row = c;
        o->prop_row = d.nchains + (d.wbuf[c] * d.nchains + c) * d.nspec;
        rtmi::PtPend *pp = d.pinfo + (size_t)(d.wbuf[c] * d.nchains + c) * d.nspec;  // (the evaluation is "accepted")
        pp->zthr = -INFINITY;
        pp->dl1 = pp->dl2 = 0.0;
        pp->n0 = 0;
        pp->prop_row = o->prop_row;
        unsigned long long *ac = d.acc + 4 * (size_t)m * d.part_stride;
        ac[0] = ac[1] = ac[2] = ac[3] = 0ull;
    }
}

// Temperature exchange: logL, levels and the ladder are staged in LDS, thread 0 runs the shared deterministic pairing
// over ALL chains (every rank the same), then every thread writes the levels back and refreshes the temperature of
// its local chains.  logl_local != NULL: every chain is local and its logL is read in place (no gather needed).
__global__ void pt_swap_kernel(int nct, int first, int nloc, const double *logl_all, const double *logl_local, int *level,
                               const double *beta_of_level, int nlevels, uint64_t seed, uint64_t round,
                               long long *swstats, double *beta)
{
    extern __shared__ __attribute__((aligned(16))) double pt_lds[];
    double *s_logl = pt_lds;
    double *s_beta = s_logl + nct;
    int *s_level = reinterpret_cast<int *>(s_beta + nlevels);
    int *s_order = s_level + nct;
    for (int i = threadIdx.x; i < nct; i += blockDim.x) {
        s_logl[i] = logl_local ? logl_local[i] : logl_all[i];
        s_level[i] = level[i];
    }
    for (int i = threadIdx.x; i < nlevels; i += blockDim.x)
        s_beta[i] = beta_of_level[i];
    __syncthreads();
    // one thread per chain: its position in the (level, index) order; then one thread per pair (the pairs are disjoint)
    for (int i = threadIdx.x; i < nct; i += blockDim.x)
        s_order[R::swap_position(nct, s_level, i)] = i;
    __syncthreads();
    const int par = (int)(round & 1ull);
    for (int j = threadIdx.x; par + 2 * j + 1 < nct; j += blockDim.x) {
        const int a = par + 2 * j;
        const bool acc = R::swap_pair(a, s_order, s_logl, s_level, s_beta, seed, round);
        if (acc) {
            const int c1 = s_order[a], c2 = s_order[a + 1];
            const int t = s_level[c1];
            s_level[c1] = s_level[c2];
            s_level[c2] = t;
        }
        atomicAdd((unsigned long long *)&swstats[1], 1ull);
        if (acc)
            atomicAdd((unsigned long long *)&swstats[0], 1ull);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nct; i += blockDim.x)
        level[i] = s_level[i];
    for (int c = threadIdx.x; c < nloc; c += blockDim.x)
        beta[c] = s_beta[s_level[first + c]];
}

// Append the chains that are at T = 1 (level 0) to the sample ring: {sweep, chain, k, sd, logL, z[K], v[K]}.
__global__ void pt_record_kernel(PtDev d, const int *level, int first, long long sweep, int cap, int *count,
                                 long long *s_meta, double *s_val)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nchains)
        return;
    if (level ? level[first + c] != 0 : d.beta[c] != 1.0)
        return;
    const int i = atomicAdd(count, 1);
    if (i >= cap)
        return;  // ring full: counted, dropped (the host sees count > capacity)
    const int k = d.k[c];
    s_meta[(size_t)i * 3 + 0] = sweep;
    s_meta[(size_t)i * 3 + 1] = first + c;
    s_meta[(size_t)i * 3 + 2] = k;
    double *o = s_val + (size_t)i * (2 + 2 * d.K);
    o[0] = d.sd[c];
    o[1] = d.logL[c];
    for (int j = 0; j < k; ++j) {
        o[2 + j] = d.z[(size_t)c * d.K + j];
        o[2 + d.K + j] = d.v[(size_t)c * d.K + j];
    }
}

}  // namespace

struct rtmi_pt {
    std::mutex mu;
    int device = 0, cus = 0;
    hipStream_t stream = nullptr;
    PtDev d;
    int nchains = 0, K = 0, ld_model = 0, max_nl = 1;
    int64_t nsrc = 0;
    // the handle's own copy of the source set (offset | depth | obs)
    double *d_set = nullptr;
    // one slab for the doubles, one for the integers
    double *d_f64 = nullptr;
    int *d_i32 = nullptr;
    unsigned long long *d_u64 = nullptr;
    double *d_partial = nullptr, *d_logpk = nullptr;
    // ladder (temperature exchange)
    int nct = 0, first = 0, nlevels = 0;
    int *d_level = nullptr, *d_order = nullptr;
    double *d_logl_all = nullptr, *d_beta_of_level = nullptr;
    long long *d_swstats = nullptr;
    uint64_t round = 0;
    // sample ring
    int s_cap = 0;
    int *d_s_count = nullptr;
    long long *d_s_meta = nullptr;
    double *d_s_val = nullptr;
    long long sweeps_done = 0;
    // captured sweeps, by number of node slots
    std::map<int, hipGraphExec_t> graphs;
    bool use_graph = true;
    // the sweep as one persistent cooperative launch (pt_sweep_kernel); falls back to the per-slot launches for good
    // the first time the launch is refused (workgroups not co-resident, LDS, the plain-kernel path selected)
    bool use_persistent = true;
    // which engine runs the sweeps: -1 = not decided yet (the first rtmi_pt_run of >= 6 sweeps times a sweep of each --
    // a persistent kernel is the wrong choice when other processes share the GPU: its waiting workgroups get
    // time-sliced against them), 0 = per-slot launches, 1 = persistent launch.  RTMI_PT_PERSISTENT=0 / 1 decides up front.
    int engine = -1;
    unsigned int *d_sync = nullptr;  // flag[nchains] | count[nchains] | abort
    PtDev *d_ptdev = nullptr;        // device copy of `d` (the persistent kernel's step reads it from memory)
    // incremental likelihood: squared residuals of every chain under its current model / under the proposal being evaluated
    double *d_r2 = nullptr;          // [2][nchains][nsrc]
    rtmi::PtInc *d_inc = nullptr;    // [nchains] what the step tells the likelihood about each chain's proposal
    int *d_cache_valid = nullptr;    // [nchains]
    unsigned long long *d_inc_counts = nullptr;  // [nchains][gx][2] sources traced / taken from the cache (statistics)
    // exact per-workgroup sums of squared residuals (Acc192), when the handle's launches take the fused kernel; NULL for
    // source sets small enough for the plain kernel, whose sums are doubles in d_partial
    unsigned long long *d_partial_x = nullptr;   // [nchains][gx][4]
    bool incremental = true;         // RTMI_PT_INCREMENTAL=0: every proposal is evaluated over all sources
    // speculative proposals (rtmi_pt_dev.hpp::pt_step_spec): nspec model rows per chain; spec = the speculative step is in use
    int nspec = 1, nrows = 0;
    bool spec = false;
    int gx_launch = 0;               // workgroups per row of the per-slot engine's batched launch
    int *d_cdf = nullptr;            // depth histogram of the (sorted) source set
    int *d_spec_i32 = nullptr;       // cur | fin | nwin | wend | winv | nact | wbuf | pend_i[4] (per chain), pslot | pinv | rowwg | alist (per row)
    unsigned long long *d_spec_u64 = nullptr;  // wctr (per chain), pctr (per row)
    double *d_pend_z = nullptr;      // [nchains][3]
    rtmi::PtPend *d_pinfo = nullptr; // [2][rows]
    unsigned long long *d_qstats = nullptr;  // RTMI_PT_QSTATS: where the workgroups' time went (printed by rtmi_pt_run)
    int *d_walloc = nullptr;         // [2][nchains] (rtmi_pt_dev.hpp::PtDev::walloc)
    int sweep_parity = 0;
    rtmi::PtSweepArgs *d_wprop = nullptr;  // device copy of the proportional engine's words ...
    rtmi::PtSweepArgs wprop;               // ... and the host's (the words of a multi-sweep launch are rewritten before it)
    bool prefer_fused = false;             // the handle's launches take the fused kernel although its source set is below the batch path's threshold (rtmi_pt_create)
    bool levels_perm = false;              // the ladder has one chain per level, levels 0 .. nct - 1 (set_ladder): a chain's level IS its position
    double *d_mb_logl = nullptr;           // [4][nct] | the exchange's mailboxes (multi-sweep launches, rtmi_kernels.hip::prop_exchange_post / _complete)
    int *d_mb_chain = nullptr;             // [4][nct] |
    unsigned long long *d_mb_seq = nullptr;  // [4][nct] |
    std::vector<rtmi::PtInc> inc0;   // (one-at-a-time step) the rows of the residual slab never change: a chain's record as reset by rtmi_pt_set_chain
    unsigned int epoch = 0;          // slots run through the persistent kernel so far
    int kbound = 1;  // upper bound of max k over the chains, known to the host without a read-back
    int64_t launches = 0, models = 0;
};

namespace {

// the calling thread routes this handle's launches to the fused kernel (rtmi_host.h::set_path_override) while in scope
struct PathScope {
    int old;
    explicit PathScope(bool on) : old(path_override())
    {
        if (on)
            set_path_override(2);
    }
    ~PathScope() { set_path_override(old); }
    PathScope(const PathScope &) = delete;
    PathScope &operator=(const PathScope &) = delete;
};

int blocks_for(int n) { return (n + PT_BLOCK - 1) / PT_BLOCK; }
size_t pack_lds(const rtmi_pt *h)
{
    return (size_t)PT_BLOCK * (size_t)h->K * (2 * sizeof(double) + sizeof(int));
}
// pt_step_kernel with the narrowest lane group that holds K nodes
void launch_step(rtmi_pt *h, int decide, int sigma, int propose, int ivo, int ipar)
{
    const int G = h->K <= 16 ? 16 : (h->K <= 32 ? 32 : 64);
    const int cpb = PT_STEP_THREADS / G;
    const dim3 g((h->nchains + cpb - 1) / cpb), b(PT_STEP_THREADS);
    if (G == 16)
        hipLaunchKernelGGL(pt_step_kernel<16>, g, b, 0, h->stream, h->d, decide, sigma, propose, ivo, ipar);
    else if (G == 32)
        hipLaunchKernelGGL(pt_step_kernel<32>, g, b, 0, h->stream, h->d, decide, sigma, propose, ivo, ipar);
    else
        hipLaunchKernelGGL(pt_step_kernel<64>, g, b, 0, h->stream, h->d, decide, sigma, propose, ivo, ipar);
}

// pt_step_spec_kernel, likewise
void launch_step_spec(rtmi_pt *h, int begin, int eval)
{
    const int G = h->K <= 16 ? 16 : (h->K <= 32 ? 32 : 64);
    const int cpb = PT_STEP_THREADS / 64;
    const dim3 g((h->nchains + cpb - 1) / cpb), b(PT_STEP_THREADS);
    if (G == 16)
        hipLaunchKernelGGL(pt_step_spec_kernel<16>, g, b, 0, h->stream, h->d, begin, eval);
    else if (G == 32)
        hipLaunchKernelGGL(pt_step_spec_kernel<32>, g, b, 0, h->stream, h->d, begin, eval);
    else
        hipLaunchKernelGGL(pt_step_spec_kernel<64>, g, b, 0, h->stream, h->d, begin, eval);
}

// the batched likelihood over the handle's model rows (nchains x nspec; rows without a proposal are inactive)
int launch_likelihood(rtmi_pt *h)
{
    if ((h->d_partial_x != nullptr) != batch_is_fused(h->nsrc))
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt: the kernel path was changed (rtmi_set_option \"path\") while a sampler handle exists");
    int gx = 0;
    const size_t n = (size_t)h->nsrc;
    int rc = launch_batch(h->d.V, h->d.D, h->d.NL, h->ld_model, h->nrows, h->max_nl, h->d_set, h->d_set + n, h->nsrc,
                          nullptr, h->d_set + 2 * n, h->d_partial, true, h->cus, &gx, h->stream,
                          h->incremental ? h->d.inc : nullptr, h->d_r2, h->d_inc_counts, h->d_partial_x, h->d.part_stride,
                          h->spec ? 1 : 0, h->d_pinfo);
    if (rc)
        return rc;
    if (gx != h->gx_launch)
        return fail(RTMI_ERR_HIP, "rtmi_pt: launch geometry changed (%d workgroups per model, handle sized for %d); do not "
                                  "change the scheduling knobs while a handle exists", gx, h->gx_launch);
    return RTMI_OK;
}

// One sweep of the speculative step as per-slot launches: a pass = the step (decide the window of the pass before,
// propose the next one) + the batched likelihood over every chain's rows.  A chain retires at least one slot per pass,
// so 2 x nslots_k passes complete every chain; a chain that is done early (most are: a pass retires two to three slots)
// proposes nothing, its rows are inactive and their workgroups leave at once.  The step after the last pass decides the
// last window and makes the sigma move of the chains that still have it to make.
int enqueue_sweep_spec(rtmi_pt *h, int nslots_k)
{
    const int npass = 2 * nslots_k;
    for (int p = 0; p < npass; ++p) {
        launch_step_spec(h, p == 0 ? 1 : 0, 0);
        const int rc = launch_likelihood(h);
        if (rc)
            return rc;
    }
    launch_step_spec(h, 0, 0);
    HIP_TRY(hipGetLastError());
    return RTMI_OK;
}

// One sweep with `nslots_k` node slots enqueued on the handle's stream (directly, or into a capture).
int enqueue_sweep(rtmi_pt *h, int nslots_k)
{
    if (h->spec)
        return enqueue_sweep_spec(h, nslots_k);
    launch_step(h, PT_NONE, 0, PT_BD, 0, 0);
    int rc = launch_likelihood(h);
    if (rc)
        return rc;
    int prev = PT_BD;
    for (int ivo = 0; ivo < nslots_k; ++ivo)
        for (int ipar = 0; ipar < 2; ++ipar) {
            if (ivo == 0 && ipar == 0)
                continue;  // the surface node's depth is not a parameter (:727)
            launch_step(h, prev, 0, PT_MH, ivo, ipar);
            rc = launch_likelihood(h);
            if (rc)
                return rc;
            prev = PT_MH;
        }
    launch_step(h, prev, 1, PT_NONE, 0, 0);
    HIP_TRY(hipGetLastError());
    return RTMI_OK;
}

// S sweeps in ONE launch of the proportional engine, the temperature exchange and the sample record inside the kernel
// (rtmi_kernels.hip::prop_exchange): every chain of the ladder is local and sits on a level of its own.  Returns -1 when the
// launch is refused (the caller goes on with single sweeps).
int sweep_multi(rtmi_pt *h, int S, int swap_every, int record_every, int s0)
{
    const int kmax = h->d.pr.kmax;
    rtmi::PtSweepArgs &w = h->wprop;
    w.nsweeps = S;
    w.swap_every = swap_every;
    w.record_every = record_every;
    w.s0 = s0;
    w.sweeps_done = h->sweeps_done;
    w.round = h->round;
    w.seed = h->d.seed;
    w.nct = h->nct;
    w.level = h->d_level;
    w.beta_of_level = h->d_beta_of_level;
    w.swstats = h->d_swstats;
    w.mb_logl = h->d_mb_logl;
    w.mb_chain = h->d_mb_chain;
    w.mb_seq = h->d_mb_seq;
    w.s_cap = h->d_s_count ? h->s_cap : 0;
    w.s_count = h->d_s_count;
    w.s_meta = h->d_s_meta;
    w.s_val = h->d_s_val;
    HIP_TRY(hipMemcpyAsync(h->d_wprop, &w, sizeof w, hipMemcpyHostToDevice, h->stream));
    const int nslots = S * (kmax + 1);  // (the kernel's bound of the passes: 2 x this + 2)
    const size_t n = (size_t)h->nsrc;
    if (launch_pt_prop(h->d.V, h->d.D, h->d.NL, h->ld_model, h->nchains, h->max_nl, h->d_set, h->d_set + n, h->nsrc, h->d_set + 2 * n,
                       h->cus, h->d_ptdev, sizeof h->d, h->d_wprop, h->nspec, h->epoch, h->sweep_parity, nslots, h->stream,
                       h->incremental ? h->d.inc : nullptr, h->d_r2, h->d_inc_counts, h->d_partial_x, h->d.part_stride,
                       h->d_pinfo, h->nrows * h->d.part_stride) != RTMI_OK) {
        w.nsweeps = 1;
        HIP_TRY(hipMemcpyAsync(h->d_wprop, &w, sizeof w, hipMemcpyHostToDevice, h->stream));
        return -1;
    }
    const unsigned int span = (unsigned int)(2 * nslots + 4);
    h->epoch += span > 4096u ? span : 4096u;
    h->sweep_parity ^= 1;
    h->launches += 1;
    const long long before = h->sweeps_done;
    h->sweeps_done += S;
    if (swap_every > 0)
        h->round += (uint64_t)(h->sweeps_done / swap_every - before / swap_every);
    h->kbound = h->kbound + S < kmax ? h->kbound + S : kmax;
    return RTMI_OK;
}

// engine: 0 per-slot launches, 1 the persistent launch (falls back if refused).  ev0 / ev1 (may be NULL) are recorded on the
// stream right around the sweep's launches.
int sweep_once(rtmi_pt *h, int engine, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, int *slots_out = nullptr)
{
    // perturbation slots cover every node a chain can have during this sweep: k grows by at most one per sweep
    const int kmax = h->d.pr.kmax;
    int ks = h->kbound + 1;
    if (ks > kmax)
        ks = kmax;
    h->kbound = ks;
    if (slots_out)
        *slots_out = 2 * ks;
    struct Closer {  // records ev1 however the function is left
        hipEvent_t e;
        hipStream_t s;
        ~Closer()
        {
            if (e)
                (void)hipEventRecord(e, s);
        }
    };
    if (engine == 1 && h->use_persistent && h->spec && h->nchains <= h->cus && h->d_ptdev && h->d_partial_x) {
        // the speculative sweep, workgroups dealt to the chains in proportion to their slots: one cooperative launch
        if (h->wprop.nsweeps > 1) {  // (a multi-sweep launch went before: this one is a single sweep, the exchange is the host's)
            h->wprop.nsweeps = 1;
            HIP_TRY(hipMemcpyAsync(h->d_wprop, &h->wprop, sizeof h->wprop, hipMemcpyHostToDevice, h->stream));
        }
        if (ev0)
            HIP_TRY(hipEventRecord(ev0, h->stream));
        Closer closer{ev1, h->stream};
        const size_t n = (size_t)h->nsrc;
        if (launch_pt_prop(h->d.V, h->d.D, h->d.NL, h->ld_model, h->nchains, h->max_nl, h->d_set, h->d_set + n, h->nsrc, h->d_set + 2 * n,
                           h->cus, h->d_ptdev, sizeof h->d, h->d_wprop, h->nspec, h->epoch, h->sweep_parity, ks, h->stream,
                           h->incremental ? h->d.inc : nullptr, h->d_r2, h->d_inc_counts, h->d_partial_x, h->d.part_stride,
                           h->d_pinfo, h->nrows * h->d.part_stride) == RTMI_OK) {
            h->epoch += 4096u;  // (the passes of a sweep count from a tag of its own)
            h->sweep_parity ^= 1;
            h->launches += 1;
            return RTMI_OK;
        }
        h->use_persistent = false;
        h->engine = 0;
        closer.e = nullptr;
    } else if (engine == 1 && h->use_persistent && !h->spec && h->d_sync && h->d_ptdev && h->d_partial_x) {
        if (ev0)
            HIP_TRY(hipEventRecord(ev0, h->stream));
        Closer closer{ev1, h->stream};
        rtmi::PtSweepArgs w;
        memset(&w, 0, sizeof w);
        w.nslots_k = ks;
        w.gx = h->d.gx;
        w.nchains = h->nchains;
        w.epoch = h->epoch;
        w.flag = h->d_sync;
        w.count = h->d_sync + h->nchains;
        w.abort_ = h->d_sync + 2 * (size_t)h->nchains;
        const size_t n = (size_t)h->nsrc;
        w.nspec = 0;
        if (launch_pt_sweep(h->d.V, h->d.D, h->d.NL, h->ld_model, h->nchains, h->max_nl, h->d_set, h->d_set + n, h->nsrc,
                            h->d_set + 2 * n, h->d_partial, h->cus, h->d_ptdev, sizeof h->d, &w, h->stream,
                            h->incremental ? h->d.inc : nullptr, h->d_r2, h->d_inc_counts, h->d_partial_x,
                            h->d.part_stride) == RTMI_OK) {
            h->epoch += 2u * (unsigned int)ks;
            h->launches += 1;
            return RTMI_OK;
        }
        h->use_persistent = false;
        h->engine = 0;
        closer.e = nullptr;
    }
    h->launches += 2 * ks;  // likelihood launches (spec: passes; most of the later ones find every row inactive)
    if (!h->use_graph) {
        if (ev0)
            HIP_TRY(hipEventRecord(ev0, h->stream));
        Closer closer{ev1, h->stream};
        return enqueue_sweep(h, ks);
    }
    auto it = h->graphs.find(ks);
    if (it == h->graphs.end()) {
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue_sweep(h, ks);
        const hipError_t e = hipStreamEndCapture(h->stream, &graph);
        if (rc)
            return rc;
        if (e != hipSuccess)
            return fail(RTMI_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        hipGraphExec_t exec = nullptr;
        HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        HIP_TRY(hipGraphDestroy(graph));
        it = h->graphs.emplace(ks, exec).first;
    }
    if (ev0)
        HIP_TRY(hipEventRecord(ev0, h->stream));
    Closer closer{ev1, h->stream};
    HIP_TRY(hipGraphLaunch(it->second, h->stream));
    return RTMI_OK;
}

int finish(rtmi_pt *h)
{
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int st = rtmi_device_status())
        return device_fault(st);
    return RTMI_OK;
}

// after a synchronise: the exact maximum of k, so that the next sweeps launch no more slots than needed
int refresh_kbound(rtmi_pt *h)
{
    std::vector<int> k((size_t)h->nchains);
    HIP_TRY(hipMemcpy(k.data(), h->d.k, k.size() * sizeof(int), hipMemcpyDeviceToHost));
    int m = 1;
    for (int v : k)
        m = v > m ? v : m;
    h->kbound = m;
    return RTMI_OK;
}

int enqueue_swap(rtmi_pt *h, bool local_logl = false)
{
    if (!h->d_level)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt: no ladder (call rtmi_pt_set_ladder first)");
    const size_t lds = (size_t)h->nct * (sizeof(double) + 2 * sizeof(int)) + (size_t)h->nlevels * sizeof(double) + 16;
    if (lds > 64 * 1024)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt: a ladder of %d chains does not fit the exchange kernel's LDS", h->nct);
    hipLaunchKernelGGL(pt_swap_kernel, dim3(1), dim3(256), lds, h->stream, h->nct, h->first, h->nchains,
                       (const double *)h->d_logl_all, local_logl ? (const double *)h->d.logL : (const double *)nullptr,
                       h->d_level, (const double *)h->d_beta_of_level, h->nlevels, h->d.seed, h->round, h->d_swstats,
                       h->d.beta);
    ++h->round;
    HIP_TRY(hipGetLastError());
    return RTMI_OK;
}

}  // namespace

extern "C" {

rtmi_pt *rtmi_pt_create(int nchains, int ld_model, int64_t ndata, const rtmi_pt_prior *prior, uint64_t seed,
                        uint64_t stream_id)
{
    if (nchains < 1 || nchains > 65535 || ld_model < 2 || ld_model - 1 > RTMI_MAX_LAYERS || !prior || ndata < 1 ||
        prior->kmax + 1 > ld_model || prior->kmin < 1 || prior->kmin > prior->kmax || prior->kmax > RTMI_PT_MAX_K) {
        fail(RTMI_ERR_BAD_ARG, "rtmi_pt_create: bad arguments (1 <= kmin <= kmax <= %d, ld_model >= kmax + 1)",
             RTMI_PT_MAX_K);
        return nullptr;
    }
    Context &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (!c.ready || c.set_n < 1 || !c.set_has_obs) {
        fail(RTMI_ERR_BAD_ARG, "rtmi_pt_create: call rtmi_set_sources with observations first");
        return nullptr;
    }
    if (ndata != c.set_n) {
        fail(RTMI_ERR_BAD_ARG, "rtmi_pt_create: ndata = %lld but the resident source set holds %lld observations",
             (long long)ndata, (long long)c.set_n);
        return nullptr;
    }
    if (hipSetDevice(c.device) != hipSuccess)
        return nullptr;
    {
        // pt_pack_current_kernel keeps K doubles x 2 + K ints per thread in LDS: above 64 KiB that needs the opt-in
        const size_t need = (size_t)PT_BLOCK * (size_t)prior->kmax * (2 * sizeof(double) + sizeof(int));
        if (need > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(&pt_pack_current_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)need) != hipSuccess) {
            fail(RTMI_ERR_HIP, "rtmi_pt_create: cannot reserve %zu bytes of LDS for the sweep kernels", need);
            return nullptr;
        }
    }
    rtmi_pt *h = new rtmi_pt;
    h->device = c.device;
    h->cus = c.cus;
    h->nchains = nchains;
    h->K = prior->kmax;
    h->ld_model = ld_model;
    h->max_nl = prior->kmax > 1 ? prior->kmax - 1 : 1;
    h->nsrc = c.set_n;
    if (const char *e = getenv("RTMI_PT_GRAPH"))
        h->use_graph = strcmp(e, "0") != 0;
    if (const char *e = getenv("RTMI_PT_INCREMENTAL"))
        h->incremental = strcmp(e, "0") != 0;
    if (const char *e = getenv("RTMI_PT_PERSISTENT")) {
        h->use_persistent = strcmp(e, "0") != 0;
        h->engine = h->use_persistent ? 1 : 0;
    }
    // Small source sets (the reference's own example has 20 sources): the batch path would take the plain kernel, whose sums are
    // doubles -- no speculative step, no persistent sweep, a launch pair per slot: 0.45 ms per sweep of 32 chains at 20 sources,
    // 0.81 ms at 1500.  The handle takes the fused kernel's sampler form instead where its chains fit the persistent engine
    // (0.35 ms per sweep at 20 ... 1500 sources, profiles/r04ar/pt_small_sets.txt); rtmi_set_option("path", ...) still decides
    // when it is set, and RTMI_PT_SMALL_FUSED=0 keeps the plain kernel.
    h->prefer_fused = !batch_is_fused(c.set_n) && nchains <= c.cus && h->use_persistent &&
                      !(getenv("RTMI_PT_SMALL_FUSED") && !strcmp(getenv("RTMI_PT_SMALL_FUSED"), "0"));
    PathScope path_scope(h->prefer_fused);
    const bool fused_path = batch_is_fused(c.set_n);  // which kernel the handle's likelihood launches will take
    // Speculative proposals (rtmi_pt_dev.hpp::pt_step_spec) need the exact sums of the fused kernel's sampler form: with them
    // a chain is bit for bit the chain of the one-at-a-time step.  RTMI_PT_SPECULATE = rows per chain (0: the one-at-a-time
    // step pt_step_group).
    // Default: speculate where a pass is bound by the LATENCY of a likelihood evaluation -- a chain's workgroups hold a
    // couple of records per lane or fewer (BASELINE config 3: 1e4 sources over 8 workgroups; one proposal at a time 0.92 ms per
    // sweep, six rows along the predicted path in multi-sweep launches 0.59: profiles/r04ab/pt_scan.txt) -- and not where it is
    // bound by throughput (config 4's share, 1e5 sources per chain: every speculated model that is thrown away is work the
    // machine had no room for, and the window's step costs more than the one-at-a-time step: 3.10 -> 3.7 ... 3.9 ms).  Rows
    // per chain: K <= 16: six (four groups of the step's wave work side by side, a window is two rounds of them or more; on two
    // workgroups of 512 lanes per CU, rtmi_launch_pt_prop); K <= 32: two.
    const int gx0 = batch_grid_x(nchains, h->max_nl, c.set_n, c.cus);
    int want_spec = (c.set_n / (gx0 > 0 ? gx0 : 1) <= RTMI_PT_SPECULATE_MAX_SHARE) ? (prior->kmax <= 16 ? 6 : 2) : 0;
    if (const char *e = getenv("RTMI_PT_SPECULATE"))
        want_spec = atoi(e);
    if (want_spec > PT_SPEC_MAX)
        want_spec = PT_SPEC_MAX;
    h->spec = fused_path && want_spec >= 1;
    h->nspec = h->spec ? want_spec : 1;
    h->nrows = nchains * h->nspec;
    const size_t nc = (size_t)nchains, K = (size_t)h->K, ldm = (size_t)ld_model, n = (size_t)c.set_n, nr = (size_t)h->nrows;
    const int gx = batch_grid_x(nchains, h->max_nl, c.set_n, c.cus);     // workgroups per CHAIN (persistent engine)
    h->gx_launch = batch_grid_x(h->nrows, h->max_nl, c.set_n, c.cus);    // workgroups per ROW of the batched launch
    const int pstride = gx > h->gx_launch ? gx : h->gx_launch;  // (with the speculative step a row's sums are ONE accumulator)
    const size_t nf64 = 4 * nc * K + 2 * nr * K + 4 * nc + nr + 2 * nr * ldm;
    const size_t ni32 = nc + 3 * nr;
    const size_t nsi32 = 11 * nc + 4 * nr, nsu64 = nc + nr;
    const int ncdf = 4096;
    bool ok = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void **)&h->d_set, 3 * n * sizeof(double)) == hipSuccess &&
              hipMalloc((void **)&h->d_f64, nf64 * sizeof(double)) == hipSuccess &&
              hipMalloc((void **)&h->d_i32, ni32 * sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&h->d_u64, 9 * nc * sizeof(unsigned long long)) == hipSuccess &&
              hipMalloc((void **)&h->d_partial, nr * (size_t)pstride * sizeof(double)) == hipSuccess &&
              hipMalloc((void **)&h->d_logpk, ((size_t)prior->kmax + 2) * sizeof(double)) == hipSuccess &&
              hipMalloc((void **)&h->d_sync, (2 * nc + 1) * sizeof(unsigned int)) == hipSuccess &&
              hipMalloc((void **)&h->d_inc, nr * sizeof(rtmi::PtInc)) == hipSuccess &&
              hipMalloc((void **)&h->d_cache_valid, nc * sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&h->d_spec_i32, nsi32 * sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&h->d_spec_u64, nsu64 * sizeof(unsigned long long)) == hipSuccess &&
              hipMalloc((void **)&h->d_pend_z, 3 * nc * sizeof(double)) == hipSuccess &&
              hipMalloc((void **)&h->d_pinfo, 2 * nr * sizeof(rtmi::PtPend)) == hipSuccess &&
              hipMalloc((void **)&h->d_cdf, (size_t)ncdf * sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&h->d_walloc, 2 * nc * sizeof(int)) == hipSuccess &&
              (!fused_path || hipMalloc((void **)&h->d_partial_x, 4 * nr * (size_t)pstride * sizeof(unsigned long long)) == hipSuccess);
    // The residual cache of the incremental likelihood (the chains' current residuals + one row per model row, twice with the
    // speculative step: the two sets alternate) and its counters exist only where they are read: the incremental likelihood
    // switched on AND the fused kernel's sampler form (exact sums).  A cache that does not fit (many chains over 1e6+ sources)
    // is not an error: the handle evaluates every proposal in full instead.
    const size_t r2rows = h->spec ? nc + 2 * nr : 2 * nc;
    if (ok && h->incremental && fused_path) {
        if (hipMalloc((void **)&h->d_r2, r2rows * n * sizeof(double)) != hipSuccess ||
            hipMalloc((void **)&h->d_inc_counts, 2 * nr * (size_t)pstride * sizeof(unsigned long long)) != hipSuccess) {
            (void)hipGetLastError();
            if (h->d_r2)
                (void)hipFree(h->d_r2);
            h->d_r2 = nullptr;
            h->d_inc_counts = nullptr;
            h->incremental = false;
            fprintf(stderr, "rtmi_pt_create: no room for the residual cache (%.1f GiB): every proposal is evaluated over all sources\n",
                    (double)(r2rows * n * sizeof(double)) / (1024.0 * 1024.0 * 1024.0));
        }
    } else {
        h->incremental = false;
    }
    if (ok) {
        std::vector<int> w0(2 * nc, 3);  // (k = 1)
        ok = hipMemcpy(h->d_walloc, w0.data(), w0.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok && h->spec && getenv("RTMI_PT_QSTATS"))  // diagnostic: where the workgroups' time goes (printed by rtmi_pt_run)
        ok = hipMalloc((void **)&h->d_qstats, (16 + 8 * nc) * sizeof(unsigned long long)) == hipSuccess &&   // [8 .. 16): phase clocks of the step; then per chain: ticks, sweeps, longest, passes, leader's shares / waits / steps
             hipMemset(h->d_qstats, 0, (16 + 8 * nc) * sizeof(unsigned long long)) == hipSuccess;
    if (ok) {
        rtmi::PtSweepArgs w;
        memset(&w, 0, sizeof w);
        w.nchains = h->nchains;
        w.flag = h->d_sync;
        w.count = h->d_sync + h->nchains;
        w.abort_ = h->d_sync + 2 * (size_t)h->nchains;
        w.nspec = h->nspec;
        w.walloc = h->d_walloc;
        w.stats = h->d_qstats;
        w.adapt = (getenv("RTMI_PT_ADAPT") && !strcmp(getenv("RTMI_PT_ADAPT"), "0")) ? 0 : 1;
        w.poison_lds = getenv("RTMI_PT_POISON_LDS") ? 1 : 0;  // (diagnostic: pt_sweep_prop_kernel fills its LDS with NaN patterns first)
        w.nsweeps = 1;
        h->wprop = w;
        ok = hipMalloc((void **)&h->d_wprop, sizeof w) == hipSuccess && hipMemcpy(h->d_wprop, &w, sizeof w, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok)
        ok = hipMemsetAsync(h->d_sync, 0, (2 * nc + 1) * sizeof(unsigned int), h->stream) == hipSuccess &&
             hipMemsetAsync(h->d_inc, 0, nr * sizeof(rtmi::PtInc), h->stream) == hipSuccess &&
             hipMemsetAsync(h->d_cache_valid, 0, nc * sizeof(int), h->stream) == hipSuccess &&
             hipMemsetAsync(h->d_spec_i32, 0, nsi32 * sizeof(int), h->stream) == hipSuccess &&
             hipMemsetAsync(h->d_spec_u64, 0, nsu64 * sizeof(unsigned long long), h->stream) == hipSuccess &&
             hipMemsetAsync(h->d_pend_z, 0, 3 * nc * sizeof(double), h->stream) == hipSuccess &&
             hipMemsetAsync(h->d_pinfo, 0, 2 * nr * sizeof(rtmi::PtPend), h->stream) == hipSuccess &&
             (!h->d_inc_counts || hipMemsetAsync(h->d_inc_counts, 0, 2 * nr * (size_t)pstride * sizeof(unsigned long long), h->stream) == hipSuccess) &&
             (!h->d_partial_x || hipMemsetAsync(h->d_partial_x, 0, 4 * nr * (size_t)pstride * sizeof(unsigned long long), h->stream) == hipSuccess) &&
             (!h->d_r2 || hipMemsetAsync(h->d_r2, 0, r2rows * n * sizeof(double), h->stream) == hipSuccess) &&
             hipMemsetAsync(h->d_f64, 0, nf64 * sizeof(double), h->stream) == hipSuccess &&
             hipMemsetAsync(h->d_i32, 0, ni32 * sizeof(int), h->stream) == hipSuccess &&
             hipMemsetAsync(h->d_u64, 0, 9 * nc * sizeof(unsigned long long), h->stream) == hipSuccess &&
             hipMemsetAsync(h->d_partial, 0, nr * (size_t)pstride * sizeof(double), h->stream) == hipSuccess;
    PtDev &d = h->d;
    memset(&d, 0, sizeof d);
    if (ok) {
        // The handle OWNS its source set (a later rtmi_set_sources does not retarget it) -- and, where its likelihood is the
        // fused kernel's sampler form, keeps it SORTED BY DEPTH (NaN last): the sampler only ever asks for sums over the
        // sources, which do not depend on their order (exact sums), and in depth order the sources a proposal does not affect
        // are a prefix of the set, a workgroup's records are in layer order as they are loaded (no sort in the kernel).
        std::vector<double> set(3 * n);
        ok = hipMemcpy(set.data(), c.d_set, 3 * n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
        std::vector<int> cdf((size_t)ncdf, 0);
        double zmax = -INFINITY, zmin = INFINITY;
        for (size_t i = 0; i < n; ++i) {
            const double z = set[n + i];
            if (z == z && z > zmax)
                zmax = z;
            if (z == z && z < zmin)
                zmin = z;
        }
        d.zmax_src = zmax;  // the deepest source: proposals that change the model only below it need no launch
        d.cdf_z0 = zmin;
        d.cdf_inv = (zmax > zmin && zmax - zmin < INFINITY) ? (double)ncdf / (zmax - zmin) : 0.0;
        if (ok && fused_path) {
            std::vector<size_t> perm(n);
            for (size_t i = 0; i < n; ++i)
                perm[i] = i;
            const double *zs = set.data() + n;
            std::stable_sort(perm.begin(), perm.end(), [zs](size_t a, size_t b) {
                const double za = zs[a], zb = zs[b];
                if (za != za || zb != zb)
                    return zb != zb && za == za;  // NaN after every number
                return za < zb;
            });
            std::vector<double> sorted(3 * n);
            for (size_t i = 0; i < n; ++i)
                for (int a = 0; a < 3; ++a)
                    sorted[(size_t)a * n + i] = set[(size_t)a * n + perm[i]];
            set.swap(sorted);
            // depth histogram: cdf[g] = sources whose bin is below g (rtmi_pt_dev.hpp::prefix_above uses the same formula)
            if (d.cdf_inv > 0.0) {
                std::vector<int> cnt((size_t)ncdf + 1, 0);
                for (size_t i = 0; i < n; ++i) {
                    const double z = set[n + i];
                    if (z != z)
                        continue;
                    const double t = (z - d.cdf_z0) * d.cdf_inv;
                    const int g = t >= (double)(ncdf - 1) ? ncdf - 1 : (t > 0.0 ? (int)t : 0);
                    cnt[(size_t)g + 1] += 1;
                }
                int run = 0;
                for (int g = 0; g < ncdf; ++g) {
                    run += cnt[(size_t)g];
                    cdf[(size_t)g] = run;
                }
            }
        }
        ok = ok && hipMemcpyAsync(h->d_set, set.data(), 3 * n * sizeof(double), hipMemcpyHostToDevice, h->stream) == hipSuccess &&
             hipMemcpyAsync(h->d_cdf, cdf.data(), (size_t)ncdf * sizeof(int), hipMemcpyHostToDevice, h->stream) == hipSuccess &&
             hipStreamSynchronize(h->stream) == hipSuccess;
    }
    std::vector<double> lpk((size_t)prior->kmax + 2, 0.0);
    for (int k = 1; k <= prior->kmax + 1; ++k)
        lpk[(size_t)k] = R::log_pk(*prior, k);
    lpk[0] = lpk[1];
    if (ok)
        ok = hipMemcpyAsync(h->d_logpk, lpk.data(), lpk.size() * sizeof(double), hipMemcpyHostToDevice, h->stream) ==
             hipSuccess;
    d.nchains = nchains;
    d.K = h->K;
    d.ld_model = ld_model;
    d.gx = gx;
    d.ndata = ndata;
    d.pr = *prior;
    d.seed = seed;
    d.stream_id = (uint32_t)stream_id;
    d.logpk = h->d_logpk;
    d.nsrc = c.set_n;
    d.cdf = (fused_path && d.cdf_inv > 0.0) ? h->d_cdf : nullptr;
    d.ncdf = ncdf;
    d.nspec = h->nspec;
    d.spec_step = h->spec ? 1 : 0;
    d.dbg = (h->d_qstats && getenv("RTMI_PT_QSTATS") && atoi(getenv("RTMI_PT_QSTATS")) >= 3) ? h->d_qstats + 8 : nullptr;
    d.spec_prop = (getenv("RTMI_PT_SPEC_PROP") ? atoi(getenv("RTMI_PT_SPEC_PROP")) : 0) ? 1 : 0;
    d.part_stride = pstride;
    d.gx_launch = h->gx_launch;
    if (ok) {
        double *f = h->d_f64;
        d.z = f, f += nc * K;
        d.v = f, f += nc * K;
        d.pz = f, f += nr * K;
        d.pv = f, f += nr * K;
        d.sd = f, f += nc;
        d.logL = f, f += nc;
        d.ss = f, f += nc;
        d.beta = f, f += nc;
        d.plp = f, f += nr;
        d.V = f, f += nr * ldm;
        d.D = f, f += nr * ldm;
        d.wz = f, f += nc * K;
        d.wv = f, f += nc * K;
        int *i = h->d_i32;
        d.k = i, i += nc;
        d.pk = i, i += nr;
        d.pactive = i, i += nr;
        d.NL = (int32_t *)i;
        d.ctr = h->d_u64;
        d.stats = (long long *)(h->d_u64 + nc);
        d.partial = h->d_partial;
        d.partial_x = h->d_partial_x;
        d.inc = h->d_inc;
        d.cache_valid = h->d_cache_valid;
        d.incremental = (h->incremental && h->d_partial_x) ? 1 : 0;  // (exact sums: the fused kernel path)
        int *q = h->d_spec_i32;
        d.cur = q, q += nc;
        d.fin = q, q += nc;
        d.nwin = q, q += nc;
        d.wend = q, q += nc;
        d.winv = q, q += nc;
        d.nact = q, q += nc;
        d.wbuf = q, q += nc;
        d.pend_i = q, q += 4 * nc;
        d.pslot = q, q += nr;
        d.pinv = q, q += nr;
        d.rowwg = q, q += nr;
        d.alist = q, q += nr;
        d.wctr = h->d_spec_u64;
        d.pctr = h->d_spec_u64 + nc;
        d.pinfo = h->d_pinfo;
        d.walloc = h->d_walloc;
        d.acc = h->d_partial_x;
        // one-at-a-time step: a chain's rows of the residual slab never change (current residuals: row c; its proposal's: row
        // nchains + c); the speculative step writes them with every proposal
        h->inc0.assign(nc, rtmi::PtInc{});
        for (size_t q2 = 0; q2 < nc; ++q2) {
            h->inc0[q2].mover = 1;
            h->inc0[q2].cur_row = (int)q2;
            h->inc0[q2].pend_row = h->inc0[q2].prop_row = (int)(nc + q2);
        }
        if (!h->spec)
            ok = hipMemcpyAsync(h->d_inc, h->inc0.data(), nc * sizeof(rtmi::PtInc), hipMemcpyHostToDevice, h->stream) == hipSuccess;
        // every chain starts as a half-space at the middle of the velocity prior, sigma 0.01, beta 1 (as the host
        // engine of round 1 did); rtmi_pt_set_chain overrides
        std::vector<double> init(nc * K, 0.0), sd0(nc, 0.01), b0(nc, 1.0);
        std::vector<int> k0(nc, 1);
        for (size_t q2 = 0; q2 < nc; ++q2)
            init[q2 * K] = 0.5 * (prior->vmin + prior->vmax);
        ok = ok && hipMemcpyAsync(d.v, init.data(), nc * K * sizeof(double), hipMemcpyHostToDevice, h->stream) == hipSuccess &&
             hipMemcpyAsync(d.sd, sd0.data(), nc * sizeof(double), hipMemcpyHostToDevice, h->stream) == hipSuccess &&
             hipMemcpyAsync(d.beta, b0.data(), nc * sizeof(double), hipMemcpyHostToDevice, h->stream) == hipSuccess &&
             hipMemcpyAsync(d.k, k0.data(), nc * sizeof(int), hipMemcpyHostToDevice, h->stream) == hipSuccess &&
             hipMalloc((void **)&h->d_ptdev, sizeof(PtDev)) == hipSuccess &&
             hipMemcpyAsync(h->d_ptdev, &d, sizeof(PtDev), hipMemcpyHostToDevice, h->stream) == hipSuccess &&
             hipStreamSynchronize(h->stream) == hipSuccess;
    }
    if (!ok) {
        fail(RTMI_ERR_HIP, "rtmi_pt_create: device allocation or initialisation failed (%s)",
             hipGetErrorString(hipGetLastError()));
        rtmi_pt_destroy(h);
        return nullptr;
    }
    return h;
}

void rtmi_pt_destroy(rtmi_pt *h)
{
    if (!h)
        return;
    (void)hipSetDevice(h->device);
    if (h->stream)
        (void)hipStreamSynchronize(h->stream);
    for (auto &g : h->graphs)
        (void)hipGraphExecDestroy(g.second);
    void *bufs[] = {h->d_set, h->d_f64, h->d_i32, h->d_u64, h->d_partial, h->d_logpk, h->d_level, h->d_order,
                    h->d_logl_all, h->d_beta_of_level, h->d_swstats, h->d_s_count, h->d_s_meta, h->d_s_val, h->d_sync, h->d_ptdev, h->d_inc, h->d_cache_valid, h->d_r2, h->d_inc_counts, h->d_partial_x,
                    h->d_cdf, h->d_spec_i32, h->d_spec_u64, h->d_pend_z, h->d_pinfo, h->d_qstats, h->d_walloc, h->d_wprop, h->d_mb_logl, h->d_mb_chain, h->d_mb_seq};
    for (void *b : bufs)
        if (b)
            (void)hipFree(b);
    if (h->stream)
        (void)hipStreamDestroy(h->stream);
    delete h;
}

int rtmi_pt_set_chain(rtmi_pt *h, int chain, int k, const double *depths, const double *vels, double sd, double beta)
{
    if (!h || chain < 0 || chain >= h->nchains || k < 1 || k > h->K || !depths || !vels)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_set_chain: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    const size_t c = (size_t)chain, K = (size_t)h->K;
    HIP_TRY(hipMemcpyAsync(h->d.z + c * K, depths, (size_t)k * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d.v + c * K, vels, (size_t)k * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d.sd + c, &sd, sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d.beta + c, &beta, sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d.k + c, &k, sizeof(int), hipMemcpyHostToDevice, h->stream));
    // the residual cache belongs to the model that was here: invalid until a model of this chain has been evaluated in full
    HIP_TRY(hipMemsetAsync(h->d_cache_valid + c, 0, sizeof(int), h->stream));
    if (h->spec) {
        // no window awaits a decision, no accepted proposal's residuals are pending
        HIP_TRY(hipMemsetAsync(h->d_inc + c * (size_t)h->nspec, 0, (size_t)h->nspec * sizeof(rtmi::PtInc), h->stream));
        HIP_TRY(hipMemsetAsync(h->d.nwin + c, 0xFF, sizeof(int), h->stream));  // -1
        HIP_TRY(hipMemsetAsync(h->d.nact + c, 0, sizeof(int), h->stream));
        HIP_TRY(hipMemsetAsync(h->d.cur + c, 0, sizeof(int), h->stream));
        HIP_TRY(hipMemsetAsync(h->d.fin + c, 0, sizeof(int), h->stream));
        HIP_TRY(hipMemsetAsync(h->d.pend_i + 4 * c, 0, 4 * sizeof(int), h->stream));
        const int wk = 2 * k + 1;
        HIP_TRY(hipMemcpyAsync(h->d_walloc + c, &wk, sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->d_walloc + h->nchains + c, &wk, sizeof(int), hipMemcpyHostToDevice, h->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(h->d_inc + c, &h->inc0[c], sizeof(rtmi::PtInc), hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (k > h->kbound)
        h->kbound = k;
    return RTMI_OK;
}

int rtmi_pt_set_beta(rtmi_pt *h, const double *beta)
{
    if (!h || !beta)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_set_beta: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->d.beta, beta, (size_t)h->nchains * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RTMI_OK;
}

int rtmi_pt_get_all(rtmi_pt *h, int *k, double *depths, double *vels, double *sd, double *logL, double *ss, double *beta)
{
    if (!h)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_get_all: NULL handle");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    const size_t nc = (size_t)h->nchains, K = (size_t)h->K;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (k)
        HIP_TRY(hipMemcpy(k, h->d.k, nc * sizeof(int), hipMemcpyDeviceToHost));
    if (depths)
        HIP_TRY(hipMemcpy(depths, h->d.z, nc * K * sizeof(double), hipMemcpyDeviceToHost));
    if (vels)
        HIP_TRY(hipMemcpy(vels, h->d.v, nc * K * sizeof(double), hipMemcpyDeviceToHost));
    if (sd)
        HIP_TRY(hipMemcpy(sd, h->d.sd, nc * sizeof(double), hipMemcpyDeviceToHost));
    if (logL)
        HIP_TRY(hipMemcpy(logL, h->d.logL, nc * sizeof(double), hipMemcpyDeviceToHost));
    if (ss)
        HIP_TRY(hipMemcpy(ss, h->d.ss, nc * sizeof(double), hipMemcpyDeviceToHost));
    if (beta)
        HIP_TRY(hipMemcpy(beta, h->d.beta, nc * sizeof(double), hipMemcpyDeviceToHost));
    if (int st = rtmi_device_status())
        return device_fault(st);
    return RTMI_OK;
}

int rtmi_pt_get_chain(const rtmi_pt *hc, int chain, int *k, double *depths, double *vels, double *sd, double *logL,
                      double *ss)
{
    rtmi_pt *h = const_cast<rtmi_pt *>(hc);
    if (!h || chain < 0 || chain >= h->nchains)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_get_chain: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t c = (size_t)chain, K = (size_t)h->K;
    int kk = 0;
    HIP_TRY(hipMemcpy(&kk, h->d.k + c, sizeof(int), hipMemcpyDeviceToHost));
    if (k)
        *k = kk;
    if (depths)
        HIP_TRY(hipMemcpy(depths, h->d.z + c * K, (size_t)kk * sizeof(double), hipMemcpyDeviceToHost));
    if (vels)
        HIP_TRY(hipMemcpy(vels, h->d.v + c * K, (size_t)kk * sizeof(double), hipMemcpyDeviceToHost));
    if (sd)
        HIP_TRY(hipMemcpy(sd, h->d.sd + c, sizeof(double), hipMemcpyDeviceToHost));
    if (logL)
        HIP_TRY(hipMemcpy(logL, h->d.logL + c, sizeof(double), hipMemcpyDeviceToHost));
    if (ss)
        HIP_TRY(hipMemcpy(ss, h->d.ss + c, sizeof(double), hipMemcpyDeviceToHost));
    return RTMI_OK;
}

int rtmi_pt_evaluate(rtmi_pt *h)
{
    if (!h)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_evaluate: NULL handle");
    std::lock_guard<std::mutex> lk(h->mu);
    PathScope path_scope(h->prefer_fused);
    HIP_TRY(hipSetDevice(h->device));
    const dim3 g(blocks_for(h->nchains)), b(64);
    hipLaunchKernelGGL(pt_pack_current_kernel, g, b, pack_lds(h), h->stream, h->d);
    int rc = launch_likelihood(h);
    if (rc)
        return rc;
    if (h->spec)
        launch_step_spec(h, 0, 1);
    else
        launch_step(h, PT_EVAL, 0, PT_NONE, 0, 0);
    HIP_TRY(hipGetLastError());
    h->launches += 1;
    h->models += h->nchains;
    rc = finish(h);
    return rc ? rc : refresh_kbound(h);
}

int rtmi_pt_sweep(rtmi_pt *h, int nsweeps)
{
    return rtmi_pt_run(h, nsweeps, 0, nullptr, 0);
}

int rtmi_pt_set_ladder(rtmi_pt *h, int nchains_total, int first_chain, const int *level, const double *beta_of_level,
                       int nlevels)
{
    if (!h || !level || !beta_of_level || nlevels < 1 || nchains_total < h->nchains || first_chain < 0 ||
        first_chain + h->nchains > nchains_total)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_set_ladder: bad argument");
    for (int i = 0; i < nchains_total; ++i)
        if (level[i] < 0 || level[i] >= nlevels)
            return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_set_ladder: level[%d] = %d outside [0, %d)", i, level[i], nlevels);
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    void *old[] = {h->d_level, h->d_order, h->d_logl_all, h->d_beta_of_level, h->d_swstats, h->d_mb_logl, h->d_mb_chain, h->d_mb_seq};
    for (void *b : old)
        if (b)
            HIP_TRY(hipFree(b));
    h->d_mb_logl = nullptr;
    h->d_mb_chain = nullptr;
    h->d_mb_seq = nullptr;
    h->d_level = h->d_order = nullptr;
    h->d_logl_all = h->d_beta_of_level = nullptr;
    h->d_swstats = nullptr;
    const size_t nct = (size_t)nchains_total;
    HIP_TRY(hipMalloc((void **)&h->d_level, nct * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&h->d_order, nct * sizeof(int)));
    HIP_TRY(hipMalloc((void **)&h->d_logl_all, nct * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&h->d_beta_of_level, (size_t)nlevels * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&h->d_swstats, 2 * sizeof(long long)));
    HIP_TRY(hipMemcpy(h->d_level, level, nct * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_beta_of_level, beta_of_level, (size_t)nlevels * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(h->d_swstats, 0, 2 * sizeof(long long)));
    h->nct = nchains_total;
    h->first = first_chain;
    h->nlevels = nlevels;
    h->round = 0;
    {
        // one chain per level, levels 0 .. nct - 1: the exchange can then run inside a multi-sweep launch (a chain's level is its
        // position in the exchange's order, and stays a permutation under exchanges)
        std::vector<char> seen(nct, 0);
        bool perm = nlevels == nchains_total;
        for (size_t i = 0; perm && i < nct; ++i) {
            perm = !seen[(size_t)level[i]];
            seen[(size_t)level[i]] = 1;
        }
        h->levels_perm = perm;
        HIP_TRY(hipMalloc((void **)&h->d_mb_logl, 4 * nct * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&h->d_mb_chain, 4 * nct * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&h->d_mb_seq, 4 * nct * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(h->d_mb_logl, 0, 4 * nct * sizeof(double)));
        HIP_TRY(hipMemset(h->d_mb_chain, 0, 4 * nct * sizeof(int)));
        HIP_TRY(hipMemset(h->d_mb_seq, 0, 4 * nct * sizeof(unsigned long long)));
    }
    std::vector<double> b((size_t)h->nchains);
    for (int c = 0; c < h->nchains; ++c)
        b[(size_t)c] = beta_of_level[level[first_chain + c]];
    HIP_TRY(hipMemcpy(h->d.beta, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice));
    return RTMI_OK;
}

// New inverse temperatures for the levels of the ladder in place: who sits on which level, the exchange's counters and its
// position in the random stream stay as they are (the burn-in adaptation of prjmh_temper_rf.f90:363-383 re-spaces the ladder
// while the run goes on).
int rtmi_pt_set_ladder_betas(rtmi_pt *h, const double *beta_of_level, int nlevels)
{
    if (!h || !beta_of_level)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_set_ladder_betas: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->d_level || nlevels != h->nlevels)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_set_ladder_betas: %s", h->d_level ? "nlevels differs from the ladder's" : "no ladder (call rtmi_pt_set_ladder first)");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<int> lev((size_t)h->nct);
    HIP_TRY(hipMemcpy(lev.data(), h->d_level, lev.size() * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<double> b((size_t)h->nchains);
    for (int c = 0; c < h->nchains; ++c)
        b[(size_t)c] = beta_of_level[lev[(size_t)(h->first + c)]];
    HIP_TRY(hipMemcpy(h->d_beta_of_level, beta_of_level, (size_t)nlevels * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d.beta, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice));
    return RTMI_OK;
}

int rtmi_pt_get_ladder(rtmi_pt *h, int *level, int64_t *swap_stats2)
{
    if (!h || !h->d_level)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_get_ladder: no ladder");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (level)
        HIP_TRY(hipMemcpy(level, h->d_level, (size_t)h->nct * sizeof(int), hipMemcpyDeviceToHost));
    if (swap_stats2) {
        long long s[2];
        HIP_TRY(hipMemcpy(s, h->d_swstats, sizeof s, hipMemcpyDeviceToHost));
        swap_stats2[0] = s[0];
        swap_stats2[1] = s[1];
    }
    return RTMI_OK;
}

// The exchange with the gather done by the caller (MPI_Allgather, gloo, ...): logl_all[nchains_total] on the host.
int rtmi_pt_swap_round_host(rtmi_pt *h, const double *logl_all)
{
    if (!h || !logl_all)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_swap_round_host: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->d_level)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt: no ladder (call rtmi_pt_set_ladder first)");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->d_logl_all, logl_all, (size_t)h->nct * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));  // logl_all is the caller's (pageable) buffer
    return enqueue_swap(h);
}

static int swap_round_locked(rtmi_pt *h, rtmi_comm *comm)
{
    const int nranks = comm ? rtmi_comm_size(comm) : 1;
    if (!h->d_level)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt: no ladder (call rtmi_pt_set_ladder first)");
    if (h->nct != h->nchains * nranks || (comm && h->first != rtmi_comm_rank(comm) * h->nchains))
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_swap_round: the ladder has %d chains, the communicator %d ranks x %d "
                                      "local chains (rank r owns chains r*n .. r*n+n-1)", h->nct, nranks, h->nchains);
    if (nranks > 1) {
        // the ONLY data-path collective of the sampler: nchains doubles per rank, once per swap round
        const int rc = rtmi_comm_allgather_f64(comm, h->d.logL, h->d_logl_all, h->nchains, h->stream);
        if (rc)
            return rc;
    }
    return enqueue_swap(h, nranks == 1);  // every chain local: logL is read in place
}

int rtmi_pt_swap_round(rtmi_pt *h, rtmi_comm *comm)
{
    if (!h)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_swap_round: NULL handle");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    return swap_round_locked(h, comm);
}

int rtmi_pt_sample_buffer(rtmi_pt *h, int capacity)
{
    if (!h || capacity < 0)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_sample_buffer: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    void *old[] = {h->d_s_count, h->d_s_meta, h->d_s_val};
    for (void *b : old)
        if (b)
            HIP_TRY(hipFree(b));
    h->d_s_count = nullptr;
    h->d_s_meta = nullptr;
    h->d_s_val = nullptr;
    h->s_cap = 0;
    if (capacity == 0)
        return RTMI_OK;
    HIP_TRY(hipMalloc((void **)&h->d_s_count, sizeof(int)));
    HIP_TRY(hipMalloc((void **)&h->d_s_meta, (size_t)capacity * 3 * sizeof(long long)));
    HIP_TRY(hipMalloc((void **)&h->d_s_val, (size_t)capacity * (2 + 2 * (size_t)h->K) * sizeof(double)));
    HIP_TRY(hipMemset(h->d_s_count, 0, sizeof(int)));
    HIP_TRY(hipMemset(h->d_s_val, 0, (size_t)capacity * (2 + 2 * (size_t)h->K) * sizeof(double)));
    h->s_cap = capacity;
    return RTMI_OK;
}

int rtmi_pt_samples(rtmi_pt *h, int max_samples, int64_t *sweep, int *chain, int *k, double *sd, double *logL,
                    double *depths, double *vels, int *dropped)
{
    if (!h || max_samples < 0)
        return -RTMI_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->d_s_count)
        return 0;
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
        return -RTMI_ERR_HIP;
    int cnt = 0;
    if (hipMemcpy(&cnt, h->d_s_count, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
        return -RTMI_ERR_HIP;
    if (dropped)
        *dropped = cnt > h->s_cap ? cnt - h->s_cap : 0;
    int n = cnt < h->s_cap ? cnt : h->s_cap;
    if (n > max_samples)
        n = max_samples;
    const size_t K = (size_t)h->K, w = 2 + 2 * K;
    std::vector<long long> meta((size_t)n * 3);
    std::vector<double> val((size_t)n * w);
    if (n > 0) {
        if (hipMemcpy(meta.data(), h->d_s_meta, meta.size() * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(val.data(), h->d_s_val, val.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            return -RTMI_ERR_HIP;
    }
    // in the order (sweep, chain): the ring is filled in the order the chains get there (within a multi-sweep launch the chains
    // are not in step), the caller sees one order whatever the engine
    std::vector<int> ord((size_t)n);
    for (int i = 0; i < n; ++i)
        ord[(size_t)i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&meta](int a, int b) {
        const long long sa = meta[(size_t)a * 3], sb = meta[(size_t)b * 3];
        return sa != sb ? sa < sb : meta[(size_t)a * 3 + 1] < meta[(size_t)b * 3 + 1];
    });
    for (int o = 0; o < n; ++o) {
        const int i = ord[(size_t)o];
        if (sweep)
            sweep[o] = meta[(size_t)i * 3];
        if (chain)
            chain[o] = (int)meta[(size_t)i * 3 + 1];
        if (k)
            k[o] = (int)meta[(size_t)i * 3 + 2];
        if (sd)
            sd[o] = val[(size_t)i * w];
        if (logL)
            logL[o] = val[(size_t)i * w + 1];
        if (depths)
            memcpy(depths + (size_t)o * K, &val[(size_t)i * w + 2], K * sizeof(double));
        if (vels)
            memcpy(vels + (size_t)o * K, &val[(size_t)i * w + 2 + K], K * sizeof(double));
    }
    if (hipMemset(h->d_s_count, 0, sizeof(int)) != hipSuccess)
        return -RTMI_ERR_HIP;
    return n;
}

int rtmi_pt_run(rtmi_pt *h, int nsweeps, int swap_every, rtmi_comm *comm, int record_every)
{
    if (!h || nsweeps < 0 || swap_every < 0 || record_every < 0)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_run: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    PathScope path_scope(h->prefer_fused);
    HIP_TRY(hipSetDevice(h->device));
    if (record_every > 0 && !h->d_s_count)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_run: record_every > 0 needs rtmi_pt_sample_buffer first");
    // Undecided engine: a window of ten sweeps in the first long enough call -- persistent: one warm, then a block of four
    // timed as a whole; per-slot launches: one warm (captures its graph), then a block of four -- ONE event
    // synchronisation, and the persistent launch stays unless its block took more than 1.08x the time per proposal slot.
    // Blocks, not single sweeps: an event right around a cooperative launch also times the hand-over between the
    // stream's queue and the cooperative queue, which back-to-back launches do not pay (single-sweep timing chose the
    // per-slot launches for config 4's share, where the persistent launch is 10 % faster).  The case to catch is a GPU
    // shared with other processes, where the persistent kernel is several times slower.  The chains do not depend on the
    // engine (bit-identical), so the trial is ordinary work.  The window opens once the sweeps have their full length (see
    // below) and ten more sweeps are to go; a call too short for that leaves the question to the next call.
    const bool trial = h->engine < 0 && h->use_persistent && nsweeps >= 12;
    int t0 = -1;  // first sweep of the window
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int block_slots[2] = {0, 0};
    if (trial)
        for (auto &e : ev)
            HIP_TRY(hipEventCreate(&e));
    struct EvGuard {
        hipEvent_t *e;
        ~EvGuard()
        {
            for (int i = 0; i < 4; ++i)
                if (e[i])
                    (void)hipEventDestroy(e[i]);
        }
    } ev_guard{ev};
    // Several sweeps per launch, the exchange inside the kernel: where every chain of the ladder is local and has a level of its
    // own (RTMI_PT_MULTI=0 keeps one launch per sweep; a ladder with several chains on a level, chains on other ranks, or no
    // ladder: one launch per sweep, the exchange between the launches)
    const bool multi_env = !(getenv("RTMI_PT_MULTI") && !strcmp(getenv("RTMI_PT_MULTI"), "0"));
    bool multi_ok = multi_env && h->use_persistent && h->spec && h->nchains <= h->cus && h->d_ptdev && h->d_partial_x &&
                    (!comm || rtmi_comm_size(comm) == 1) && h->d_level && h->levels_perm && h->nct == h->nchains && h->first == 0 &&
                    h->nct >= 2 && h->d_mb_seq;
    for (int s = 0; s < nsweeps; ++s) {
        int rc;
        if (multi_ok && h->engine == 1 && nsweeps - s >= 2) {  // (a single sweep: one launch, the exchange by the host's kernel)
            int S = nsweeps - s;
            const int smax = 2000 / (h->d.pr.kmax + 2) > 1 ? 2000 / (h->d.pr.kmax + 2) : 1;
            S = S > smax ? smax : S;
            const int rcm = sweep_multi(h, S, swap_every, record_every, s);
            if (rcm == RTMI_OK) {
                s += S - 1;
                continue;
            }
            if (rcm > 0)
                return rcm;
            multi_ok = false;  // refused: single sweeps from here on
        }
        if (trial && t0 < 0 && h->engine < 0 && nsweeps - s >= 10) {
            // not before the sweeps have the length they keep: the bound of k that sizes a sweep grows by one per sweep
            // up to kmax (it is only refreshed at a synchronisation), and a sweep of kmax node slots, many of them
            // without a proposal for most chains, is what a long run consists of
            const int kmax = h->d.pr.kmax, ks_next = h->kbound + 1 < kmax ? h->kbound + 1 : kmax;
            if (ks_next >= kmax)
                t0 = s;
        }
        if (t0 >= 0 && s - t0 < 10 && h->engine < 0) {
            const int u = s - t0, blk = u / 5, v = u % 5;  // v = 0: warm; 1..4: the timed block
            int slots = 0;
            rc = sweep_once(h, blk == 0 ? 1 : 0, v == 1 ? ev[2 * blk] : nullptr, v == 4 ? ev[2 * blk + 1] : nullptr, &slots);
            if (v >= 1)
                block_slots[blk] += slots;
        } else {
            rc = sweep_once(h, h->engine == 1 ? 1 : 0);
        }
        if (rc)
            return rc;
        if (t0 >= 0 && s - t0 == 9 && h->engine < 0) {
            float ms_p = 0.0f, ms_g = 0.0f;
            HIP_TRY(hipEventSynchronize(ev[3]));
            HIP_TRY(hipEventElapsedTime(&ms_p, ev[0], ev[1]));
            HIP_TRY(hipEventElapsedTime(&ms_g, ev[2], ev[3]));
            const float per_p = ms_p / (float)(block_slots[0] > 0 ? block_slots[0] : 1);
            const float per_g = ms_g / (float)(block_slots[1] > 0 ? block_slots[1] : 1);
            h->engine = per_p <= 1.08f * per_g ? 1 : 0;
            if (getenv("RTMI_PT_VERBOSE"))
                fprintf(stderr, "rtmi: sweep engines timed: persistent %.1f us per slot, per-slot launches %.1f us per slot -> %s\n",
                        1e3 * per_p, 1e3 * per_g, h->engine ? "persistent" : "per-slot launches");
        }
        ++h->sweeps_done;
        if (swap_every > 0 && h->sweeps_done % swap_every == 0) {
            rc = swap_round_locked(h, comm);
            if (rc)
                return rc;
        }
        if (record_every > 0 && (s + 1) % record_every == 0) {  // counted within this call
            hipLaunchKernelGGL(pt_record_kernel, dim3(blocks_for(h->nchains)), dim3(64), 0, h->stream, h->d,
                               (const int *)h->d_level, h->first, h->sweeps_done, h->s_cap, h->d_s_count, h->d_s_meta,
                               h->d_s_val);
            HIP_TRY(hipGetLastError());
        }
    }
    const int rc = finish(h);  // the one synchronisation of the call
    if (!rc && h->d_qstats) {
        std::vector<unsigned long long> stv(16 + 8 * (size_t)h->nchains);
        unsigned long long *st = stv.data();
        HIP_TRY(hipMemcpy(st, h->d_qstats, stv.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(h->d_qstats, 0, stv.size() * sizeof(unsigned long long)));
        if (h->d.dbg) {
            const unsigned long long *q = st + 8;
            const double n = q[6] ? (double)q[6] : 1.0;
            fprintf(stderr, "rtmi:   the step, per call (%llu calls, %.2f slots per window): loads %.1f us, decisions %.1f, sigma + state %.1f, the window's rounds %.1f, "
                            "prefix + words + stores out %.1f; whole call %.1f us\n", q[6], (double)q[7] / n, 0.01 * q[0] / n, 0.01 * q[1] / n, 0.01 * q[2] / n,
                    0.01 * q[3] / n, 0.01 * q[4] / n, 0.01 * q[5] / n);
        }
        if (st[5] > 0 && getenv("RTMI_PT_QSTATS") && atoi(getenv("RTMI_PT_QSTATS")) >= 2) {
            // how long the chains' sweeps take (their leading workgroups, kernel entry to the end of the chain's sweep)
            std::vector<double> beta(h->nchains);
            std::vector<int> kk(h->nchains);
            HIP_TRY(hipMemcpy(beta.data(), h->d.beta, beta.size() * sizeof(double), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(kk.data(), h->d.k, kk.size() * sizeof(int), hipMemcpyDeviceToHost));
            double mean_all = 0.0, mean_max = 0.0;
            for (int c = 0; c < h->nchains; ++c) {
                const unsigned long long *q = st + 16 + 8 * (size_t)c;
                const double m = q[1] ? 0.01 * (double)q[0] / (double)q[1] : 0.0, np = q[3] ? (double)q[3] : 1.0;
                mean_all += m / h->nchains;
                mean_max = m > mean_max ? m : mean_max;
                fprintf(stderr, "rtmi:   chain %2d beta %.4f k %2d: sweep %.0f us on average, longest %.0f us, %.2f passes per sweep; per pass its leading "
                                "workgroup: own share %.0f us, then %.0f us waiting for the chain's others; steps (+ exchanges) %.0f us per sweep, of which %.0f us waiting for the partner's mailbox\n", c, beta[c], kk[c], m,
                        0.01 * (double)q[2], q[1] ? (double)q[3] / (double)q[1] : 0.0, 0.01 * (double)q[4] / np, 0.01 * (double)q[5] / np,
                        q[1] ? 0.01 * (double)q[6] / (double)q[1] : 0.0, q[1] ? 0.01 * (double)q[7] / (double)q[1] : 0.0);
            }
            fprintf(stderr, "rtmi:   mean over the chains %.0f us, slowest chain on average %.0f us\n", mean_all, mean_max);
        }
        if (st[5] > 0)
            fprintf(stderr, "rtmi: speculative sweeps (%d): %llu likelihood shares (%.1f us each), %llu steps (%.1f us each); of the "
                            "workgroups' time %.1f %% in the likelihood, %.1f %% in steps, %.1f %% waiting\n", nsweeps, st[3],
                    st[3] ? 0.01 * (double)st[0] / (double)st[3] : 0.0, st[4], st[4] ? 0.01 * (double)st[1] / (double)st[4] : 0.0,
                    100.0 * (double)st[0] / (double)st[5], 100.0 * (double)st[1] / (double)st[5], 100.0 * (double)st[2] / (double)st[5]);
    }
    return rc ? rc : refresh_kbound(h);
}

int rtmi_pt_stats(const rtmi_pt *hc, int64_t *out8)
{
    rtmi_pt *h = const_cast<rtmi_pt *>(hc);
    if (!h || !out8)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_stats: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<long long> st((size_t)h->nchains * 8);
    HIP_TRY(hipMemcpy(st.data(), h->d.stats, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (int j = 0; j < 7; ++j) {
        long long s = 0;
        for (int c = 0; c < h->nchains; ++c)
            s += st[(size_t)c * 8 + j];
        out8[j < 6 ? j : 7] = s;
    }
    out8[6] = h->launches;
    out8[7] += h->models;  // + the models of rtmi_pt_evaluate
    return RTMI_OK;
}

int rtmi_pt_chain_stats(const rtmi_pt *hc, int64_t *out)
{
    rtmi_pt *h = const_cast<rtmi_pt *>(hc);
    if (!h || !out)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_chain_stats: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    static_assert(sizeof(long long) == sizeof(int64_t), "stats are 64-bit counters");
    HIP_TRY(hipMemcpy(out, h->d.stats, (size_t)h->nchains * 8 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return RTMI_OK;
}

// Sources the likelihood launches of this handle have traced / have taken from the residual cache so far (sources of
// proposals only; the incremental likelihood off: cached stays 0 and traced is not counted).
int rtmi_pt_likelihood_counts(const rtmi_pt *hc, int64_t *traced, int64_t *cached)
{
    rtmi_pt *h = const_cast<rtmi_pt *>(hc);
    if (!h || !traced || !cached)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_likelihood_counts: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *traced = *cached = 0;
    if (!h->d_inc_counts)  // the incremental likelihood is off for this handle
        return RTMI_OK;
    std::vector<unsigned long long> c((size_t)2 * h->nrows * h->d.part_stride);  // one pair per workgroup
    HIP_TRY(hipMemcpy(c.data(), h->d_inc_counts, c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long tr = 0, ca = 0;
    for (size_t i = 0; i + 1 < c.size(); i += 2) {
        tr += c[i];
        ca += c[i + 1];
    }
    *traced = (int64_t)tr;
    *cached = (int64_t)ca;
    return RTMI_OK;
}

int rtmi_pt_stream_positions(rtmi_pt *h, uint64_t *ctr)
{
    if (!h || !ctr)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_stream_positions: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(ctr, h->d.ctr, (size_t)h->nchains * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RTMI_OK;
}

int rtmi_pt_speculation(rtmi_pt *h, int64_t *out4)
{
    if (!h || !out4)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_pt_speculation: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<long long> st((size_t)h->nchains * 8);
    HIP_TRY(hipMemcpy(st.data(), h->d.stats, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
    long long discarded = 0;
    for (int c = 0; c < h->nchains; ++c)
        discarded += st[(size_t)c * 8 + 7];
    out4[0] = h->spec ? h->nspec : 1;
    out4[1] = discarded;
    out4[2] = h->engine;
    out4[3] = h->d_partial_x ? 1 : 0;
    return RTMI_OK;
}

// The pairing and acceptance rule of the exchange as a pure host function (no GPU): what pt_swap_kernel runs on the
// device.  level[] is updated in place; returns accepted swaps (>= 0) or -RTMI_ERR_BAD_ARG.
int rtmi_pt_swap_decide(int nchains_total, const double *logl_all, int *level, const double *beta_of_level, int nlevels,
                        uint64_t seed, uint64_t round, int64_t *proposed)
{
    if (nchains_total < 1 || !logl_all || !level || !beta_of_level || nlevels < 1)
        return -RTMI_ERR_BAD_ARG;
    for (int i = 0; i < nchains_total; ++i)
        if (level[i] < 0 || level[i] >= nlevels)
            return -RTMI_ERR_BAD_ARG;
    std::vector<int> order((size_t)nchains_total);
    long long prop = 0;
    const int acc = R::swap_decide(nchains_total, logl_all, level, beta_of_level, seed, round, order.data(), &prop);
    if (proposed)
        *proposed += prop;
    return acc;
}

}  // extern "C"
