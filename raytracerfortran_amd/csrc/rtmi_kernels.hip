// rtmi_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the layered travel-time forward model.
//
// What is computed (reference: /root/reference/subroutineR-quiet.f90, module raymod):
//   per source k:  nl = whichLayer(depths, z_k)                       (:9-32)
//                  stack (H_i, V_i), i <= nl, above the source         (:38-69 InsertLayer)
//                  ray parameter p by the guarded Newton / bisection
//                  hybrid on f(p) = R - sum H_i V_i p / sqrt(1-p^2 V_i^2)   (:77-153, :226-405)
//                  t = sum H_i / (V_i sqrt(1 - p^2 V_i^2)), or -999    (:156-169)
// The reference stops its root-find at |f| < 0.1 m, so t depends on the exact iteration
// sequence: the kernels reproduce the reference's arithmetic operation for operation in IEEE
// fp64 (rtmi_device.hpp).  THIS FILE MUST BE COMPILED WITH -ffp-contract=off (build.py does).
//
// Two kernels (DESIGN.md has the measurements):
//
//  (1) trace_kernel -- one lane per source in input order.  The three solver branches of GetPTime
//      are folded into one per-lane state machine around a single wave-converged "evaluate f and
//      f' at x" loop (one sqrt + two divides per layer; f and f' share the sqrt, solvebst's
//      duplicate costFunc call is evaluated once).  Used for small NSrc and for the many-models
//      batch (grid.y = model).  Its cost is set by the slowest lane of each wave: deepest source x
//      most evaluations, i.e. ~20 % of the executed lane-instructions are useful on random input.
//
//  (2) fused_kernel -- the large-NSrc path (and batches of models), same arithmetic per source, scheduled
//      so that lanes stay busy.  One persistent workgroup per CU; workgroup b owns the 64-source chunks
//      b, b + gridDim.x, ... and keeps their records in LDS, round by round:
//        a.  load: coalesced reads of offset/depth, whichLayer, start value p0, and a STABLE counting
//            sort of the records in LDS by DESCENDING layer index (ballots, no atomics: slots are the
//            same on every run), so that a wave's layer loop has one trip count at any time; straight
//            top-layer rays are finished here;
//        a2. first evaluation of every record at full occupancy: GetPTime's branch (:136-153); sources
//            converged at p0 are done; every other one gets a class by the evaluations it is going to need
//            (Newton-only sources last; bisection sources by a closed-form estimate, trip_class in
//            rtmi_device.hpp, which only orders the work), and the per-chunk class counts are taken here;
//        a3. the solve order (class by class, most evaluations first, each by descending layer index) is
//            built from those counts and one 64-lane scan, deterministically;
//        b.  solve: every wave pulls records from the workgroup's queue; a lane that finishes takes the
//            next record once `refill_min` lanes are idle, so the evaluation loop runs with most lanes
//            live instead of waiting for the slowest source; no global memory traffic inside the loop
//            (records and results live in LDS).  Longest first keeps the end of the round short.  Waves
//            0..3 -- one per SIMD, the oldest waves, which the issue arbiter prefers -- are "express"
//            waves: the other waves hand them, through an LDS ring, every source that has already taken
//            `promote_age` trips and, in the drain, their last `retire_max` live lanes (resumable lane
//            states); `tail_group` steers the last records of the queue to the first wave groups.
//            Wave-uniform values are kept scalar on purpose (readfirstlane / readlane / DPP);
//        c.  time: t = sum H_i/(V_i cos_i) from the final ray parameters in sorted order, written to
//            timeP (the only HBM write); with observations the squared residuals are reduced per
//            workgroup in a fixed order (deterministic logL).
//      HBM traffic stays at the algorithmic 24 B per source (offset/depth are read a second time from
//      L2 in the sort's second pass); no workspace, no global atomics.
//  Both kernels are templates on the scalar type: double is the reference's precision, float the
//  fp32 path of BASELINE config 5 (its layer term has a guarded form of its own, everything else is IEEE).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "rtmi_device.hpp"
#include "rtmi_kernels.h"
#include "rtmi_pt_dev.hpp"

namespace rtmi {

// =============================================================================================
// (1) single-launch path
// =============================================================================================

// One source per lane, all lanes of the wave together (wave-uniform loops inside);
// `valid` = false lanes carry a dummy source and their result is discarded.
template <typename T>
__device__ __forceinline__ T trace_one(const LayerTable<T> &t, int nlayers, T R, T z, bool valid, T &p_out)
{
    const int nl = which_layer(t, nlayers, z);
    const bool straight = (nl == 1);  // source in the top layer (:94-97)
    const bool active = valid && !straight;
    const int il = active ? nl - 1 : 1;  // 0-based layer of the source; dummy lanes use layer 1
    Lane<T> s;
    lane_setup(s, t, il, R, z);
    const T p0 = active ? initial_p0(t, il, R, z) : T(0.0);
    lane_start(s, p0);
    if (!active || p0 != p0)
        s.st = ST_DONE;
    const int nfull_wave = wave_max_i32(active ? s.nfull : 0);
    const bool use_pred = model_terms_nonneg(t);  // the plain kernel's trips are as long as their slowest lane: every evaluation counts

    while (__any(s.st != ST_DONE)) {
        T f, fp;
        eval_f_fp(t, s.xe, s.nfull, nfull_wave, s.hvl, s.vvl, s.R, s.numer_ok, s.st != ST_DONE, f, fp);
        const bool was_init = s.st == ST_INIT;
        lane_advance(t, s, f, fp, use_pred);
        if (was_init && s.skip0 && s.st == ST_BIS)  // solvebst entered with f(x1) > 0: idle trips need not be made
            lane_skip_idle_trips(t, s, il, z - t.D[il - 1]);
    }

    T tt = travel_time(t, s.x, s.nfull, nfull_wave, il, z, active);
    if (!s.conv)
        tt = -T(999.0);
    p_out = s.x;
    if (straight) {
        tt = div_ieee(T(sqrt(z * z + R * R)), t.V[0]);
        p_out = __builtin_nan("");
    }
    return tt;
}

// A model row whose d_nlayers[m] cannot be used (whole workgroup, before anything touches LDS).  Returns true when
// the workgroup must leave: 0 with allow_skip is an inactive row (nothing written but a zero partial sum); anything
// else outside [1, max_nlayers] would overrun the LDS layer table, so the row is reported through the fault word
// (bit 8) and its outputs are filled with NaN -- loud, never silently wrong.
template <typename T>
__device__ __forceinline__ bool reject_model_row(int nlayers, int max_nlayers, int allow_skip, int *d_status,
                                                 T *timeP_row, int64_t nsrc, T *partial_slot, int bx, int gx)
{
    if (nlayers >= 1 && nlayers <= max_nlayers)
        return false;
    const bool skip = allow_skip && nlayers == 0;
    if (!skip) {
        if (threadIdx.x == 0 && bx == 0 && d_status)
            atomicOr(d_status, 8);
        if (timeP_row)
            for (int64_t k = (int64_t)bx * blockDim.x + threadIdx.x; k < nsrc; k += (int64_t)gx * blockDim.x)
                timeP_row[k] = T(__builtin_nan(""));
    }
    if (partial_slot && threadIdx.x == 0)
        *partial_slot = skip ? T(0.0) : T(__builtin_nan(""));
    return true;
}

// grid.x: source tiles (grid-stride), grid.y: models.
template <typename T, bool WANT_P, bool WANT_T, bool WANT_SS>
__global__ __launch_bounds__(RTMI_BLOCK) void trace_kernel(TraceArgsT<T> a)
{
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];  // storage for either element type
    T *lds = reinterpret_cast<T *>(lds_raw);
    const int m = blockIdx.y;
    const int nlayers = a.d_nlayers ? a.d_nlayers[m] : a.nlayers;
    if (a.d_nlayers && reject_model_row<T>(nlayers, a.max_nlayers, a.allow_skip, a.d_status,
                                           WANT_T ? a.d_timeP + (int64_t)m * a.nsrc : nullptr, a.nsrc,
                                           WANT_SS ? a.d_partial + ((int64_t)m * gridDim.x + blockIdx.x) : nullptr,
                                           (int)blockIdx.x, (int)gridDim.x))
        return;
    const LayerTable<T> t = carve_table(lds, nlayers, false, a.skiptab != 0);
    stage_model(t, a.d_vels + (int64_t)m * a.ld_model, a.d_depths + (int64_t)m * a.ld_model, nlayers);

    T ss = T(0.0);  // this thread's sum of squared residuals (fused logL)
    T *out = a.d_timeP + (int64_t)m * a.nsrc;
    const int64_t ntiles = (a.nsrc + RTMI_BLOCK - 1) / RTMI_BLOCK;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t k = tile * RTMI_BLOCK + threadIdx.x;
        const bool valid = k < a.nsrc;
        const T R = valid ? a.d_src_offset[k] : T(1.0);
        const T z = valid ? a.d_src_depth[k] : T(1.0);
        T p;
        const T tt = trace_one(t, nlayers, R, z, valid, p);
        if (valid) {
            if (WANT_T)
                out[k] = tt;
            if (WANT_P)
                a.d_p_final[k] = p;
            if (WANT_SS) {
                const T r = a.d_obs[k] - tt;
                ss = ss + r * r;
            }
        }
    }

    if (WANT_SS) {
        // block reduction in a fixed order: lanes by xor-butterfly, then waves 0..3
        __shared__ T wsum[RTMI_BLOCK / 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            ss = ss + __shfl_xor(ss, off, 64);
        if ((threadIdx.x & 63) == 0)
            wsum[threadIdx.x >> 6] = ss;
        __syncthreads();
        if (threadIdx.x == 0) {
            T s = wsum[0];
            for (int w = 1; w < RTMI_BLOCK / 64; ++w)
                s = s + wsum[w];
            a.d_partial[(int64_t)m * gridDim.x + blockIdx.x] = s;
        }
    }
}

// (1b) SMALL launches (the reference's own example: NSrc = 20): a GROUP of G lanes per source, one layer term per lane.
// With a lane per source a launch of a few sources is one wave walking the slowest source's dependent trips alone: every trip
// evaluates the stack's terms one after the other (eight layers x 35 dependent double-precision instructions: about 1 us), and
// the kernel took 27 us for 20 sources -- the whole of what the reference needs on one CPU core (27.7 us).  Here lane j of a
// source's group computes term j of costFunc / costFunc_Prime (:184-221) -- the same guarded forms, the plain operators for the
// whole wave when any operand is outside the guards -- leaves it in LDS, and every lane of the group adds the terms up IN LAYER
// ORDER (the stack's full layers, the source's own partial layer, then the zeros of the lanes beyond the stack: x + 0 = x), so
// f, f' and with them every iterate are the reference's bit for bit; the state machine (lane_advance) runs redundantly in the
// lanes of a group.  The travel time (:156, :165-166) is summed the same way.  G = 8, 16, 32, 64 >= nlayers + 1.
template <int G, bool WANT_P>
__global__ __launch_bounds__(RTMI_BLOCK) void trace_small_kernel(TraceArgsT<double> a)
{
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    const int nlayers = a.nlayers;
    const LayerTable<double> t = carve_table(lds_raw, nlayers, false, a.skiptab != 0);
    // behind the table: one {tf, tp} per lane
    double2 *scr = reinterpret_cast<double2 *>(lds_raw + ((RTMI_TABLE_ELEMS(nlayers, false, a.skiptab) + 1) & ~(size_t)1));
    constexpr int SPB = RTMI_BLOCK / G;            // sources per workgroup and trip
    const int j = threadIdx.x & (G - 1);           // this lane's term
    const int gb = threadIdx.x & ~(G - 1);         // the group's first lane (index into scr)
    // the first trip's source is requested before the model is staged (a small launch is all latency)
    const int64_t k_first = (int64_t)blockIdx.x * SPB + threadIdx.x / G;
    double R_next = k_first < a.nsrc ? a.d_src_offset[k_first] : 1.0, z_next = k_first < a.nsrc ? a.d_src_depth[k_first] : 1.0;
    stage_model(t, a.d_vels, a.d_depths, nlayers);
    const bool use_pred = model_terms_nonneg(t);
    for (int64_t base = (int64_t)blockIdx.x * SPB; base < a.nsrc; base += (int64_t)gridDim.x * SPB) {
        const int64_t k = base + threadIdx.x / G;
        const bool valid = k < a.nsrc;
        const double R = R_next, z = z_next;
        {
            const int64_t kn = k + (int64_t)gridDim.x * SPB;  // the next trip's, in flight during this one
            R_next = kn < a.nsrc ? a.d_src_offset[kn] : 1.0;
            z_next = kn < a.nsrc ? a.d_src_depth[kn] : 1.0;
        }
        const int nl = which_layer(t, nlayers, z);
        const bool straight = (nl == 1);  // source in the top layer (:94-97)
        const bool active = valid && !straight;
        const int il = active ? nl - 1 : 1;
        Lane<double> s;
        lane_setup(s, t, il, R, z);
        const double p0 = active ? initial_p0(t, il, R, z, active) : 0.0;
        lane_start(s, p0);
        if (!active || p0 != p0)
            s.st = ST_DONE;
        // this lane's layer of the source's stack: a full layer, the source's own partial layer, or none
        const bool full = active && j < s.nfull, own = active && j == s.nfull;
        const int jt = full ? j : t.nul;
        const double hv = own ? s.hvl : t.HW[jt].hv, vv = own ? s.vvl : t.HW[jt].vv;
        const int nt_wave = wave_max_i32(active ? s.nfull + 1 : 0);
        // f, f' at s.xe: this lane's term, then the sums in layer order (every lane of the group ends with the same f, f')
        auto evaluate = [&](double &f, double &fp) {
            const double x = s.xe, xx = x * x;
            double tf, tp;
            const bool ok = layer_term<true>(xx, x, hv, vv, tf, tp) && s.numer_ok && mag_ok(x);
            if (__any(s.st != ST_DONE && !ok))  // rare: some operand near the ends of the range, 0, Inf or NaN
                layer_term<false>(xx, x, hv, vv, tf, tp);
            scr[threadIdx.x] = make_double2(tf, tp);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double sf = 0.0, sp = 0.0;
            for (int i = 0; i < nt_wave; ++i) {  // (a group never straddles a wave)
                const double2 e = scr[gb + i];
                sf = sf + e.x;
                sp = sp + e.y;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            f = s.R - sf;
            fp = -sp;
        };
        if (__any(s.st != ST_DONE)) {
            // GetPTime's branch after cf = f(p0), cfp = f'(p0) (:136-153), as the fused kernel's first-evaluation pass takes it
            double f, fp;
            evaluate(f, fp);
            const bool live = s.st == ST_INIT;
            Lane<double> n = s;
            lane_advance_init(t, n, f, fp, live);
            if (live)
                s = n;
            if (live && s.st == ST_BIS0 && s.skip0) {  // solvebst entered with f(x1) > 0 without evaluating it; idle trips need not be made
                lane_enter_bisection(s, false);
                lane_skip_idle_trips(t, s, il, z - t.D[il - 1]);
            }
        }
        while (__any(s.st != ST_DONE)) {
            double f, fp;
            evaluate(f, fp);
            lane_advance_solve(t, s, f, fp, use_pred);
        }
        // t = sum H_i / (V_i cos_i): this lane's layer, then the sum in layer order
        {
            const double pp = s.x * s.x;
            const double hj = own ? z - t.D[il - 1] : t.H[jt], vj = own ? t.V[il] : t.V[jt], vvj = own ? t.VV[il] : t.VV[jt];
            const double tj = div_ieee(hj, double(vj * sqrt(1.0 - pp * vvj)));
            scr[threadIdx.x] = make_double2(tj, 0.0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double tt = 0.0;
            for (int i = 0; i < nt_wave; ++i)
                tt = tt + scr[gb + i].x;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!s.conv)
                tt = -999.0;
            double p_out = s.x;
            if (straight) {
                tt = div_ieee(double(sqrt(z * z + R * R)), t.V[0]);
                p_out = __builtin_nan("");
            }
            if (valid && j == 0) {
                a.d_timeP[k] = tt;
                if (WANT_P)
                    a.d_p_final[k] = p_out;
            }
        }
    }
}

// logL[m] = -(N/2) ln(2 pi) - ( SS/(2 sd^2) + N ln sd )     loglhood.f90:194-196
__global__ void loglike_kernel(const double *__restrict__ partial, int nparts, const double *__restrict__ sd,
                               int64_t ndat, int nmodels, double *__restrict__ logL, double *__restrict__ ss_out)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nmodels)
        return;
    double s = 0.0;
    for (int i = 0; i < nparts; ++i)
        s = s + partial[(int64_t)m * nparts + i];
    const double PI2 = 3.141592653589793238462643383279502884197;  // data_type.f90:5
    const double n = (double)ndat;
    const double sdm = sd[m];
    logL[m] = -(n / 2.0) * log(2.0 * PI2) - (s / (2.0 * (sdm * sdm)) + n * log(sdm));
    if (ss_out)
        ss_out[m] = s;  // SUM(DresRT**2): lets a caller re-price sigma without re-tracing (loglhood.f90:195)
}

// =============================================================================================
// (2) fused large-NSrc kernel: CU-wide LDS work queue sorted by layer + lane refill
// =============================================================================================

// Sort bin of a layer index: exact for <= 63 layers, proportional above.  Monotone in nl.
__device__ __forceinline__ int bin_of(int nl, int nl_max)
{
    return nl_max <= 63 ? nl : (int)(((long long)nl * 63) / nl_max);
}

#ifndef RTMI_PRED_MIN_PER_LANE
#define RTMI_PRED_MIN_PER_LANE 2  // records per lane from which a round consults newton_lands_converged (lane_advance_solve)
#endif
#ifndef RTMI_PREFETCH
#define RTMI_PREFETCH 4      // chunks of source data a wave keeps in flight in the load / sort passes
#endif
#define RTMI_RK_LOC_MASK 0x1FFFu      // rk: local source index within the workgroup's round (< 8192)
#define RTMI_RK_NL_SHIFT 13           // rk: nl in bits 13..23 (nl <= 1025)
#define RTMI_RK_NL_MASK 0x7FFu
#define RTMI_RK_CLASS_SHIFT 24        // rk: class of the solve order in bits 24..25 (0 = most evaluations expected)
#define RTMI_RK_CLASSES 4             // RTMI_TRIP_CLASSES classes of bisection sources + the Newton-only class
#define RTMI_RK_CONV 0x80000000u      // rk: solver reported convergence
#define RTMI_RK_SHORT 0x20000000u     // rk: Newton straight from p0 (short class of the solve queue)
#define RTMI_RK_DONE 0x40000000u      // rk: p_final is in rp (set by the first evaluation, or when the solve loop finishes the source)
#define RTMI_RK_SKIP 0x04000000u      // rk: (sorted source sets) a slot without a record: finished, nothing to compute or to write

// One persistent workgroup per CU (blockDim = 64 x waves, up to 1024).  Workgroup b owns the 64-source
// chunks b, b + gridDim.x, ...; per round it holds up to `slots` of their records in LDS.
// grid.y = model (many models over one shared source set: loglhood.f90:128-146 per proposal); every
// workgroup row b of a model owns chunks b, b + gridDim.x, ... of the sources.
// ROWS: the launch's layer tables carry one row per stack height (models of up to RTMI_ROWS_MAX_NL interfaces); a
// template parameter so that each instantiation holds one form of the evaluation loop only (code size).
// The body is a device function of (model m, workgroup row bx of gx rows): the kernel below calls it with the launch's
// grid indices, the persistent sweep kernel of the tempering sampler (pt_sweep_kernel) once per proposal slot.
template <typename T, bool WANT_P, bool WANT_T, bool WANT_SS, bool DIAG, int TF, bool INCR = false, bool SWEEP = false>
__device__ __forceinline__ void fused_body(FusedArgsT<T> a, const int m, const int bx, const int gx, T *lds)
{
    // (`a` by value; by const reference -- fields re-read from the argument block where they are used -- was measured: C2 +2 %)
    constexpr bool ROWS = TF == 1;  // table form (rtmi_device.hpp): 1 = row tables, 1 and 2 = constant table stride
    // INC: the sampler's form of the likelihood -- squared residuals are accumulated EXACTLY (Acc192, rtmi_device.hpp), so
    // the sum does not depend on who adds what in which order, and (with a.d_inc) unaffected sources come from the cache
    constexpr bool INC = INCR && WANT_SS && !WANT_T && !WANT_P && std::is_same<T, double>::value;  // its own instantiation
    // SORTED: the sampler's form runs over a source set that is sorted by depth (ascending, NaN last; rtmi_pt_create sorts the
    // handle's copy once): a workgroup's records are in layer order as they are loaded -- no counting, no scan, no scatter --
    // and the sources a proposal does not affect are a PREFIX of the set that is never loaded (PtInc::n0).
    constexpr bool SORTED = INC;
    const int pstride = a.part_stride ? a.part_stride : gx;  // rows of d_partial_x / d_inc_counts are [m][pstride]
    a.nlayers = a.d_nlayers ? a.d_nlayers[m] : a.nlayers;
    if (a.d_nlayers && reject_model_row<T>(a.nlayers, a.max_nlayers, a.allow_skip, a.d_status,
                                           WANT_T ? a.d_timeP + (int64_t)m * a.nsrc : nullptr, a.nsrc,
                                           (WANT_SS && !INC) ? a.d_partial + ((int64_t)m * gx + bx) : nullptr, bx, gx)) {
        if (INC && threadIdx.x == 0) {
            const bool skip = a.allow_skip && a.nlayers == 0;
            if (a.acc_atomic) {
                if (!skip)  // an unusable row: NaN, like the plain form
                    __hip_atomic_fetch_or(a.d_partial_x + 4 * ((int64_t)m * pstride) + 3, RTMI_ACC_NAN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                unsigned long long *o = a.d_partial_x + 4 * ((int64_t)m * pstride + bx);
                o[0] = o[2] = o[3] = 0ull;
                o[1] = skip ? 0ull : RTMI_ACC_NAN;
            }
        }
        return;
    }
    a.d_vels += (int64_t)m * a.ld_model;
    a.d_depths += (int64_t)m * a.ld_model;
    if (WANT_T)
        a.d_timeP += (int64_t)m * a.nsrc;
    // incremental likelihood (PtInc in rtmi_kernels.h): this model's record and residual rows
    const bool inc = INC && a.d_inc != nullptr;
    Acc192 acc = acc_zero();  // (INC) this thread's squared residuals
    PtInc ic = {};
    unsigned int n_cached = 0, n_traced = 0;  // this WAVE's sources (statistics of the incremental likelihood; scalar)
    T *r2cur = nullptr, *r2pend = nullptr, *r2prop = nullptr;
    int64_t n0 = 0;  // (SORTED) sources [0, n0) are not loaded: their residuals come from the cache
    if (inc) {
        ic = a.d_inc[m];
        if (ic.pending && ic.pend_idx > 0) {  // the pending proposal's own record
            const PtPend pp = a.d_pinfo[ic.pend_idx - 1];
            ic.pzthr = pp.zthr;
            ic.pdl1 = pp.dl1;
            ic.pdl2 = pp.dl2;
            ic.pn0 = pp.n0;
            ic.pend_row = pp.prop_row;
        }
        r2cur = a.d_r2 + (int64_t)ic.cur_row * a.nsrc;
        r2pend = a.d_r2 + (int64_t)ic.pend_row * a.nsrc;
        r2prop = a.d_r2 + (int64_t)ic.prop_row * a.nsrc;
        n0 = ic.n0;
    }
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index: scalar
    const int nwib = blockDim.x >> 6;
    const int slots = a.slots;
    // Source data of chunk c of the round that starts at j0 (0 outside the round / the source set).  The loads of
    // a wave's next chunk are always requested before the current one is used, the very first ones before the
    // layer table is built.
    const int64_t nchunks_all = (a.nsrc - n0 + 63) / 64;
    auto depth_of = [&](int64_t j0, int c) {
        const int64_t chunk = (int64_t)bx + (j0 + c) * (int64_t)gx;
        const int64_t k = n0 + chunk * 64 + lane;
        return (c < slots / 64 && chunk < nchunks_all && k < a.nsrc) ? a.d_src_depth[k] : T(0.0);
    };
    auto offset_of = [&](int64_t j0, int c) {
        const int64_t chunk = (int64_t)bx + (j0 + c) * (int64_t)gx;
        const int64_t k = n0 + chunk * 64 + lane;
        return (c < slots / 64 && chunk < nchunks_all && k < a.nsrc) ? a.d_src_offset[k] : T(0.0);
    };
    // a window of PF chunks per wave is kept in flight (C2: a wave's four chunks of the round, all at once; the forms of the
    // kernel with residual sums are at their register limit and keep two)
    constexpr int PF = INC ? 1 : (WANT_SS ? 2 : RTMI_PREFETCH);
    T zw[PF], Rw[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        zw[j] = depth_of(0, wib + j * nwib);
        if constexpr (SORTED)
            Rw[j] = offset_of(0, wib + j * nwib);
    }
    const LayerTable<T> t = carve_table<TF>(lds, a.nlayers, ROWS, a.skiptab != 0);
    // DET: the records' slots decide the order of the floating-point residual sum (the batch API's logL), so the sort must
    // be the same on every run: ranks by ballots.  Every other form of the kernel (travel times only; the sampler's
    // order-independent sums) is indifferent to WHICH slot of its bin a record gets: ranks by one LDS atomic per lane.
    constexpr bool DET = WANT_SS && !INC;
    T ss = T(0.0);  // this thread's sum of squared residuals (fused logL)

    // LDS behind the table: first everything of FIXED size (bin counters, queue words, the hand-over ring) -- with row tables
    // the table's size is a constant, so these sit at compile-time offsets and cost the solve loop no scalar registers -- then
    // the per-record arrays, whose positions depend on `slots`.
    // (SWEEP: inlined into the slot loop of the persistent sweep kernel, whose register allocation is at the edge, the fully
    // constant form of this offset -- skiptab is 1 whenever TF != 0 -- makes the compiler spill FOUR times as many values to
    // scratch, all over the solve loop: sweep 0.92 -> 1.01 ms, bisected in profiles/r03g/pt_bisect.txt.  There the launch's
    // value is read instead; everywhere else the constant is the better form.)
    char *fx = reinterpret_cast<char *>(lds) + (RTMI_TABLE_ELEMS(a.nlayers, TF, (TF != 0 && !SWEEP) ? 1 : a.skiptab) * sizeof(T) + 15) / 16 * 16;
    uint32_t *hist = reinterpret_cast<uint32_t *>(fx);        // [16 waves][64 bins] counts, then first slots
    uint32_t *qctl = hist + 16 * 64;                          // [0] queue head, [1] records sorted, [2] long, [3] short
    // hand-over pool of the drain (see "wave retirement" below): RTMI_POOL suspended lane states
    T *pl_xe = reinterpret_cast<T *>(qctl + 4);
    T *pl_x = pl_xe + RTMI_POOL;
    T *pl_xb = pl_x + RTMI_POOL;
    T *pl_dx = pl_xb + RTMI_POOL;
    int *pl_slot = reinterpret_cast<int *>(pl_dx + RTMI_POOL);
    int *pl_state = pl_slot + RTMI_POOL;                        // st | k << 8 | conv << 16
    // the ring's flags and control words are polled: volatile, and typed as LDS so that the polls are ds_read, not flat loads
    typedef __attribute__((address_space(3))) int lds_int;
    volatile lds_int *pl_ready = (volatile lds_int *)(pl_state + RTMI_POOL);
    volatile lds_int *pl_ctl = pl_ready + RTMI_POOL;            // [0] tail (reserved), [1] head (claimed), [2] local waves still solving
    // (pl_ctl[4 .. 4 + 9]: cls_base, below.)  The record arrays follow the 16 control words; every size above is a multiple
    // of 16 bytes, so they are 16-byte aligned.  [Derived from the GENERIC pointer pl_state: an address_space(3) pointer
    // turned into an integer is a bare LDS offset, not an address.]
    T *rR = reinterpret_cast<T *>(pl_state + 2 * RTMI_POOL + 16);  // offset }
    T *rz = rR + slots;                                  // depth             } sorted by descending layer
    T *rp = rz + slots;                                  // p0, later p_final }
    uint32_t *rk = reinterpret_cast<uint32_t *>(rp + slots);  // loc | nl << 13 | conv << 31
    uint32_t *tnl = rk + slots;                               // nl per local index (between the sort passes)
    uint16_t *order = reinterpret_cast<uint16_t *>(tnl + slots); // solve order: slots of the long class, then the short
    // exact solve order: per 64-slot chunk, counts and first order[] positions of the two classes
    uint32_t *cls_base = reinterpret_cast<uint32_t *>(pl_state + 2 * RTMI_POOL + 4);  // [9] first order[] position of a class
    // per 64-slot chunk, counts and first positions of the four classes as 16-bit fields of a 64-bit word; they
    // live in `hist`, which is free once the records are in their slots
    unsigned long long *ch_cnt = reinterpret_cast<unsigned long long *>(hist);  // [128]
    unsigned long long *ch_start = ch_cnt + 128;                                  // [128]
    // (SORTED) the control words of a round: queue head, records, the hand-over ring -- before the round's load pass
    auto init_round_ctl = [&]() {
        if (wib == 0) {
            if (lane == 0) {
                qctl[0] = 0;
                qctl[1] = 0;
                qctl[2] = 0;
                qctl[3] = 0;
                pl_ctl[0] = 0;
                pl_ctl[1] = 0;
                pl_ctl[2] = nwib > 4 ? nwib - 4 : 0;
            }
            for (int i = lane; i < RTMI_POOL; i += 64)
                pl_ready[i] = 0;
        }
    };
    if constexpr (SORTED) {
        init_round_ctl();  // of the first round: before the barriers of stage_model
    } else if constexpr (!DET) {
        if (threadIdx.x < 64)
            hist[threadIdx.x] = 0u;  // the bin counters of the first round: zeroed before the barriers of stage_model
    }
    if constexpr (SORTED) {
        // The unaffected prefix [0, n0) of the depth-sorted set: squared residuals under the chain's current model, from the
        // cache -- from the pending proposal's row where THAT proposal changed them (sources from pn0 on, by their depths) --
        // and the chain's mover row brings r2cur up to date on the way.  Nothing else of these sources is ever loaded.
        if (inc && n0 > 0) {
            const int64_t stride = (int64_t)gx * blockDim.x;
#pragma unroll 1
            for (int64_t kw = (int64_t)bx * blockDim.x + wib * 64; kw < n0; kw += stride) {  // wave-uniform
                const int64_t k = kw + lane;
                const bool valid = k < n0;
                bool moved = false;
                if (valid && ic.pending && k >= ic.pn0) {
                    const double z = (double)a.d_src_depth[k];
                    moved = !(z < ic.pzthr && z != ic.pdl1 && z != ic.pdl2);
                }
                if (valid) {
                    T r2c;
                    if (moved) {
                        r2c = r2pend[k];
                        if (ic.mover)
                            r2cur[k] = r2c;
                    } else {
                        r2c = r2cur[k];
                    }
                    acc_add_double(acc, (double)r2c);
                }
                n_cached += (unsigned int)__popcll(__ballot(valid));
            }
        }
    }
    stage_model<true>(t, a.d_vels, a.d_depths, a.nlayers);

    const int64_t b = bx;
    const int64_t nblocks = gx;
    const int64_t nchunks = (a.nsrc - n0 + 63) / 64;  // (n0 = 0 but for the sampler's form: chunks of the LOADED part of the set)
    const int CH = slots / 64;
    const int nl_max = a.nlayers + 1;

    // Diagnostic phase clocks (NULL in normal runs): per workgroup {start, after sort, after solve, after time}
    // of its FIRST round, in shader-clock ticks.  The stamps go to a buffer nothing else reads.
    unsigned long long *stamps = (DIAG && a.d_stamps) ? a.d_stamps + RTMI_STAMPS_PER_WG * bx : nullptr;
    const bool fine = DIAG && stamps && a.diag_fine;  // per-trip clocks and counters (they slow the loop down by a quarter)
    for (int64_t j0 = 0; b + j0 * nblocks < nchunks; j0 += CH) {
        // ---- a. load + workgroup-local STABLE counting sort ------------------------------------
        // No atomics: lane b of each wave counts the wave's records of bin b in a register (one
        // ballot per distinct bin of a chunk), so slots -- and with them the summation order of the
        // fused logL -- are the same on every run.
        if (DIAG && stamps && j0 == 0 && threadIdx.x == 0)
            stamps[0] = clock64();
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        if constexpr (SORTED) {
            // ---- a. (sorted source set) ONE load pass, no sort --------------------------------------------------------
            // The set is sorted by depth, so this workgroup's sources -- chunks b, b + gx, ... of the loaded part, in order --
            // come with non-decreasing layer index: the record of local source `loc` goes to slot T - 1 - loc (descending
            // layer order, as the counting sort of the other forms leaves it).  Sources that need no record -- not affected
            // by the proposal (residual from the cache), straight rays of the top layer (finished here) -- are the
            // SHALLOWEST of the round, i.e. the highest slots: the round's record count is 1 + the highest slot that holds a
            // record, and a slot below it without one (a source exactly on dl1 / dl2 has neighbours that are cached) is
            // marked RTMI_RK_SKIP: finished, nothing to compute, nothing to write.
            if (j0 > 0) {
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    zw[j] = depth_of(j0, wib + j * nwib);
                    Rw[j] = offset_of(j0, wib + j * nwib);
                }
                init_round_ctl();  // (the round before ended with a barrier)
                __syncthreads();
            }
            const int64_t my_left = (nchunks - b + nblocks - 1) / nblocks - j0;  // this workgroup's chunks from this round on
            const int nmy = (int)(my_left < CH ? my_left : CH);
            const int64_t last_ord = (nchunks - 1 - b) / nblocks;
            const bool has_last = (nchunks - 1 - b) % nblocks == 0 && last_ord < j0 + nmy;  // the set's last (partial) chunk
            const int T_round = 64 * nmy - (has_last ? (int)(64 * nchunks - (a.nsrc - n0)) : 0);
            int top = 0;  // 1 + the highest slot this lane gave a record
            auto load1 = [&](const int c, const int64_t chunk, const T R, const T z) {
                const int64_t k = n0 + chunk * 64 + lane;
                const bool valid = k < a.nsrc;
                const int loc = c * 64 + lane;
                const int pos = T_round - 1 - loc;  // valid lanes: 0 <= pos
                // (incremental likelihood) not affected by the proposal: z above everything it changes; counted per wave
                const bool cached = inc && valid && (double)z < ic.zthr && (double)z != ic.dl1 && (double)z != ic.dl2;
                if (inc) {
                    const unsigned long long cb = __ballot(cached), vb = __ballot(valid);
                    n_cached += (unsigned int)__popcll(cb);
                    n_traced += (unsigned int)__popcll(vb & ~cb);
                }
                const int nl = which_layer(t, a.nlayers, z);
                T tt0 = T(0.0);
                if (nl == 1)
                    tt0 = straight_time(z, R, t.V[0]);  // (:94-97)
                if (valid) {
                    T r2c = T(0.0);
                    if (inc) {
                        // a source the chain's pending (accepted) proposal changed has its current residual in that proposal's
                        // row; the chain's mover row copies it to r2cur
                        const bool moved = ic.pending && k >= ic.pn0 &&
                                           !((double)z < ic.pzthr && (double)z != ic.pdl1 && (double)z != ic.pdl2);
                        if (moved) {
                            if (cached || ic.mover)
                                r2c = r2pend[k];
                            if (ic.mover)
                                r2cur[k] = r2c;
                        } else if (cached) {
                            r2c = r2cur[k];
                        }
                    }
                    if (cached) {
                        acc_add_double(acc, (double)r2c);  // its residual under the current model
                        rk[pos] = RTMI_RK_SKIP | RTMI_RK_DONE | (2u << RTMI_RK_NL_SHIFT);
                    } else if (nl == 1) {
                        const T r = a.d_obs[k] - tt0;
                        acc_add_double(acc, (double)(r * r));
                        if (inc)
                            r2prop[k] = r * r;
                        rk[pos] = RTMI_RK_SKIP | RTMI_RK_DONE | (2u << RTMI_RK_NL_SHIFT);
                    } else {
                        // (the start value p0 is computed by the first-evaluation pass, from the record)
                        rR[pos] = R;
                        rz[pos] = z;
                        rk[pos] = (uint32_t)loc | ((uint32_t)nl << RTMI_RK_NL_SHIFT);
                        top = pos + 1 > top ? pos + 1 : top;
                    }
                }
            };
#pragma unroll 1
            for (int cb = wib; cb < CH; cb += PF * nwib) {
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const int c = cb + j * nwib;
                    const int64_t chunk = b + (j0 + c) * nblocks;
                    if (c < CH && chunk < nchunks) {  // wave-uniform
                        const T R = Rw[j], z = zw[j];
                        Rw[j] = offset_of(j0, c + PF * nwib);
                        zw[j] = depth_of(j0, c + PF * nwib);
                        load1(c, chunk, R, z);
                    }
                }
            }
            top = wave_max_i32(top);
            if (lane == 0 && top > 0)
                atomicMax(&qctl[1], (uint32_t)top);
        } else {
            uint32_t mycnt = 0;
            if (j0 > 0) {
#pragma unroll
                for (int j = 0; j < PF; ++j)
                    zw[j] = depth_of(j0, wib + j * nwib);
                if constexpr (!DET) {
                    if (threadIdx.x < 64)
                        hist[threadIdx.x] = 0u;  // (the round before ended with a barrier; nothing has touched hist since)
                    __syncthreads();
                }
            }
            // The window of PF chunks is indexed statically (the loop over a wave's chunks is unrolled by the window):
            // a chunk's registers are reloaded for the chunk PF steps ahead as soon as they are consumed, and no
            // register is moved -- a move would make the compiler wait for every load in flight.
            auto pass1 = [&](const int c, const int64_t chunk, const T z) {
                const int64_t k = chunk * 64 + lane;
                const bool valid = k < a.nsrc;
                // incremental likelihood: a source the proposal does not affect is neither located nor sorted (its squared
                // residual is added from the cache in the second pass)
                const bool unaffected = inc && (double)z < ic.zthr && (double)z != ic.dl1 && (double)z != ic.dl2;
                int nl = unaffected ? 1 : which_layer(t, a.nlayers, z);
                if (!valid)
                    nl = 1;  // like a straight ray: never queued
                const int bin = bin_of(nl, nl_max);
                uint32_t rank = 0;  // DET: of this record among the wave's records of its bin (chunks in order, lanes in order)
                if constexpr (!DET) {
                    if (nl > 1)
                        rank = atomicAdd(&hist[bin], 1u);  // of this record among the workgroup's records of its bin, in no particular order
                }
                unsigned long long todo = DET ? __ballot(nl > 1) : 0ull;
                while (todo) {
                    const int lb = __builtin_amdgcn_readlane(bin, __ffsll((long long)todo) - 1);  // scalar
                    const unsigned long long grp = __ballot(nl > 1 && bin == lb);
                    const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)mycnt, lb);
                    if (nl > 1 && bin == lb)
                        rank = before + (uint32_t)__popcll(grp & lt_mask);
                    if (lane == lb)
                        mycnt += (uint32_t)__popcll(grp);
                    todo &= ~grp;
                }
                tnl[c * 64 + lane] = (uint32_t)nl | (rank << 11);  // nl <= 1025, rank < 8192
            };
#pragma unroll 1
            for (int cb = wib; cb < CH; cb += PF * nwib) {
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const int c = cb + j * nwib;
                    const int64_t chunk = b + (j0 + c) * nblocks;
                    if (c < CH && chunk < nchunks) {  // wave-uniform
                        const T z = zw[j];
                        zw[j] = depth_of(j0, c + PF * nwib);
                        pass1(c, chunk, z);
                    }
                }
            }
            if constexpr (DET)
                hist[wib * 64 + lane] = mycnt;
#pragma unroll
            for (int j = 0; j < PF; ++j) {  // in flight during the scan
                Rw[j] = offset_of(j0, wib + j * nwib);
                zw[j] = depth_of(j0, wib + j * nwib);
            }
            __syncthreads();
            if (DIAG && stamps && j0 == 0 && threadIdx.x == 0)
                stamps[4] = clock64();  // depths loaded, bins counted
            if (wib == 0) {
                uint32_t tot = 0;
                if constexpr (DET) {
                    for (int w = 0; w < nwib; ++w)
                        tot += hist[w * 64 + lane];
                } else {
                    tot = hist[lane];
                }
                uint32_t incl = tot;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t o = __shfl_up(incl, off, 64);
                    if (lane >= off)
                        incl += o;
                }
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                uint32_t run = total - incl;  // first slot of bin `lane` when bins are laid out descending
                if constexpr (DET) {
                    for (int w = 0; w < nwib; ++w) {
                        const uint32_t c = hist[w * 64 + lane];
                        hist[w * 64 + lane] = run;  // first slot of (wave w, bin lane)
                        run += c;
                    }
                } else {
                    hist[lane] = run;               // first slot of bin lane
                }
                if (lane == 0) {
                    qctl[0] = 0;
                    qctl[1] = total;
                    qctl[2] = 0;
                    qctl[3] = 0;
                    pl_ctl[0] = 0;
                    pl_ctl[1] = 0;
                    pl_ctl[2] = nwib > 4 ? nwib - 4 : 0;
                }
                for (int i = lane; i < RTMI_POOL; i += 64)
                    pl_ready[i] = 0;
            }
            __syncthreads();
            auto pass2 = [&](const int c, const int64_t chunk, const T R, const T z) {
                const int64_t k = chunk * 64 + lane;
                const bool valid = k < a.nsrc;
                const uint32_t packed = valid ? tnl[c * 64 + lane] : 1u;
                const int nl = (int)(packed & 0x7FFu);
                // slot = first slot of (this wave, the record's bin) + the record's rank there (counted in the first pass)
                const uint32_t pos = hist[(DET ? wib * 64 : 0) + bin_of(nl, nl_max)] + (packed >> 11);
                // (incremental likelihood) not affected by the proposal, as in the first pass; counted per wave
                const bool cached = inc && valid && (double)z < ic.zthr && (double)z != ic.dl1 && (double)z != ic.dl2;
                if (inc) {
                    const unsigned long long cb = __ballot(cached), vb = __ballot(valid);
                    n_cached += (unsigned int)__popcll(cb);
                    n_traced += (unsigned int)__popcll(vb & ~cb);
                }
                if (valid) {
                    T r2c = T(0.0);
                    if (inc) {
                        // residuals of the last accepted proposal that are still in r2prop: this source's moves to r2cur now
                        const bool moved = ic.pending && !((double)z < ic.pzthr && (double)z != ic.pdl1 && (double)z != ic.pdl2);
                        if (moved) {
                            r2c = r2pend[k];
                            r2cur[k] = r2c;
                        } else if (cached) {
                            r2c = r2cur[k];
                        }
                    }
                    if (cached) {
                        if constexpr (INC)
                            acc_add_double(acc, (double)r2c);  // not affected by the proposal: its residual under the current model
                    } else if (nl == 1) {
                        // straight ray in the top layer (:94-97): finished here
                        const T tt = straight_time(z, R, t.V[0]);
                        if (WANT_T)
                            a.d_timeP[k] = tt;
                        if (WANT_P)
                            a.d_p_final[k] = __builtin_nan("");
                        if (WANT_SS) {
                            const T r = a.d_obs[k] - tt;
                            if constexpr (INC) {
                                acc_add_double(acc, (double)(r * r));
                                if (inc)
                                    r2prop[k] = r * r;
                            } else {
                                ss = ss + r * r;
                            }
                        }
                    } else {
                        // (the start value p0 is computed by the first-evaluation pass, from the record)
                        rR[pos] = R;
                        rz[pos] = z;
                        rk[pos] = (uint32_t)(c * 64 + lane) | ((uint32_t)nl << RTMI_RK_NL_SHIFT);
                    }
                }
            };
#pragma unroll 1
            for (int cb = wib; cb < CH; cb += PF * nwib) {
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const int c = cb + j * nwib;
                    const int64_t chunk = b + (j0 + c) * nblocks;
                    if (c < CH && chunk < nchunks) {  // wave-uniform
                        const T R = Rw[j], z = zw[j];
                        Rw[j] = offset_of(j0, c + PF * nwib);
                        zw[j] = depth_of(j0, c + PF * nwib);
                        pass2(c, chunk, R, z);
                    }
                }
            }
        }
        __syncthreads();
        if (DIAG && stamps && j0 == 0 && threadIdx.x == 0)
            stamps[5] = clock64();  // records in their slots, p0 computed
        const int nq = __builtin_amdgcn_readfirstlane((int)qctl[1]);  // wave-uniform values are kept in SGPRs

        // ---- a2. first evaluation of every record, in sorted order (one layer count per wave) -------
        // GetPTime's branch (:136-153) is taken here, at full lane occupancy.  It also tells which
        // sources are long (bisection first: 2 + up to 20 + up to 16 further evaluations) and which are
        // short (Newton from p0, typically 2-3 more): the solve queue takes the long class first, so
        // that the drain at the end of the round -- when lanes can no longer be refilled -- is made of
        // short sources.  Sources that converge at p0 are finished here.
        for (int q0 = wib * 64; q0 < nq; q0 += nwib * 64) {
            const int q = q0 + lane;
            const bool valid = q < nq;
            Lane<T> s;
            const uint32_t key = valid ? rk[q] : (2u << RTMI_RK_NL_SHIFT);
            const int nl = (int)((key >> RTMI_RK_NL_SHIFT) & RTMI_RK_NL_MASK);
            const bool rec = valid && !(SORTED && (key & RTMI_RK_SKIP));  // (a slot without a record carries no source data)
            const T Rq = rec ? rR[q] : T(1.0), zq = rec ? rz[q] : T(1.0);
            const T p0 = initial_p0(t, nl - 1, Rq, zq, rec);  // GetPTime's head (:112-133)
            lane_setup(s, t, nl - 1, Rq, zq);
            lane_start(s, p0);
            const bool live = rec && (p0 == p0);  // NaN p0: the reference never returns -> -999
            if (!live)
                s.st = ST_DONE;
            const int nfw = wave_max_i32(live ? s.nfull : 0);
            T f, fp;
            eval_f_fp<ROWS>(t, s.xe, s.nfull, nfw, s.hvl, s.vvl, s.R, s.numer_ok, live, f, fp);
            {
                // every lane of the wave takes part (wave-uniform choice of the divide); dead lanes are restored
                const Lane<T> keep = s;
                lane_advance_init(t, s, f, fp, live);
                if (!live)
                    s = keep;
            }
            if (rec) {
                uint32_t k2 = key;
                if (s.st == ST_DONE) {
                    rp[q] = s.x;
                    k2 |= RTMI_RK_DONE;
                    if (s.conv)
                        k2 |= RTMI_RK_CONV;
                } else {
                    rp[q] = s.xe;  // BIS0: 1e-10; NEWTON: the first Newton iterate (x == xe)
                    if (s.st != ST_BIS0)
                        k2 |= RTMI_RK_SHORT | ((uint32_t)(RTMI_RK_CLASSES - 1) << RTMI_RK_CLASS_SHIFT);
                    else
                        k2 |= (uint32_t)trip_class(t, nl - 1, s.R, rz[q]) << RTMI_RK_CLASS_SHIFT;
                }
                rk[q] = k2;
            }
            // per-chunk class counts and the lane's rank within its class: what the solve order is built from
            {
                const bool todo = valid && s.st != ST_DONE;
                const int cls = (int)((rk[q < nq ? q : 0] >> RTMI_RK_CLASS_SHIFT) & (RTMI_RK_CLASSES - 1));
                unsigned long long w0 = 0, mine = 0;
#pragma unroll
                for (int k = 0; k < RTMI_RK_CLASSES; ++k) {
                    const unsigned long long bk = __ballot(todo && cls == k);
                    if (cls == k)
                        mine = bk;
                    w0 |= (unsigned long long)__popcll(bk) << (16 * k);
                }
                if (lane == 0)
                    ch_cnt[q0 >> 6] = w0;
                // finished records (class 7) follow the solve order in order[]: the time pass walks all of it
                const unsigned long long fin = __ballot(valid && !todo);
                if (valid)
                    tnl[q] = todo ? ((uint32_t)cls | ((uint32_t)__popcll(mine & lt_mask) << 3))
                                  : (7u | ((uint32_t)__popcll(fin & lt_mask) << 3));
            }
        }
        __syncthreads();
        if (DIAG && stamps && j0 == 0 && threadIdx.x == 0)
            stamps[6] = clock64();  // first evaluation done

        // ---- a3. exact solve order ----------------------------------------------------------------------
        // order[] = class 0, class 1, ... (most expected evaluations first; the short class last), each in slot
        // order (= descending layer index), built from per-chunk ballots (deterministic).  Longest first is
        // what keeps the end of the round short: the last records handed out are the cheapest ones.
        const int nchq = (nq + 63) >> 6;  // <= 127; the counts were taken in a2
        if (wib == 0) {
            // lane l owns chunks 2l and 2l+1: exclusive prefix of the counts (16-bit fields, no carry: <= 8128)
            const int c0 = 2 * lane, c1 = 2 * lane + 1;
            const unsigned long long n0 = c0 < nchq ? ch_cnt[c0] : 0ull;
            const unsigned long long n1 = c1 < nchq ? ch_cnt[c1] : 0ull;
            unsigned long long in_n = n0 + n1;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned long long on = __shfl_up(in_n, off, 64);
                if (lane >= off)
                    in_n += on;
            }
            const unsigned long long tot = __shfl(in_n, 63, 64);
            const unsigned long long ex_n = in_n - (n0 + n1);
            if (c0 < nchq)
                ch_start[c0] = ex_n;
            if (c1 < nchq)
                ch_start[c1] = ex_n + n0;
            if (lane == 0) {
                uint32_t run = 0;
                for (int k = 0; k < RTMI_RK_CLASSES; ++k) {
                    cls_base[k] = run;
                    run += (uint32_t)((tot >> (16 * k)) & 0xFFFFull);
                }
                cls_base[RTMI_RK_CLASSES] = run;  // records that need the solve loop
            }
        }
        __syncthreads();
        const int n_todo = __builtin_amdgcn_readfirstlane((int)cls_base[RTMI_RK_CLASSES]);
        for (int c = wib; c < nchq; c += nwib) {
            const int q = c * 64 + lane;
            const uint32_t v = q < nq ? tnl[q] : 0xFFFFFFFFu;  // class | rank within (chunk, class) << 3
            if (v != 0xFFFFFFFFu) {
                const int cls = (int)(v & 7u);
                const unsigned long long cs = ch_start[c];
                if (cls == 7) {
                    // finished by the first evaluation: after the solve order, in slot order (every chunk before c is full)
                    uint32_t todo_before = 0;
#pragma unroll
                    for (int k = 0; k < RTMI_RK_CLASSES; ++k)
                        todo_before += (uint32_t)((cs >> (16 * k)) & 0xFFFFull);
                    order[(uint32_t)n_todo + (uint32_t)(c * 64) - todo_before + (v >> 3)] = (uint16_t)q;
                } else {
                    const uint32_t st = (uint32_t)((cs >> (16 * cls)) & 0xFFFFull);
                    order[cls_base[cls] + st + (v >> 3)] = (uint16_t)q;
                }
            }
        }
        __syncthreads();
        if (DIAG && stamps && j0 == 0 && threadIdx.x == 0)
            stamps[1] = clock64();

        // ---- b. solve: every wave pulls from the workgroup's queue, lanes refill as they finish ----
        //
        // The solve phase is bound by its critical path as much as by throughput: a source may need 38
        // evaluations one after the other, and with four waves sharing a SIMD one trip of a wave takes ~10-12k
        // cycles of wall time against ~4.5k when the wave has the SIMD's issue slots to itself.  Hence two kinds
        // of waves (when the workgroup has more than four):
        //   express waves 0..3 (one per SIMD) prefer, over fresh records,
        //     the sources other waves hand to them through an LDS ring;
        //   local waves 4.. hand over ("promote") every source that has already taken `promote_age` trips -- the
        //     ones that turn out to be long -- and, in the drain, their last `retire_max` live lanes, then leave.
        // A handed-over source is resumable state: slot, iterate, bracket, solver state, trip counter.
        Lane<T> s;
        s.st = ST_DONE;
        s.nfull = 0;
        s.R = s.hvl = s.vvl = s.vmi = s.safe = s.clampv = s.x = s.xe = s.xb = s.dx = T(0.0);
        s.k = 0;
        s.conv = false;
        s.numer_ok = true;
        s.skip0 = false;
        int slot = 0;
        int age = 0;             // trips this lane's source has taken in this wave
        bool exhausted = false;  // wave-uniform: the queue head has passed n_todo
        const int tail_group = a.tail_group < n_todo / 6 ? a.tail_group : n_todo / 6;  // short rounds: keep every wave fed
        // hand-overs pay once a round holds at least a record per lane (drain) or two (promotion); below that
        // every wave is in its end-game from the start
        const int promote_age = n_todo >= 2 * (int)blockDim.x ? a.promote_age : 0;
        const bool use_pred = n_todo >= RTMI_PRED_MIN_PER_LANE * (int)blockDim.x && model_terms_nonneg(t);  // see lane_advance_solve
        const bool split = nwib > 4 && n_todo >= (int)blockDim.x && (a.retire_max > 0 || promote_age > 0);
        const bool local = split && wib >= 4;
        const bool express = split && wib < 4;
        // End-game concentration: the last records of the queue go to the first waves only (wave group g =
        // wib / 4 stops refilling once fewer than g * tail_group records remain).
        const int stop_below = (wib >> 2) * tail_group;
        bool left = false;  // (local) this wave has signed off from rg[3]
        int spins = 0;
        int nfull_wave = 0;
        // diagnostic occupancy counters (only touched when stamps are requested)
        unsigned long long dg_iters = 0, dg_busy = 0, dg_trips = 0, dg_terms = 0, dg_refills = 0, dg_t0 = 0, dg_t1 = 0;
        unsigned long long dg_sec[6] = {0, 0, 0, 0, 0, 0};  // ticks in: tail check, hand-over, adoption, refill, wave max + exit test, evaluation
        for (;;) {
            if (DIAG && fine)
                dg_t0 = clock64();
            unsigned long long idle = __ballot(s.st == ST_DONE);
            int nidle = __popcll(idle);
            bool changed = false;
            if (DIAG && fine) {
                dg_t1 = clock64();
                dg_sec[0] += dg_t1 - dg_t0;
                dg_t0 = dg_t1;
            }

            // -- local waves: hand over old sources (any time) or the last live lanes (drain) ------------
            if (local && nidle < 64) {
                const bool old_src = promote_age > 0 && s.st != ST_DONE && age >= promote_age;
                const bool drain = exhausted && 64 - nidle <= a.retire_max;
                const unsigned long long pm = drain ? ~idle : __ballot(old_src);
                const int np = __popcll(pm);
                if (np > 0 && (drain || np >= RTMI_PROMOTE_MIN || exhausted)) {
                    int base = -1;
                    if (lane == 0) {
                        int tl = pl_ctl[0];
#pragma unroll 1
                        for (;;) {
                            if (tl - pl_ctl[1] + np > RTMI_POOL)  // claimed-but-unread slots count as occupied
                                break;
                            const int prev = atomicCAS(pl_state + 2 * RTMI_POOL + 0, tl, tl + np);
                            if (prev == tl) {
                                base = tl;
                                break;
                            }
                            tl = prev;
                        }
                    }
                    base = __builtin_amdgcn_readfirstlane(base);  // lane 0's value, as a scalar
                    if (base >= 0) {
                        if ((pm >> lane) & 1ull) {
                            const int i = base + __popcll(pm & lt_mask);
                            const int e = i & (RTMI_POOL - 1), gen = i / RTMI_POOL + 1;
                            int guard = 0;  // the slot's previous tenant must have been read
#pragma unroll 1
                            while (pl_ready[e] != -(gen - 1) && ++guard < (1 << 22))
                                __builtin_amdgcn_s_sleep(1);
                            if (guard >= (1 << 22) && a.d_status)
                                atomicOr(a.d_status, 1);  // never observed: the ring cannot deadlock (see above)
                            pl_xe[e] = s.xe;
                            pl_x[e] = s.x;
                            pl_xb[e] = s.xb;
                            pl_dx[e] = s.dx;
                            pl_slot[e] = slot;
                            pl_state[e] = s.st | (s.k << 8) | (s.conv ? 1 << 16 : 0);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            pl_ready[e] = gen;
                            s.st = ST_DONE;
                        }
                        idle = __ballot(s.st == ST_DONE);
                        nidle = __popcll(idle);
                        changed = true;
                    }
                }
            }

            if (DIAG && fine) {
                dg_t1 = clock64();
                dg_sec[1] += dg_t1 - dg_t0;
                dg_t0 = dg_t1;
            }
            // -- express waves: adopt handed-over sources into idle lanes ----------------------------------
            if (express && nidle > 0 &&
                __builtin_amdgcn_readfirstlane(pl_ctl[1]) < __builtin_amdgcn_readfirstlane(pl_ctl[0])) {
                int base = 0, take = 0;
                if (lane == 0) {
                    int head = pl_ctl[1];
#pragma unroll 1
                    for (;;) {
                        const int avail = pl_ctl[0] - head;
                        take = avail < nidle ? avail : nidle;
                        if (take <= 0) {
                            take = 0;
                            break;
                        }
                        const int prev = atomicCAS(pl_state + 2 * RTMI_POOL + 1, head, head + take);
                        if (prev == head)
                            break;
                        head = prev;
                    }
                    base = head;
                }
                base = __builtin_amdgcn_readfirstlane(base);
                take = __builtin_amdgcn_readfirstlane(take);
                if (take > 0) {
                    const int r = __popcll(idle & lt_mask);
                    if (s.st == ST_DONE && r < take) {
                        const int i = base + r;
                        const int e = i & (RTMI_POOL - 1), gen = i / RTMI_POOL + 1;
                        int guard = 0;
#pragma unroll 1
                        while (pl_ready[e] != gen && ++guard < (1 << 22))
                            __builtin_amdgcn_s_sleep(1);
                        if (guard >= (1 << 22) && a.d_status)
                            atomicOr(a.d_status, 2);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        slot = pl_slot[e];
                        const int ps = pl_state[e];
                        s.xe = pl_xe[e];
                        s.x = pl_x[e];
                        s.xb = pl_xb[e];
                        s.dx = pl_dx[e];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        pl_ready[e] = -gen;  // slot read: free for the next tenant
                        const uint32_t key = rk[slot];
                        const int nl = (int)((key >> RTMI_RK_NL_SHIFT) & RTMI_RK_NL_MASK);
                        const T keep_xe = s.xe, keep_x = s.x, keep_xb = s.xb, keep_dx = s.dx;
                        lane_setup(s, t, nl - 1, rR[slot], rz[slot]);
                        s.xe = keep_xe;
                        s.x = keep_x;
                        s.xb = keep_xb;
                        s.dx = keep_dx;
                        s.st = ps & 0xFF;
                        s.k = (ps >> 8) & 0xFF;
                        s.conv = (ps >> 16) & 1;
                        age = 0;
                    }
                    idle = __ballot(s.st == ST_DONE);
                    nidle = __popcll(idle);
                    changed = true;
                }
            }

            if (DIAG && fine) {
                dg_t1 = clock64();
                dg_sec[2] += dg_t1 - dg_t0;
                dg_t0 = dg_t1;
            }
            // -- fresh records from the queue -----------------------------------------------------------------
            if (!exhausted && nidle >= a.refill_min && stop_below > 0 &&
                n_todo - __builtin_amdgcn_readfirstlane((int)qctl[0]) < stop_below) {
                exhausted = true;
                if (DIAG && stamps && j0 == 0 && lane == 0)
                    stamps[8 + RTMI_STAMPS_PER_WAVE * wib] = clock64();
            }
            if (!exhausted && nidle >= a.refill_min) {
                uint32_t base = 0;
                if (lane == 0)
                    base = atomicAdd(&qctl[0], (uint32_t)nidle);
                base = __builtin_amdgcn_readfirstlane(base);
                exhausted = (int)base + nidle >= n_todo;
                ++dg_refills;
                if (DIAG && stamps && j0 == 0 && exhausted && lane == 0)
                    stamps[8 + RTMI_STAMPS_PER_WAVE * wib] = clock64();  // this wave saw the queue run dry
                if (s.st == ST_DONE) {
                    const int qi = (int)base + __popcll(idle & lt_mask);
                    if (qi < n_todo) {
                        // long class, then short class, both in descending layer order
                        slot = (int)order[qi];
                        const uint32_t key = rk[slot];
                        const int nl = (int)((key >> RTMI_RK_NL_SHIFT) & RTMI_RK_NL_MASK);
                        lane_setup(s, t, nl - 1, rR[slot], rz[slot]);
                        lane_start(s, rp[slot]);
                        if (key & RTMI_RK_SHORT) {
                            s.st = ST_NEWTON;  // solve(p0) after its first iteration
                            s.k = 2;
                        } else if (s.skip0) {
                            lane_enter_bisection(s, false);  // f(1e-10) > 0 without evaluating it (lane_setup)
                            lane_skip_idle_trips(t, s, nl - 1, rz[slot] - t.D[nl - 2]);
                        } else {
                            s.st = ST_BIS0;    // solvebst(1d-10, ...): rp holds x1 = 1e-10
                        }
                        age = 0;
                    }
                }
                changed = true;
            }
            if (DIAG && fine) {
                dg_t1 = clock64();
                dg_sec[3] += dg_t1 - dg_t0;
                dg_t0 = dg_t1;
            }
            if (changed)
                nfull_wave = wave_max_i32(s.st != ST_DONE ? s.nfull : 0);

            if (__all(s.st == ST_DONE)) {
                bool finished = exhausted;
                if (exhausted && local && !left) {
                    if (lane == 0)
                        atomicSub(pl_state + 2 * RTMI_POOL + 2, 1);  // after this wave's hand-overs were published
                    left = true;
                }
                if (exhausted && express) {
                    // stay until every local wave has signed off and the ring is drained
                    finished = (__builtin_amdgcn_readfirstlane(pl_ctl[2]) <= 0 &&
                                __builtin_amdgcn_readfirstlane(pl_ctl[1]) >= __builtin_amdgcn_readfirstlane(pl_ctl[0])) ||
                               ++spins > (1 << 24);
                    if (spins > (1 << 24) && a.d_status && lane == 0)
                        atomicOr(a.d_status, 4);
                    if (!finished)
                        __builtin_amdgcn_s_sleep(4);
                }
                if (finished) {
                    if (DIAG && stamps && j0 == 0 && lane == 0) {
                        stamps[9 + RTMI_STAMPS_PER_WAVE * wib] = clock64();  // this wave's last source finished
                        stamps[10 + RTMI_STAMPS_PER_WAVE * wib] = dg_iters;   // evaluation trips of the wave
                        stamps[11 + RTMI_STAMPS_PER_WAVE * wib] = dg_busy;    // sum over trips of lanes carrying a live source
                        stamps[12 + RTMI_STAMPS_PER_WAVE * wib] = dg_trips;  // sum over trips of layer-loop length (nfull_wave + 1)
                        stamps[13 + RTMI_STAMPS_PER_WAVE * wib] = dg_terms | (dg_refills << 40);  // useful layer terms | refill events
                        for (int i = 0; i < 6; ++i)
                            stamps[14 + i + RTMI_STAMPS_PER_WAVE * wib] = dg_sec[i];
                    }
                    break;
                }
                continue;  // idle: the next trip refills / adopts
            }
            T f, fp;
            const bool was_busy = (s.st != ST_DONE);
            if (DIAG && fine) {
                dg_t1 = clock64();
                dg_sec[4] += dg_t1 - dg_t0;
                const unsigned long long bm = __ballot(was_busy);
                ++dg_iters;
                dg_busy += (unsigned long long)__popcll(bm);
                dg_trips += (unsigned long long)(nfull_wave + 1);
                int terms = was_busy ? s.nfull + 1 : 0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1)
                    terms += __shfl_xor(terms, off, 64);
                dg_terms += (unsigned long long)terms;
                dg_t0 = clock64();
            }
            eval_f_fp<ROWS>(t, s.xe, s.nfull, nfull_wave, s.hvl, s.vvl, s.R, s.numer_ok, was_busy, f, fp);
            if (DIAG && fine) {
                dg_t1 = clock64();
                dg_sec[5] += dg_t1 - dg_t0;
            }
            lane_advance_solve(t, s, f, fp, use_pred);
            ++age;
            if (was_busy && s.st == ST_DONE) {
                rp[slot] = s.x;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // p before the flag the time pass polls
                atomicOr(&rk[slot], s.conv ? (RTMI_RK_DONE | RTMI_RK_CONV) : RTMI_RK_DONE);  // one ds_or, no round trip
            }
        }
        // Rounds with express waves go on WITHOUT a barrier: the local waves, which leave the solve loop while the
        // express waves still finish the longest sources, start on the travel times at once (below).
        if (!split)
            __syncthreads();
        if (DIAG && stamps && j0 == 0 && threadIdx.x == 0)
            stamps[2] = clock64();

        // ---- c. travel times, in the order of the solve queue ------------------------------------------------
        // order[] lists every record: the solve queue (longest first) and then the records finished by their first
        // evaluation, each part in slot order, so a chunk of 64 still has one or two layer counts.  The chunks are
        // dealt to the waves statically.  With express waves, the local waves take all chunks but the last four and
        // start as soon as they leave the solve loop; a chunk is taken when all its sources are finished (flag
        // RTMI_RK_DONE, set after p is stored), otherwise the wave goes on to its next chunk and comes back -- the
        // head of the queue was handed out first and is long finished, so nearly the whole pass runs in the shadow of
        // the solve's tail.  The last four chunks (the last sources handed out) wait for the express waves, which
        // finish last anyway.  Squared residuals of the fused logL are parked in LDS and added up after the pass.
        {
            typedef __attribute__((address_space(3))) uint32_t lds_u32;
            volatile lds_u32 *rk_poll = (volatile lds_u32 *)rk;
            const int C = (nq + 63) >> 6;
            const int E = split ? (C < 4 ? C : 4) : 0;
            const int c_first = express ? C - E + wib : (split ? wib - 4 : wib);
            const int c_stride = split ? nwib - 4 : nwib;
            const int c_end = express ? (wib < E ? c_first + 1 : c_first) : C - E;
            const int nmine = c_end > c_first ? (c_end - c_first + c_stride - 1) / c_stride : 0;  // <= 11
            uint32_t pending = nmine > 0 ? (nmine >= 32 ? 0xFFFFFFFFu : (1u << nmine) - 1u) : 0u;
            int polls = 0;
#pragma unroll 1
            while (pending) {
                bool progress = false;
                uint32_t scan = pending;
#pragma unroll 1
                while (scan) {
                    const int j = __builtin_ctz(scan);
                    scan &= scan - 1u;
                    const int q0 = (c_first + j * c_stride) * 64;
                    const bool valid = q0 + lane < nq;
                    const int q = valid ? (int)order[q0 + lane] : 0;
                    if (split) {
                        if (!__all(!valid || (rk_poll[q] & RTMI_RK_DONE) != 0u))
                            continue;
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                    pending &= ~(1u << j);
                    progress = true;
                    const uint32_t key = valid ? rk[q] : (2u << RTMI_RK_NL_SHIFT);
                    const int il = (int)((key >> RTMI_RK_NL_SHIFT) & RTMI_RK_NL_MASK) - 1;
                    const bool rec = valid && !(SORTED && (key & RTMI_RK_SKIP));
                    const T p = rec ? rp[q] : T(0.0);
                    const T z = rec ? rz[q] : T(1.0);
                    const int nfw = wave_max_i32(rec ? il : 0);
                    T tt = travel_time(t, p, il, nfw, il, z, rec);
                    if (!(key & RTMI_RK_CONV))
                        tt = -T(999.0);
                    if (rec) {
                        const int loc = (int)(key & RTMI_RK_LOC_MASK);
                        const int64_t k = n0 + (b + (j0 + (loc >> 6)) * nblocks) * 64 + (loc & 63);
                        if (WANT_T)
                            a.d_timeP[k] = tt;
                        if (WANT_P)
                            a.d_p_final[k] = p;
                        if (WANT_SS) {
                            const T r = a.d_obs[k] - tt;
                            if constexpr (INC) {
                                acc_add_double(acc, (double)(r * r));
                                if (inc)
                                    r2prop[k] = r * r;
                            } else {
                                rR[q] = r * r;  // the record's R is not needed any more
                            }
                        }
                    }
                }
                if (!progress) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++polls > (1 << 22)) {  // never expected: flag it and leave the rest unwritten
                        if (a.d_status && lane == 0)
                            atomicOr(a.d_status, 4);
                        break;
                    }
                }
            }
        }
        __syncthreads();
        if (WANT_SS && !INC) {
            // the squared residuals, added up in SLOT order by all waves: an order that depends on nothing but the
            // sources and the launch geometry (not on the scheduling knobs, not on which chunk was ready first)
            for (int q = wib * 64 + lane; q < nq; q += nwib * 64)
                ss = ss + rR[q];
            __syncthreads();  // before the next round reuses the record area
        }
        if (DIAG && stamps && j0 == 0 && threadIdx.x == 0)
            stamps[3] = clock64();
    }

    if constexpr (INC) {
        // The sampler's form: order-independent sums (Acc192) -- butterfly over the lanes, the waves through LDS, one record
        // of four words per workgroup (two used); any order would give the same words.  Statistics ride along (no atomics:
        // the launches of a handle are stream-ordered and every workgroup owns its words).
        const bool counting = inc && a.d_inc_counts != nullptr;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            Acc192 o;
            o.w0 = __shfl_xor(acc.w0, off, 64);
            o.w1 = __shfl_xor(acc.w1, off, 64);
            acc_add(acc, o);
        }
        unsigned long long *wx = reinterpret_cast<unsigned long long *>(rR);  // the record area is free now
        if (lane == 0) {
            wx[4 * wib + 0] = acc.w0;
            wx[4 * wib + 1] = acc.w1;
            wx[4 * wib + 2] = n_traced;
            wx[4 * wib + 3] = n_cached;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            Acc192 tot = acc_zero();
            unsigned long long tr = 0, ca = 0;
            for (int w = 0; w < nwib; ++w) {
                Acc192 o;
                o.w0 = wx[4 * w + 0];
                o.w1 = wx[4 * w + 1];
                acc_add(tot, o);
                tr += wx[4 * w + 2];
                ca += wx[4 * w + 3];
            }
            if (a.acc_atomic) {
                // the row's accumulator: limbs of 48 bits in 64-bit words (no carry between words until 2^16 sums are in)
                unsigned long long *o = a.d_partial_x + 4 * ((int64_t)m * pstride);
                const unsigned long long v1 = tot.w1 & ~RTMI_ACC_FLAGS, fl = tot.w1 & RTMI_ACC_FLAGS;
                __hip_atomic_fetch_add(o + 0, tot.w0 & 0xFFFFFFFFFFFFull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(o + 1, (tot.w0 >> 48) | ((v1 & 0xFFFFFFFFull) << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(o + 2, v1 >> 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fl)
                    __hip_atomic_fetch_or(o + 3, fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                unsigned long long *o = a.d_partial_x + 4 * ((int64_t)m * pstride + bx);
                o[0] = tot.w0;
                o[1] = tot.w1;
                o[2] = 0ull;
                o[3] = 0ull;
            }
            if (counting) {
                if (a.acc_atomic) {
                    // (a row's shares are not bounded by pstride here: every workgroup of the LAUNCH owns a pair of words -- the
                    // launches of a handle are stream-ordered, so plain additions do)
                    unsigned long long *oc = a.d_inc_counts + 2 * (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) % a.count_words);
                    oc[0] += tr;
                    oc[1] += ca;
                } else {
                    unsigned long long *oc = a.d_inc_counts + 2 * ((int64_t)m * pstride + bx);
                    oc[0] += tr;
                    oc[1] += ca;
                }
            }
        }
    } else if (WANT_SS) {
        // workgroup reduction in a fixed order: lanes by xor-butterfly, then waves 0..nwib-1
        T *wsum = rR;  // the record area is free now (the round loop ended with a barrier)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            ss = ss + __shfl_xor(ss, off, 64);
        if (lane == 0)
            wsum[wib] = ss;
        __syncthreads();
        if (threadIdx.x == 0) {
            T tot = wsum[0];
            for (int w = 1; w < nwib; ++w)
                tot = tot + wsum[w];
            a.d_partial[(int64_t)m * gx + bx] = tot;
        }
    }
}

template <typename T, bool WANT_P, bool WANT_T, bool WANT_SS, bool DIAG = false, int TF = 0, bool INCR = false>
__global__ __launch_bounds__(RTMI_FUSED_MAX_BLOCK) void fused_kernel(FusedArgsT<T> a)
{
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];  // storage for either element type
    fused_body<T, WANT_P, WANT_T, WANT_SS, DIAG, TF, INCR>(a, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x,
                                                             reinterpret_cast<T *>(lds_raw));
}

// ---- the tempering sampler's sweep as ONE persistent launch ---------------------------------------------------------
// What a sweep is (rtmi_pt.hip): per proposal slot, the step of every chain (decide the previous proposal, draw the
// next one, write it as a model row) and then the likelihood of every proposed model over the shared sources.  As
// separate launches (two per slot) every chain waits at every slot for the slowest workgroup of ALL chains, and the
// launch boundaries themselves are a tenth of the sweep.  Here every chain owns gx workgroups for the whole sweep:
//   leading workgroup (row 0):  step of slot s  ->  publish (flag[chain] = epoch + s + 1)
//   all gx workgroups:          wait for the flag, evaluate their share of the sources (fused_body), add their residual
//                               sum to partial[chain][row], arrive (count[chain] += 1)
//   leading workgroup:          wait for gx arrivals, go on to slot s + 1
// so a chain whose sources happen to finish early -- or whose proposal fell outside the prior: no evaluation at all --
// runs ahead of the others; the chains meet again at the end of the kernel (the temperature exchange needs them all).
// Results are those of the per-slot launches bit for bit: same step function (pt_step_group), same likelihood body,
// same order of the residual sums.
// The workgroups must all be resident (they wait for each other): the launch is cooperative, and every wait is bounded --
// a wait that runs out sets the abort word and bit 16 of the fault word, and every workgroup leaves.
// Memory: what one workgroup writes for another (model row, NL, partial sums) is written, fenced at device scope
// (__threadfence) by its writers, and only then announced through an atomic; the reader takes the atomic with acquire
// semantics at device scope and invalidates the scalar cache before it reads.
__device__ __noinline__ bool pt_wait_reached(volatile unsigned int *word, unsigned int target, volatile unsigned int *abort_word,
                                                int *d_status)
{
    int polls = 0;
#pragma unroll 1
    for (;;) {
        // relaxed polls (an acquire per poll would invalidate the caches under the workgroups that are computing), one
        // acquire fence once the word is there
        const unsigned int v = __hip_atomic_load((unsigned int *)word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - target) >= 0) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);  // hipcc: agent scope
            return true;
        }
        if (__hip_atomic_load((unsigned int *)abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
            return false;
        __builtin_amdgcn_s_sleep(16);  // 2 ... 16 measured alike
        if (++polls > (1 << 23)) {  // seconds: never expected
            __hip_atomic_store((unsigned int *)abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d_status)
                atomicOr(d_status, 16);
            return false;
        }
    }
}

// The step is a real call (its own registers; the handle's device view is read from memory inside it): inlined next to
// the likelihood body it drove that body into scratch.
__device__ __noinline__ void pt_step_call(const rtmi_pt_dev::PtDev *dp, int chain, int decide, int sigma, int propose,
                                          int ivo, int ipar)
{
    const rtmi_pt_dev::PtDev d = *dp;
    if (d.K <= 16) {
        if (threadIdx.x < 16)
            rtmi_pt_dev::pt_step_group<16>(d, chain, decide, sigma, propose, ivo, ipar);
    } else if (d.K <= 32) {
        if (threadIdx.x < 32)
            rtmi_pt_dev::pt_step_group<32>(d, chain, decide, sigma, propose, ivo, ipar);
    } else if (threadIdx.x < 64) {
        rtmi_pt_dev::pt_step_group<64>(d, chain, decide, sigma, propose, ivo, ipar);
    }
}

__device__ __noinline__ bool pt_step_spec_call(const rtmi_pt_dev::PtDev *dp, int chain, int begin, int wg_total, double beta_alt)
{
    const rtmi_pt_dev::PtDev d = *dp;  // (called by the 64 lanes of wave 0: the lane groups take the window's slots side by side)
    bool diverged = false;
    if (threadIdx.x < 64) {
        if (d.K <= 16)
            diverged = rtmi_pt_dev::pt_step_spec<16>(d, chain, begin, 0, wg_total, beta_alt);
        else if (d.K <= 32)
            diverged = rtmi_pt_dev::pt_step_spec<32>(d, chain, begin, 0, wg_total, beta_alt);
        else
            diverged = rtmi_pt_dev::pt_step_spec<64>(d, chain, begin, 0, wg_total, beta_alt);
    }
    return diverged;
}

// ---- the speculative sweep, workgroups dealt to the chains IN PROPORTION TO THEIR SLOTS ------------------------------------
// With the speculative step (rtmi_pt_dev.hpp::pt_step_spec) a pass of a chain is up to nspec model rows side by side, and what
// bounds a sweep is the chain with the most nodes: it has the most slots, hence the most passes, and with the same number of
// workgroups for every chain (pt_sweep_kernel's layout) the others idle while it finishes.  Here the launch's workgroups (one
// per CU) are dealt to the chains by weight = slots of the chain's sweep (2 k + 1, as the chain stood when its last sweep
// ended): every chain at least one, the rest in proportion, so that passes x time-per-pass comes out about equal for all.
// The weights are read from walloc[parity] -- written by the chains' leading workgroups at the end of the sweep before, into the
// half nobody reads during that sweep -- so every workgroup of the launch computes the same layout.  Within a chain the
// protocol is pt_sweep_kernel's: the leading workgroup takes the step and publishes it (flag), the chain's workgroups deal
// themselves to the rows that go through the likelihood (workgroup w of gxc evaluates the a-th such row, a = w mod nact, as
// share w / nact of ceil((gxc - a) / nact) -- or, with fewer workgroups than rows, every nact-th row on its own), add their
// exact sums into the rows' accumulators and arrive (count).  A window of proposals that all lie below the deepest source needs
// no pass at all: the leading workgroup takes the next step at once.
// (The kernel around the likelihood body keeps next to nothing in registers: the body is at its register limit, and whatever
// lives across it is paid for in spills inside its loops.  The layout, the handshakes and the steps are real calls; what a
// workgroup has to remember between them -- its chain, its row, the pass -- sits in LDS.)
enum { PROP_GO = 0, PROP_CHAIN, PROP_ROW, PROP_GX, PROP_BASE, PROP_NACT, PROP_BEGIN, PROP_SWDONE, PROP_XPEND, PROP_XLP, PROP_T0, PROP_WAIT, PROP_WORDS = 12 };

__device__ __noinline__ void prop_layout(const PtSweepArgs *wp, int parity, volatile int *ctl)
{
    // first_c = c + floor(R P_c / W), R = nwg - nchains spare workgroups, P_c = weights before chain c; wave 0 calls
    const int lane = threadIdx.x & 63, nwg = gridDim.x, nchains = wp->nchains;
    const int *wts = wp->walloc + (size_t)parity * nchains;
    long long W = 0;
    for (int c0 = 0; c0 < nchains; c0 += 64) {
        int wc = c0 + lane < nchains ? wts[c0 + lane] : 0;
        wc = (c0 + lane < nchains && wc < 1) ? 1 : wc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            wc += __shfl_xor(wc, off, 64);
        W += wc;
    }
    const long long R = nwg - nchains;
    // consecutive blockIdx values go to consecutive XCDs (8 of them, an L2 each): the layout is made over the workgroups in
    // XCD-major order, so that a chain's workgroups -- which pass model rows, residual rows and sums to each other every pass
    // -- sit on one XCD where its share allows
    const long long vb = (nwg & 7) == 0 ? (long long)(blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3) : (long long)blockIdx.x;
    if ((nwg & 7) == 0 && (nchains & 7) == 0 && nchains <= 64 && nwg >= nchains) {
        // NO CHAIN ON TWO XCDs: the chains are dealt to the XCDs first (nchains / 8 each, in order), and an XCD's workgroups to ITS
        // chains by weight.  With one proportional layout over the whole launch a chain at an XCD boundary had its workgroups on two
        // L2s -- every hand-over of the pass through memory -- and ran a fifth slower than the others (RTMI_PT_QSTATS=2,
        // profiles/r04aw/chains2.txt: own share 109 against 95 us, 45 against 31 us waiting for its workgroups); in a multi-sweep
        // launch every other chain then waits for that one, round after round (120 us per sweep at its partner's mailbox).
        const int cx = nchains >> 3, wx = nwg >> 3;
        const int c = lane;
        int wc = c < nchains ? wts[c] : 0;
        wc = (c < nchains && wc < 1) ? 1 : wc;
        int incl = wc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += o;
        }
        const int x = c < nchains ? c / cx : 0, gs = x * cx;             // the chain's XCD, the first chain of that XCD
        const int sb = __shfl(incl, gs > 0 ? gs - 1 : 0, 64);            // (every lane takes part in the shuffle: its source lane must)
        const int before = gs > 0 ? sb : 0;                               // weights of the XCDs before
        const int Wx = __shfl(incl, gs + cx - 1, 64) - before;           // weights of this XCD's chains
        const int P = incl - wc - before;                                 // ... of its chains before this one
        const long long Rx = wx - cx;
        const long long first = (long long)x * wx + (c - gs) + (Rx * P) / Wx;
        const long long next = (c - gs) + 1 < cx ? (long long)x * wx + (c - gs) + 1 + (Rx * (P + wc)) / Wx : (long long)(x + 1) * wx;
        if (c < nchains && vb >= first && vb < next) {
            ctl[PROP_CHAIN] = c;
            ctl[PROP_ROW] = (int)(vb - first);
            ctl[PROP_GX] = (int)(next - first);
            ctl[PROP_BASE] = (int)__hip_atomic_load(wp->count + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ctl[PROP_BEGIN] = 1;   // the chain's next step is the first of a sweep
            ctl[PROP_SWDONE] = 0;  // sweeps of this launch the chain has completed
            ctl[PROP_XPEND] = 0;   // > 0: the exchange after that sweep of the launch is posted, not completed yet
            ctl[PROP_T0] = (int)(unsigned int)__builtin_amdgcn_s_memrealtime();
            ctl[PROP_WAIT] = 0;
        }
        return;
    }
    long long carry = 0;
    for (int c0 = 0; c0 < nchains; c0 += 64) {
        const int c = c0 + lane;
        int wc = c < nchains ? wts[c] : 0;
        wc = (c < nchains && wc < 1) ? 1 : wc;
        int incl = wc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += o;
        }
        const long long P = carry + incl - wc;
        const long long first = c + (R * P) / W;
        const long long next = c + 1 < nchains ? (c + 1) + (R * (P + wc)) / W : nwg;
        if (c < nchains && vb >= first && vb < next) {
            ctl[PROP_CHAIN] = c;
            ctl[PROP_ROW] = (int)(vb - first);
            ctl[PROP_GX] = (int)(next - first);
            ctl[PROP_BASE] = (int)__hip_atomic_load(wp->count + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ctl[PROP_BEGIN] = 1;   // the chain's next step is the first of a sweep
            ctl[PROP_SWDONE] = 0;  // sweeps of this launch the chain has completed
            ctl[PROP_XPEND] = 0;   // > 0: the exchange after that sweep of the launch is posted, not completed yet
            ctl[PROP_T0] = (int)(unsigned int)__builtin_amdgcn_s_memrealtime();
            ctl[PROP_WAIT] = 0;
        }
        carry += __shfl(incl, 63, 64);
    }
}

// A chain that has completed a sweep of a multi-sweep launch: the temperature exchange of the round with its PARTNER alone, then
// the sample record (wave 0 of the chain's leading workgroup calls; lane 0 does the exchange).  The ladder has one chain per level
// (levels 0 .. nct - 1, every chain local), so a chain's position in the exchange's (level, index) order IS its level, the pair of
// the round is (a, a + 1) with a = round parity + 2 j, and the decision is rtmi_pt_rules.h::swap_pair's: same draw, same ratio.
// The two chains find each other through a mailbox per level: logL and chain of the level's occupant, valid once its sequence word
// says round + 1; both evaluate the same rule on the same numbers and each moves ITSELF.  A chain waits for nobody but its partner,
// so the chains drift apart by a sweep or so and nobody waits for the unluckiest chain of every sweep (profiles/r04z/chains_4.txt:
// a chain's sweep takes 535 us on average, the slowest of 32 took 690 us, sweep after sweep).
//
// IN TWO HALVES.  Even so a chain spent a third of its time waiting for its partner (RTMI_PT_QSTATS=2, profiles/r04au/chains.txt:
// 190 of 595 us per sweep).  But nothing of a sweep's FIRST pass depends on the temperature: proposals, bounds, model rows and
// their likelihoods do not, and a trivial move's ratio is lp + beta (L* - logL) with L* - logL = 0 exactly at a sweep's start.  So
// a chain POSTS its mailbox entry when its sweep is complete (prop_exchange_post), lays out and evaluates the first window of its
// next sweep, and COMPLETES the exchange (prop_exchange_complete: wait for the partner's entry, decide, move, record the sample --
// the chain's state is still that of the completed sweep: the first step of a sweep decides nothing) right before the first
// DECISION of that sweep.  The wait for the partner hides behind a pass.
// Mailbox sets: round mod 4.  A chain's entry of round R is overwritten by the entry of round R + 4 of whoever sits on that level
// then, who by then has completed round R + 3 >= R + 2 -- the round in which the pairs are those of round R again, and whose
// completion needs the entry of a chain that had itself completed round R + 1, hence R: the reader of the old entry is through
// with it.  (With two sets a chain whose next sweep happens to need no pass at all could overwrite its entry before a slower
// partner, busy with its own first pass, had read it.)  A sequence word that is AHEAD of its reader is a fault, never a guess.
// Returns the partner's level (the chain's temperature after the exchange is its own or that level's), -1 without a partner.
__device__ __noinline__ int prop_exchange_post(const rtmi_pt_dev::PtDev *dp, const PtSweepArgs *wp, int chain, int done)
{
    const long long total = wp->sweeps_done + done;  // sweeps of the handle, this one included
    if (!(wp->swap_every > 0 && total % wp->swap_every == 0))
        return -1;
    const unsigned long long rnd = wp->round + (unsigned long long)(total / wp->swap_every - wp->sweeps_done / wp->swap_every) - 1ull;
    const int nct = wp->nct, par = (int)(rnd & 1ull), buf = (int)(rnd & 3ull);
    const int L = __hip_atomic_load(wp->level + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int a = ((L - par) & 1) == 0 ? L : L - 1;
    if (!(a >= par && a + 1 < nct))
        return -1;
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long mine = __hip_atomic_load(reinterpret_cast<unsigned long long *>(dp->logL + chain), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(wp->mb_logl + (size_t)buf * nct + L), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(wp->mb_chain + (size_t)buf * nct + L, chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        __hip_atomic_store(wp->mb_seq + (size_t)buf * nct + L, rnd + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    return L == a ? a + 1 : a;
}

__device__ __noinline__ void prop_exchange_complete(const rtmi_pt_dev::PtDev *dp, const PtSweepArgs *wp, int chain, int done, int *d_status,
                                                    volatile int *ctl)
{
    namespace R = rtmi_pt_rules;
    const int lane = threadIdx.x & 63;
    const long long total = wp->sweeps_done + done;  // sweeps of the handle, the completed one included
    int lev = 0;
    if (wp->swap_every > 0 && total % wp->swap_every == 0) {
        int newlev = -1;
        if (lane == 0) {
            const unsigned long long rnd = wp->round + (unsigned long long)(total / wp->swap_every - wp->sweeps_done / wp->swap_every) - 1ull;
            const int nct = wp->nct, par = (int)(rnd & 1ull), buf = (int)(rnd & 3ull);
            const int L = __hip_atomic_load(wp->level + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int a = ((L - par) & 1) == 0 ? L : L - 1;
            newlev = L;
            if (a >= par && a + 1 < nct) {
                // (this chain's own entry -- what prop_exchange_post wrote: the chain's logL has not changed since)
                const double mine = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(wp->mb_logl + (size_t)buf * nct + L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                const int Lp = L == a ? a + 1 : a;
                unsigned long long *sq = wp->mb_seq + (size_t)buf * nct + Lp;
                unsigned long long got = 0ull;
                int polls = 0;
                bool ok = true;
                const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
                for (;;) {
                    got = __hip_atomic_load(sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (got >= rnd + 1ull) {
                        const unsigned long long tw = __builtin_amdgcn_s_memrealtime() - tw0;
                        ctl[PROP_WAIT] += (int)tw;  // (100 MHz ticks: the launch's weights for the next one, prop_lead)
                        if (wp->stats)  // (diagnostic: how long the chain waited for its partner's entry)
                            wp->stats[16 + 8 * (size_t)chain + 7] += tw;
                        break;
                    }
                    if (__hip_atomic_load(wp->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || ++polls > (1 << 23)) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(16);
                }
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                if (!ok || got != rnd + 1ull) {  // never expected: the partner is gone, or its mailbox is rounds ahead of its reader
                    __hip_atomic_store(wp->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (d_status)
                        atomicOr(d_status, 16);
                } else {
                    const double theirs = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(wp->mb_logl + (size_t)buf * nct + Lp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    const double l1 = L == a ? mine : theirs, l2 = L == a ? theirs : mine;
                    const double b1 = wp->beta_of_level[a], b2 = wp->beta_of_level[a + 1];
                    R::Philox p = R::philox_make(wp->seed, 0xFFFFFFFFu, (uint32_t)rnd, ((uint64_t)(rnd >> 32) << 32) | (uint32_t)a);
                    const double u = R::philox_uniform(p);
                    double lr = (b2 - b1) * (l1 - l2);
                    if (lr > 50.0)
                        lr = 50.0;
                    const bool acc = b1 != b2 && u <= exp(lr);
                    if (L == a) {  // the pair is counted once
                        atomicAdd((unsigned long long *)&wp->swstats[1], 1ull);
                        if (acc)
                            atomicAdd((unsigned long long *)&wp->swstats[0], 1ull);
                    }
                    if (acc) {
                        newlev = Lp;
                        __hip_atomic_store(wp->level + chain, newlev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        dp->beta[chain] = wp->beta_of_level[newlev];
                    }
                }
            }
        }
        lev = __shfl(newlev, 0, 64);
    } else {
        lev = wp->level ? __hip_atomic_load(wp->level + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
    }
    // the chains at T = 1 go to the sample ring: {sweep, chain, k, sd, logL, z[K], v[K]} (as pt_record_kernel, rtmi_pt.hip)
    if (wp->record_every > 0 && wp->s_cap > 0 && (wp->s0 + done) % wp->record_every == 0 && lev == 0) {
        int slot = 0;
        if (lane == 0)
            slot = atomicAdd(wp->s_count, 1);
        slot = __shfl(slot, 0, 64);
        if (slot < wp->s_cap) {
            const int K = dp->K, k = __hip_atomic_load(dp->k + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double *o = wp->s_val + (size_t)slot * (2 + 2 * K);
            if (lane == 0) {
                wp->s_meta[(size_t)slot * 3 + 0] = total;
                wp->s_meta[(size_t)slot * 3 + 1] = chain;
                wp->s_meta[(size_t)slot * 3 + 2] = k;
                o[0] = dp->sd[chain];
                o[1] = dp->logL[chain];
            }
            for (int j = lane; j < k; j += 64) {
                o[2 + j] = dp->z[(size_t)chain * K + j];
                o[2 + K + j] = dp->v[(size_t)chain * K + j];
            }
        }
    }
}

// The leading workgroup's part of a pass (wave 0 calls): the chain's step(s), the weight of the chain for the next launch's
// layout once a sweep is complete -- and, in a launch of several sweeps, its half of the exchange and the next sweep's first
// step right behind it -- then the flag.  Leaves fin | nact in LDS.
__device__ __noinline__ void prop_lead(const rtmi_pt_dev::PtDev *dp, const PtSweepArgs *wp, volatile int *ctl, int pass, unsigned int epoch,
                                       int parity, int *d_status)
{
    const int chain = ctl[PROP_CHAIN], gx = ctl[PROP_GX];
    const int nsw = wp->nsweeps > 1 ? wp->nsweeps : 1;
    int guard = 0, fin_ = 0, nact_ = 0;
#pragma unroll 1
    for (;;) {
        // While the exchange of the sweep before is posted but not completed the chain's temperature is its own or its partner's
        // level's: the step is asked to go on only where its decisions come out the same under both (pt_step_spec, beta_alt);
        // where they do not -- or where the T = 1 sample of that sweep has to be taken from the state the chain still has -- the
        // exchange is completed first (the wait for the partner).
        double beta_alt = __builtin_nan("");
        if (ctl[PROP_XPEND] != 0 && !ctl[PROP_BEGIN]) {
            const int xd = ctl[PROP_XPEND], lp = ctl[PROP_XLP];
            const bool record_due = wp->record_every > 0 && wp->s_cap > 0 && (wp->s0 + xd) % wp->record_every == 0;
            if (record_due || lp < 0) {
                prop_exchange_complete(dp, wp, chain, xd, d_status, ctl);
                ctl[PROP_XPEND] = 0;
                __threadfence();
                __builtin_amdgcn_s_dcache_inv();
            } else {
                beta_alt = wp->beta_of_level[lp];
            }
        }
        // (begin = 2: the first step of a sweep whose temperature is not known yet -- it decides nothing, and lays its window
        // out only where that does not depend on the temperature: pt_step_spec)
        const bool diverged = pt_step_spec_call(dp, chain, ctl[PROP_BEGIN] ? (ctl[PROP_XPEND] != 0 ? 2 : 1) : 0, gx, beta_alt);
        if (__any(diverged)) {  // a decision depends on which temperature it is: nothing was written; the partner, then again
            prop_exchange_complete(dp, wp, chain, ctl[PROP_XPEND], d_status, ctl);
            ctl[PROP_XPEND] = 0;
            __threadfence();
            __builtin_amdgcn_s_dcache_inv();
            continue;
        }
        __threadfence();                  // the rows are out before the flag; and this workgroup (one L1) reads them afresh, too
        __builtin_amdgcn_s_dcache_inv();  // ... and so does a step that follows at once (the chain's words are wave-uniform: scalar loads)
        ctl[PROP_BEGIN] = 0;
        fin_ = __hip_atomic_load(dp->fin + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nact_ = __hip_atomic_load(dp->nact + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (fin_) {
            if (ctl[PROP_XPEND] != 0) {  // the chain must know its level before it posts the next round's entry
                prop_exchange_complete(dp, wp, chain, ctl[PROP_XPEND], d_status, ctl);
                ctl[PROP_XPEND] = 0;
                __threadfence();
                __builtin_amdgcn_s_dcache_inv();
            }
            const int done = ctl[PROP_SWDONE] + 1;
            ctl[PROP_SWDONE] = done;
            if ((threadIdx.x & 63) == 0) {
                // The weight of this chain in the NEXT launch's layout.  One sweep per launch: its slots (2 k + 1).  A launch of
                // several sweeps moves at the pace of its slowest chain and the others wait at their partners' mailboxes: the
                // chain asks for workgroups in proportion to the ones it has x the time it was NOT waiting (RTMI_PT_ADAPT=0: slots).
                int wnext = 2 * __hip_atomic_load(dp->k + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                if (nsw > 1 && wp->adapt && done >= nsw) {
                    const unsigned int dur = (unsigned int)__builtin_amdgcn_s_memrealtime() - (unsigned int)ctl[PROP_T0];
                    const unsigned int waited = (unsigned int)ctl[PROP_WAIT];
                    const unsigned int busy = dur > waited ? dur - waited : 1u;
                    // (per mille of the launch, times the workgroups it had: sums stay far below 2^31 for any launch)
                    const unsigned long long pm = (1000ull * busy) / (dur ? dur : 1u);
                    wnext = (int)((unsigned long long)gx * (pm < 1 ? 1 : pm));
                }
                if (nsw == 1 || done >= nsw)
                    wp->walloc[(size_t)(1 - parity) * wp->nchains + chain] = wnext;
            }
            if (nsw > 1) {
                const int lp = prop_exchange_post(dp, wp, chain, done);
                if (done < nsw && __hip_atomic_load(wp->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    ctl[PROP_XPEND] = done;  // completed when a decision needs the temperature, at the latest when the next sweep is complete
                    ctl[PROP_XLP] = lp;
                    ctl[PROP_BEGIN] = 1;     // the next sweep's first window at once
                    guard = 0;
                    continue;
                }
                prop_exchange_complete(dp, wp, chain, done, d_status, ctl);  // the launch's last sweep: now
                __threadfence();
                __builtin_amdgcn_s_dcache_inv();
            }
            break;
        }
        if (nact_ > 0 || ++guard > 4 * RTMI_PT_MAX_K)
            break;
    }
    if ((threadIdx.x & 63) == 0)
        __hip_atomic_store(wp->flag + chain, epoch + (unsigned int)pass + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// After the flag: what the chain does in this pass (thread 0 of every workgroup of the chain).  ctl[PROP_NACT] = rows to
// evaluate, 0 = the chain's sweep is complete (or a guard ran out: flagged).
__device__ __noinline__ void prop_read_pass(const rtmi_pt_dev::PtDev *dp, volatile int *ctl, int *d_status)
{
    const int chain = ctl[PROP_CHAIN];
    const int fin = __hip_atomic_load(dp->fin + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int nact = __hip_atomic_load(dp->nact + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!fin && nact <= 0 && d_status)
        atomicOr(d_status, 16);  // never expected: the leading workgroup's guard ran out
    ctl[PROP_NACT] = (fin || nact <= 0) ? 0 : nact;
}

// The end of a pass (thread 0): this workgroup's arrival; the leading workgroup waits for the chain's.
__device__ __noinline__ void prop_arrive(const PtSweepArgs *wp, volatile int *ctl, int pass, int *d_status)
{
    const int chain = ctl[PROP_CHAIN];
    __threadfence();
    __hip_atomic_fetch_add(wp->count + chain, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (ctl[PROP_ROW] == 0)
        ctl[PROP_GO] = pt_wait_reached(wp->count + chain, (unsigned int)ctl[PROP_BASE] + (unsigned int)(pass + 1) * (unsigned int)ctl[PROP_GX],
                                       wp->abort_, d_status) ? 1 : 0;
}

template <int TF>
__global__ __launch_bounds__(RTMI_FUSED_MAX_BLOCK) void pt_sweep_prop_kernel(FusedArgsT<double> a, const rtmi_pt_dev::PtDev *dp, const PtSweepArgs *wp,
                                                                             unsigned int epoch, int parity, int max_pass, int lds_ctl)
{
    using namespace rtmi_pt_dev;
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    volatile int *ctl = reinterpret_cast<volatile int *>(reinterpret_cast<char *>(lds_raw) + lds_ctl);
    // (diagnostic, RTMI_PT_QSTATS: thread 0 keeps 64-bit sums of 100 MHz ticks behind the control words: [0] in the likelihood,
    // [1] in steps, [2] waiting for the chain's flag or arrivals, [3] passes, [4] steps' passes, [5] start)
    volatile unsigned long long *dg = reinterpret_cast<volatile unsigned long long *>(reinterpret_cast<char *>(lds_raw) + lds_ctl + 4 * PROP_WORDS);
    const bool diag = wp->stats != nullptr;
    if (diag && threadIdx.x == 0) {
        dg[0] = dg[1] = dg[2] = dg[3] = dg[4] = 0ull;
        dg[5] = __builtin_amdgcn_s_memrealtime();
    }
    if (wp->poison_lds) {  // (diagnostic, RTMI_PT_POISON_LDS: what the likelihood body reads of LDS it has not written shows up as NaN)
        for (int q = threadIdx.x; q < lds_ctl / 8; q += blockDim.x)
            reinterpret_cast<unsigned long long *>(lds_raw)[q] = 0x7FF8DEADBEEF0000ull + (unsigned)q;
        __syncthreads();
    }
    if (threadIdx.x < 64)
        prop_layout(wp, parity, ctl);
    __syncthreads();
#pragma unroll 1
    for (int pass = 0; pass <= max_pass; ++pass) {
        if (diag && threadIdx.x == 0)
            dg[6] = __builtin_amdgcn_s_memrealtime();
        if (ctl[PROP_ROW] == 0) {
            if (threadIdx.x < 64)
                prop_lead(dp, wp, ctl, pass, epoch, parity, a.d_status);
            if (diag && threadIdx.x == 0) {
                dg[1] += __builtin_amdgcn_s_memrealtime() - dg[6];
                dg[4] += 1ull;
            }
        } else if (threadIdx.x == 0) {
            ctl[PROP_GO] = pt_wait_reached(wp->flag + ctl[PROP_CHAIN], epoch + (unsigned int)pass + 1u, wp->abort_, a.d_status) ? 1 : 0;
            if (diag)
                dg[2] += __builtin_amdgcn_s_memrealtime() - dg[6];
        }
        __syncthreads();
        if (ctl[PROP_ROW] != 0 && !ctl[PROP_GO])
            return;
        if (threadIdx.x == 0)
            prop_read_pass(dp, ctl, a.d_status);  // (written before the flag, read after it)
        __builtin_amdgcn_s_dcache_inv();
        __syncthreads();
        const int nact = ctl[PROP_NACT];
        if (nact <= 0) {  // the sweep of this chain is complete
            if (diag && threadIdx.x == 0) {
                unsigned long long *st = wp->stats;
                __hip_atomic_fetch_add(st + 0, dg[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(st + 1, dg[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(st + 2, dg[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(st + 3, dg[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(st + 4, dg[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long life = __builtin_amdgcn_s_memrealtime() - dg[5];
                __hip_atomic_fetch_add(st + 5, life, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ctl[PROP_ROW] == 0) {  // per chain (its leading workgroup): ticks, sweeps, the longest, passes
                    unsigned long long *q = st + 16 + 8 * (size_t)ctl[PROP_CHAIN];
                    q[0] += life;
                    q[1] += (unsigned long long)(wp->nsweeps > 1 ? wp->nsweeps : 1);
                    q[2] = life > q[2] ? life : q[2];
                    q[3] += dg[3];
                    q[4] += dg[0];   // the leading workgroup's own likelihood shares
                    q[5] += dg[2];   // ... its waits for the chain's other workgroups
                    q[6] += dg[1];   // ... its steps (in a multi-sweep launch: with the exchanges and their waits)
                }
            }
            return;
        }
        if (diag && threadIdx.x == 0)
            dg[6] = __builtin_amdgcn_s_memrealtime();
        {
            // enough workgroups: this one is share row / nact of ceil((gx - a) / nact) of the a-th row, a = row mod nact;
            // fewer workgroups than rows: it takes the rows row, row + gx, ... each on its own.  (One call site: the body is
            // inlined, and a second copy of it doubles the kernel and its spills.)
            const int row = ctl[PROP_ROW], gx = ctl[PROP_GX];
            const bool many = nact <= gx;
#pragma unroll 1
            for (int ai = many ? row % nact : row; ai < nact; ai += many ? nact : gx) {
                const int e = __builtin_amdgcn_readfirstlane(dp->alist[ctl[PROP_CHAIN] * wp->nspec + ai]);
                fused_body<double, false, false, true, false, TF, true, true>(a, ctl[PROP_CHAIN] * wp->nspec + e, many ? ctl[PROP_ROW] / nact : 0,
                                                                               many ? (ctl[PROP_GX] - ai + nact - 1) / nact : 1,
                                                                               reinterpret_cast<double *>(lds_raw));
                __syncthreads();
            }
        }
        if (threadIdx.x == 0) {
            if (diag) {
                const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
                dg[0] += t1 - dg[6];
                dg[3] += 1ull;
                dg[6] = t1;
            }
            prop_arrive(wp, ctl, pass, a.d_status);
            if (diag)
                dg[2] += __builtin_amdgcn_s_memrealtime() - dg[6];
        }
        __syncthreads();
        if (ctl[PROP_ROW] == 0) {
            if (!ctl[PROP_GO])
                return;
            __builtin_amdgcn_s_dcache_inv();
        }
    }
    // never expected: a chain that is not complete after 2 k + 2 passes
    if (threadIdx.x == 0 && a.d_status)
        atomicOr(a.d_status, 16);
}

template <int TF, bool INCR>
__global__ __launch_bounds__(RTMI_FUSED_MAX_BLOCK) void pt_sweep_kernel(FusedArgsT<double> a, const rtmi_pt_dev::PtDev *dp, PtSweepArgs w)
{
    using namespace rtmi_pt_dev;
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    int *const d_status = a.d_status;
    volatile int &go_on = *reinterpret_cast<volatile int *>(reinterpret_cast<char *>(lds_raw) + w.lds_ctl);  // past the likelihood's LDS
    const int gx = w.gx;
    int chain, row;
    if (w.xcd_map) {
        // consecutive blockIdx values go to consecutive XCDs: give a chain gx workgroups of ONE XCD, so that what they
        // pass to each other stays in that XCD's L2
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        chain = xcd * (w.nchains >> 3) + loc / gx;
        row = loc % gx;
    } else {
        chain = blockIdx.x / gx;
        row = blockIdx.x % gx;
    }
    // Slots with an evaluation: birth/death, then (node 0, v), (node 1, z), (node 1, v), ...  The host sizes the sweep for
    // the largest k any chain can have by now (a bound); a chain's own k is final once its birth/death slot is decided (the
    // step of slot 1), and the slots of nodes it does not have propose nothing -- the chain leaves them out (S_c below).
    int S_c = 2 * w.nslots_k;
    volatile unsigned int *flag = w.flag + chain, *count = w.count + chain, *abort_word = w.abort_;
    // arrivals are counted from where the word stands at the start of this launch (the previous launch has finished)
    const unsigned int count_base = __hip_atomic_load((unsigned int *)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
    for (int s = 0; s < S_c; ++s) {
        const int decide = s == 0 ? (int)PT_NONE : (s == 1 ? (int)PT_BD : (int)PT_MH);
        const int propose = s == 0 ? (int)PT_BD : (int)PT_MH;
        const unsigned int target = w.epoch + (unsigned int)s + 1u;
        if (row == 0) {
            if (threadIdx.x < 64) {
                pt_step_call(dp, chain, decide, 0, propose, s >> 1, s & 1);
                __threadfence();  // the row is out before the flag; and this workgroup (one L1) reads it afresh, too
            }
            __builtin_amdgcn_s_dcache_inv();
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_store((unsigned int *)flag, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (threadIdx.x == 0)
                go_on = pt_wait_reached(flag, target, abort_word, d_status) ? 1 : 0;
            __syncthreads();
            if (!go_on)
                return;
            __builtin_amdgcn_s_dcache_inv();
        }
        if (s == 1) {
            // k of this chain for the rest of the sweep (written by the step that has just run, visible after the hand-shake)
            const int kc = __builtin_amdgcn_readfirstlane(dp->k[chain]);
            S_c = 2 * kc < S_c ? 2 * kc : S_c;
        }
        fused_body<double, false, false, true, false, TF, INCR, true>(a, chain, row, gx, reinterpret_cast<double *>(lds_raw));
        __syncthreads();  // the workgroup's residual sum is written (thread 0), LDS is free again
        if (threadIdx.x == 0) {
            __threadfence();
            __hip_atomic_fetch_add((unsigned int *)count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (row == 0)
                go_on = pt_wait_reached(count, count_base + (unsigned int)(s + 1) * (unsigned int)gx, abort_word, d_status) ? 1 : 0;
        }
        if (row == 0) {
            __syncthreads();
            if (!go_on)
                return;
            __builtin_amdgcn_s_dcache_inv();
        }
    }
    // the last decision and the sigma move (EXPLORE_MH :521-575)
    if (row == 0 && threadIdx.x < 64)
        pt_step_call(dp, chain, (int)PT_MH, 1, (int)PT_NONE, 0, 0);
}

// Diagnostic: bitwise comparison of the guarded sqrt/divide with the plain operators.
// out[0] += sqrt mismatches over w[], out[1] += divide mismatches over (num[], den[]),
// out[2] += mismatches of the shared-reciprocal layer term (w[], numerators num[] and den[]).
// EX: also out[3] += term mismatches counted WITHOUT regard to the term's own guard (what the kernels would return if the
// all-ones-divisor guard were compiled out) and out[4] += operands that guard rejected.
template <bool EX>
__global__ void fastmath_check_kernel(const double *__restrict__ w, const double *__restrict__ num,
                                      const double *__restrict__ den, int64_t n, unsigned long long *out)
{
    unsigned long long bad_s = 0, bad_d = 0, bad_t = 0, bad_raw = 0, rejected = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double a = sqrt_guarded(w[i]);
        const double b = sqrt(w[i]);
        bad_s += (__double_as_longlong(a) != __double_as_longlong(b));
        bool dok = true;
        const double c = div_guarded(num[i], den[i], dok);
        const double d = div_ieee(num[i], den[i]);
        bad_d += dok && (__double_as_longlong(c) != __double_as_longlong(d));
        // the shared-reciprocal layer term against sqrt and two plain divides (num[i] plays H*V*x, den[i] H*V)
        const double wa[1] = {w[i]}, na[1] = {num[i]}, da[1] = {den[i]};
        double tf[1], tp[1];
        const bool sound = term_guarded_n<1>(wa, na, da, tf, tp);  // false: the kernels would take the plain operators
        const double tf_ref = div_ieee(num[i], b);
        const double tp_ref = div_ieee(den[i], (b * b) * b);
        const unsigned long long mis = (__double_as_longlong(tf[0]) != __double_as_longlong(tf_ref)) +
                                       (__double_as_longlong(tp[0]) != __double_as_longlong(tp_ref));
        if (sound)
            bad_t += mis;
        if (EX && mis) {  // the first 58 offending operand indices (bit 63: the guard had accepted it), for the report
            const unsigned long long slot = atomicAdd(&out[5], 1ull);
            if (slot < 58)
                out[6 + slot] = (unsigned long long)i | (sound ? 1ull << 63 : 0ull);
        }
        bad_raw += mis;
        rejected += !sound;
    }
    if (bad_s)
        atomicAdd(&out[0], bad_s);
    if (bad_d)
        atomicAdd(&out[1], bad_d);
    if (bad_t)
        atomicAdd(&out[2], bad_t);
    if (EX) {
        if (bad_raw)
            atomicAdd(&out[3], bad_raw);
        if (rejected)
            atomicAdd(&out[4], rejected);
    }
}

// single precision: the layer term against sqrtf and two plain divides; out[0] += mismatches of num/sqrt(w),
// out[1] += mismatches of den/sqrt(w)^3
__global__ void fastmath_check_kernel_f32(const float *__restrict__ w, const float *__restrict__ num,
                                          const float *__restrict__ den, int64_t n, unsigned long long *out)
{
    unsigned long long bad_f = 0, bad_p = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float wa[1] = {w[i]}, na[1] = {num[i]}, da[1] = {den[i]};
        float tf[1], tp[1];
        const bool sound = term_guarded_n<1>(wa, na, da, tf, tp);
        const float s = sqrtf(w[i]);
        const float tf_ref = div_ieee(num[i], s);
        const float tp_ref = div_ieee(den[i], (s * s) * s);
        if (sound) {
            bad_f += (__float_as_uint(tf[0]) != __float_as_uint(tf_ref));
            bad_p += (__float_as_uint(tp[0]) != __float_as_uint(tp_ref));
        }
    }
    if (bad_f)
        atomicAdd(&out[0], bad_f);
    if (bad_p)
        atomicAdd(&out[1], bad_p);
}

// Diagnostic: the VALUES the kernels' arithmetic building blocks return, for comparison with an IEEE host (numpy):
//   o_div = div_ieee(num, den), o_divg = div_guarded(num, den) (NaN where its guard rejects), o_sqrt = sqrt_guarded(w),
//   o_tf / o_tp = the shared-reciprocal layer term num / sqrt(w), den / sqrt(w)^3 (NaN where the term's guard rejects).
template <typename T>
__global__ void fastmath_values_kernel(const T *__restrict__ w, const T *__restrict__ num, const T *__restrict__ den,
                                       int64_t n, T *o_div, T *o_divg, T *o_sqrt, T *o_tf, T *o_tp)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (o_div)
            o_div[i] = div_ieee(num[i], den[i]);
        if constexpr (std::is_same<T, double>::value) {
            if (o_divg) {
                bool ok = true;
                const double v = div_guarded(num[i], den[i], ok);
                o_divg[i] = ok ? v : __builtin_nan("");
            }
            if (o_sqrt)
                o_sqrt[i] = sqrt_guarded(w[i]);
        }
        if (o_tf || o_tp) {
            const T wa[1] = {w[i]}, na[1] = {num[i]}, da[1] = {den[i]};
            T tf[1], tp[1];
            const bool sound = term_guarded_n<1>(wa, na, da, tf, tp);
            if (o_tf)
                o_tf[i] = sound ? tf[0] : T(__builtin_nan(""));
            if (o_tp)
                o_tp[i] = sound ? tp[0] : T(__builtin_nan(""));
        }
    }
}

}  // namespace rtmi

extern "C" hipError_t rtmi_launch_fastmath_values(const double *d_w, const double *d_num, const double *d_den, int64_t n,
                                                  double *o_div, double *o_divg, double *o_sqrt, double *o_tf,
                                                  double *o_tp, hipStream_t stream)
{
    hipLaunchKernelGGL(rtmi::fastmath_values_kernel<double>, dim3(2048), dim3(256), 0, stream, d_w, d_num, d_den, n, o_div,
                       o_divg, o_sqrt, o_tf, o_tp);
    return hipGetLastError();
}

extern "C" hipError_t rtmi_launch_fastmath_values_f32(const float *d_w, const float *d_num, const float *d_den, int64_t n,
                                                      float *o_div, float *o_tf, float *o_tp, hipStream_t stream)
{
    hipLaunchKernelGGL(rtmi::fastmath_values_kernel<float>, dim3(2048), dim3(256), 0, stream, d_w, d_num, d_den, n, o_div,
                       (float *)nullptr, (float *)nullptr, o_tf, o_tp);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Launchers (host).  No allocation, no synchronisation: graph-capturable.
// ---------------------------------------------------------------------------------------------
// The name of the kernel the calling thread's last launcher call dispatched, spelled as rocprofv3 prints it: what bench.py
// reports as `roofline.kernel`, so that the line can be matched against a kernel trace by name (rtmi_last_kernel_name).
static thread_local char g_last_kernel[192] = "";
static void note_kernel(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
static void note_kernel(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_kernel, sizeof g_last_kernel, fmt, ap);
    va_end(ap);
}
static const char *tf(bool b) { return b ? "true" : "false"; }
extern "C" const char *rtmi_last_kernel_name(void) { return g_last_kernel; }

extern "C" int rtmi_launch_grid_x(int64_t nsrc, int nmodels, int num_cus)
{
    const int64_t ntiles = (nsrc + RTMI_BLOCK - 1) / RTMI_BLOCK;
    // enough workgroups to fill every CU several times over; beyond that the tiles are grid-strided
    int64_t cap = (int64_t)num_cus * RTMI_BLOCKS_PER_CU;
    if (nmodels > 1) {
        cap = cap / nmodels;
        if (cap < 8)
            cap = 8;
    }
    int64_t g = ntiles < cap ? ntiles : cap;
    if (g < 1)
        g = 1;
    return (int)g;
}

template <typename T>
static hipError_t launch_trace_t(const rtmi::TraceArgsT<T> &args, int nmodels, int max_nlayers, int grid_x,
                                 hipStream_t stream)
{
    using namespace rtmi;
    TraceArgsT<T> a = args;
    a.max_nlayers = max_nlayers;  // what the LDS table below is sized for: the kernel checks d_nlayers[m] against it
    a.skiptab = max_nlayers <= RTMI_SKIP_MAX_NL ? 1 : 0;
    const size_t lds_bytes = RTMI_TABLE_ELEMS(max_nlayers, false, a.skiptab) * sizeof(T);
    const dim3 grid(grid_x, nmodels, 1), block(RTMI_BLOCK, 1, 1);
    const bool want_p = a.d_p_final != nullptr;
    const bool want_t = a.d_timeP != nullptr;
    const bool want_ss = a.d_obs != nullptr;
    {
        const char *tn = std::is_same<T, double>::value ? "double" : "float";
        note_kernel("void rtmi::trace_kernel<%s, %s, %s, %s>(rtmi::TraceArgsT<%s>)", tn, tf(want_p && !want_ss), tf(want_t || !want_ss), tf(want_ss), tn);
    }
    if constexpr (std::is_same<T, double>::value) {
        // small single-model launches without a residual sum: a group of lanes per source (trace_small_kernel)
        if (!want_ss && want_t && nmodels == 1 && !a.d_nlayers && !a.lane_per_source && rtmi_small_kernel_takes(a.nsrc, a.nlayers)) {
            const int G = rtmi_small_group(a.nlayers);
            const int spb = RTMI_BLOCK / G;
            const int64_t nb = (a.nsrc + spb - 1) / spb;
            const dim3 sgrid((unsigned)(nb < 4096 ? nb : 4096), 1, 1);
            const size_t slds = ((RTMI_TABLE_ELEMS(max_nlayers, false, a.skiptab) + 1) & ~(size_t)1) * sizeof(double) + (size_t)RTMI_BLOCK * sizeof(double2);
            note_kernel("void rtmi::trace_small_kernel<%d, %s>(rtmi::TraceArgsT<double>)", G, tf(want_p));
#define RTMI_SMALL_LAUNCH(GG)                                                                                          \
    if (want_p)                                                                                                        \
        hipLaunchKernelGGL((trace_small_kernel<GG, true>), sgrid, block, slds, stream, a);                              \
    else                                                                                                               \
        hipLaunchKernelGGL((trace_small_kernel<GG, false>), sgrid, block, slds, stream, a);
            if (G == 8) {
                RTMI_SMALL_LAUNCH(8)
            } else if (G == 16) {
                RTMI_SMALL_LAUNCH(16)
            } else if (G == 32) {
                RTMI_SMALL_LAUNCH(32)
            } else {
                RTMI_SMALL_LAUNCH(64)
            }
#undef RTMI_SMALL_LAUNCH
            return hipGetLastError();
        }
    }
    if (want_ss && want_t)
        hipLaunchKernelGGL((trace_kernel<T, false, true, true>), grid, block, lds_bytes, stream, a);
    else if (want_ss)
        hipLaunchKernelGGL((trace_kernel<T, false, false, true>), grid, block, lds_bytes, stream, a);
    else if (want_p)
        hipLaunchKernelGGL((trace_kernel<T, true, true, false>), grid, block, lds_bytes, stream, a);
    else
        hipLaunchKernelGGL((trace_kernel<T, false, true, false>), grid, block, lds_bytes, stream, a);
    return hipGetLastError();
}

extern "C" hipError_t rtmi_launch_trace(const rtmi::TraceArgs *args, int nmodels, int max_nlayers, int grid_x,
                                        hipStream_t stream)
{
    return launch_trace_t<double>(*args, nmodels, max_nlayers, grid_x, stream);
}

extern "C" hipError_t rtmi_launch_trace_f32(const rtmi::TraceArgsT<float> *args, int nmodels, int max_nlayers,
                                            int grid_x, hipStream_t stream)
{
    return launch_trace_t<float>(*args, nmodels, max_nlayers, grid_x, stream);
}

extern "C" hipError_t rtmi_launch_loglike(const double *d_partial, int nparts, const double *d_sd, int64_t ndat,
                                          int nmodels, double *d_logL, double *d_ss, hipStream_t stream)
{
    const int block = 64;
    const int grid = (nmodels + block - 1) / block;
    hipLaunchKernelGGL(rtmi::loglike_kernel, dim3(grid), dim3(block), 0, stream, d_partial, nparts, d_sd, ndat,
                       nmodels, d_logL, d_ss);
    return hipGetLastError();
}

// Geometry of the fused kernel: one workgroup of 64 x `waves_per_simd` x 4 threads per CU (two for more than 4
// waves per SIMD), holding as many of its records in LDS as its share of the CU's 160 KiB allows.  With several
// models the CUs are divided among them (grid.y = model).  Returns workgroups PER MODEL; *threads per workgroup,
// *slots = records per workgroup and round (multiple of 64), *lds_bytes per workgroup (table sized for
// `max_nlayers`).
extern "C" int rtmi_fused_geometry(int64_t nsrc, int max_nlayers, int nmodels, int num_cus, int waves_per_simd,
                                   int elem_bytes, int *threads, int *slots, size_t *lds_bytes)
{
    const size_t slot_bytes = (size_t)3 * elem_bytes + 10;  // RTMI_SLOT_BYTES(T)
    const int64_t nchunks = (nsrc + 63) / 64;
    if (waves_per_simd < 1)
        waves_per_simd = 1;
    if (waves_per_simd > 8)
        waves_per_simd = 8;
    if (nmodels < 1)
        nmodels = 1;
    const int wg_per_cu = waves_per_simd > 4 ? 2 : 1;
    int waves = 4 * ((waves_per_simd + wg_per_cu - 1) / wg_per_cu);  // waves per workgroup, <= 16
    int64_t nblocks = ((int64_t)num_cus * wg_per_cu) / nmodels;        // per model
    if (nblocks > nchunks)
        nblocks = nchunks;
    if (nblocks < 1)
        nblocks = 1;
    const int64_t chunks_per_block = (nchunks + nblocks - 1) / nblocks;
    if (waves > chunks_per_block)
        waves = (int)chunks_per_block;  // small inputs: no idle waves
    const size_t table = (RTMI_TABLE_ELEMS(max_nlayers, RTMI_TABLE_FORM(max_nlayers), max_nlayers <= RTMI_SKIP_MAX_NL) * elem_bytes + 15) / 16 * 16;
    const size_t fixed = table + (16 * 64 + 4) * sizeof(uint32_t) + 16 +
                         (size_t)RTMI_POOL * (4 * elem_bytes + 3 * sizeof(int)) + 4 * sizeof(int) +
                         (128 + 128) * sizeof(uint32_t);
    size_t share = (size_t)160 * 1024 / wg_per_cu;
    int64_t max_slots = share > fixed + 64 * slot_bytes ? (int64_t)((share - fixed) / slot_bytes / 64 * 64) : 64;
    if (max_slots > 8128)
        max_slots = 8128;  // 13-bit local index
    int64_t want = chunks_per_block * 64;
    if (want > max_slots)
        want = max_slots;
    *threads = waves * 64;
    *slots = (int)want;
    *lds_bytes = fixed + (size_t)want * slot_bytes;
    return (int)nblocks;
}

template <typename T, bool P, bool WT, bool S, bool DIAG = false>
static hipError_t set_fused_lds_limit()
{
    // dynamic LDS above 64 KiB needs the opt-in (the three forms of the layer table)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rtmi::fused_kernel<T, P, WT, S, DIAG, 0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rtmi::fused_kernel<T, P, WT, S, DIAG, 1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rtmi::fused_kernel<T, P, WT, S, DIAG, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
}

// args: d_vels/d_depths/(d_nlayers|nlayers)/ld_model, sources, outputs, refill_min, tail_group, d_stamps filled by
// the caller; slots is set here.  *grid_x_out (may be NULL) receives the workgroups per model (= partial sums per
// model).
template <typename T>
static hipError_t prepare_fused_t()
{
    static bool attr_set = false;  // one flag per element type
    if (attr_set)
        return hipSuccess;
    hipError_t e = set_fused_lds_limit<T, false, true, false>();
    if (e == hipSuccess)
        e = set_fused_lds_limit<T, false, true, false, true>();
    if (e == hipSuccess)
        e = set_fused_lds_limit<T, true, true, false>();
    if (e == hipSuccess)
        e = set_fused_lds_limit<T, false, true, true>();
    if (e == hipSuccess)
        e = set_fused_lds_limit<T, false, false, true>();
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rtmi::fused_kernel<T, false, false, true, false, 0, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rtmi::fused_kernel<T, false, false, true, false, 1, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rtmi::fused_kernel<T, false, false, true, false, 2, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
        attr_set = true;
    return e;
}

// One-time function attributes (not stream operations): done at context creation so that the launchers stay
// graph-capturable from the first call.
extern "C" hipError_t rtmi_prepare_kernels()
{
    hipError_t e = prepare_fused_t<double>();
    if (e == hipSuccess)
        e = prepare_fused_t<float>();
    return e;
}

template <typename T>
static hipError_t launch_fused_t(const rtmi::FusedArgsT<T> &args, int nmodels, int max_nlayers, int num_cus,
                                 int waves_per_simd, int *grid_x_out, hipStream_t stream)
{
    using namespace rtmi;
    {
        const hipError_t e = prepare_fused_t<T>();  // no-op after rtmi_init / the first call
        if (e != hipSuccess)
            return e;
    }
    FusedArgsT<T> a = args;
    a.max_nlayers = max_nlayers;  // what the LDS table is sized for: the kernel checks d_nlayers[m] against it
    a.rows = RTMI_TABLE_FORM(max_nlayers);                // table form (rtmi_device.hpp): 1 = row tables, 1 / 2 = constant stride
    a.skiptab = max_nlayers <= RTMI_SKIP_MAX_NL ? 1 : 0;  // launch-wide
    a.refill_min = a.refill_min < 1 ? 1 : (a.refill_min > 64 ? 64 : a.refill_min);
    size_t lds_bytes = 0;
    int threads = 0;
    const int nblocks = rtmi_fused_geometry(a.nsrc, max_nlayers, nmodels, num_cus, waves_per_simd, (int)sizeof(T),
                                            &threads, &a.slots, &lds_bytes);
    if (grid_x_out)
        *grid_x_out = nblocks;
    if (lds_bytes > 160 * 1024)
        return hipErrorInvalidValue;
    const dim3 grid(nblocks, nmodels, 1), block(threads, 1, 1);
    const bool want_p = a.d_p_final != nullptr;
    const bool want_t = a.d_timeP != nullptr;
    const bool want_ss = a.d_obs != nullptr;
    const char *tname = std::is_same<T, double>::value ? "double" : "float";
#define RTMI_LAUNCH_FUSED(P, WT, S, DG)                                                                          \
    do {                                                                                                        \
        note_kernel("void rtmi::fused_kernel<%s, %s, %s, %s, %s, %d, false>(rtmi::FusedArgsT<%s>)", tname, tf(P), tf(WT), tf(S), \
                    tf(DG), a.rows, tname);                                                                     \
        if (a.rows == 1)                                                                                        \
            hipLaunchKernelGGL((fused_kernel<T, P, WT, S, DG, 1>), grid, block, lds_bytes, stream, a);          \
        else if (a.rows == 2)                                                                                   \
            hipLaunchKernelGGL((fused_kernel<T, P, WT, S, DG, 2>), grid, block, lds_bytes, stream, a);          \
        else                                                                                                    \
            hipLaunchKernelGGL((fused_kernel<T, P, WT, S, DG, 0>), grid, block, lds_bytes, stream, a);          \
    } while (0)
    if (want_ss && want_t)
        RTMI_LAUNCH_FUSED(false, true, true, false);
    else if (want_ss && a.d_partial_x && std::is_same<T, double>::value) {
        // the tempering sampler's form of the likelihood (exact sums, optionally incremental: PtInc): its own instantiation
        note_kernel("void rtmi::fused_kernel<%s, false, false, true, false, %d, true>(rtmi::FusedArgsT<%s>)", tname, a.rows, tname);
        if (a.rows == 1)
            hipLaunchKernelGGL((fused_kernel<T, false, false, true, false, 1, true>), grid, block, lds_bytes, stream, a);
        else if (a.rows == 2)
            hipLaunchKernelGGL((fused_kernel<T, false, false, true, false, 2, true>), grid, block, lds_bytes, stream, a);
        else
            hipLaunchKernelGGL((fused_kernel<T, false, false, true, false, 0, true>), grid, block, lds_bytes, stream, a);
    } else if (want_ss)
        RTMI_LAUNCH_FUSED(false, false, true, false);
    else if (want_p)
        RTMI_LAUNCH_FUSED(true, true, false, false);
    else if (a.d_stamps)
        RTMI_LAUNCH_FUSED(false, true, false, true);
    else
        RTMI_LAUNCH_FUSED(false, true, false, false);
#undef RTMI_LAUNCH_FUSED
    return hipGetLastError();
}

extern "C" hipError_t rtmi_launch_fused(const rtmi::FusedArgs *args, int nmodels, int max_nlayers, int num_cus,
                                        int waves_per_simd, int *grid_x_out, hipStream_t stream)
{
    return launch_fused_t<double>(*args, nmodels, max_nlayers, num_cus, waves_per_simd, grid_x_out, stream);
}

extern "C" hipError_t rtmi_launch_pt_sweep(const rtmi::FusedArgs *args, const void *ptdev, size_t ptdev_bytes,
                                           const rtmi::PtSweepArgs *w, int nchains, int max_nlayers, int num_cus,
                                           int waves_per_simd, hipStream_t stream)
{
    using namespace rtmi;
    using rtmi_pt_dev::PtDev;
    if (!args || !ptdev || !w || ptdev_bytes != sizeof(PtDev) || !args->d_obs || !args->d_nlayers || args->d_timeP)  // ptdev_bytes: layout check
        return hipErrorInvalidValue;
    {
        const hipError_t e = prepare_fused_t<double>();
        if (e != hipSuccess)
            return e;
    }
    FusedArgs a = *args;
    const PtDev *dp = static_cast<const PtDev *>(ptdev);  // a DEVICE copy of the handle's view
    PtSweepArgs ws = *w;
    a.max_nlayers = max_nlayers;
    a.rows = RTMI_TABLE_FORM(max_nlayers);
    a.skiptab = max_nlayers <= RTMI_SKIP_MAX_NL ? 1 : 0;
    a.refill_min = a.refill_min < 1 ? 1 : (a.refill_min > 64 ? 64 : a.refill_min);
    size_t lds_bytes = 0;
    int threads = 0;
    const int gx = rtmi_fused_geometry(a.nsrc, max_nlayers, nchains, num_cus, waves_per_simd, (int)sizeof(double), &threads,
                                       &a.slots, &lds_bytes);
    static const bool verbose = getenv("RTMI_PT_VERBOSE") != nullptr;  // read once
    ws.lds_ctl = (int)((lds_bytes + 15) / 16 * 16);
    lds_bytes = (size_t)ws.lds_ctl + 16;
    if (gx != ws.gx || lds_bytes > 160 * 1024 || threads < 64 || ws.nchains != nchains) {
        if (verbose)
            fprintf(stderr, "rtmi_launch_pt_sweep: geometry gx %d (sweep %d), lds %zu, threads %d\n", gx, ws.gx, lds_bytes, threads);
        return hipErrorInvalidValue;
    }
    // (a chain's share of the sources may take several rounds of its workgroups' LDS: fused_body loops over them)
    ws.xcd_map = (nchains % 8 == 0) ? 1 : 0;  // measured: 1.18 ms per sweep against 1.21 with a chain spread over the XCDs
    // table form 2 (constant stride without row tables, models of 21 ... 32 interfaces) pays in the likelihood kernel
    // (config 4's batch -2.8 %) but not with the body inlined into this kernel's slot loop (+2.7 %, measured): form 0 here
    // (the LDS the geometry reserved for the larger table is simply not all used)
    if (a.rows == 2)
        a.rows = 0;
    const void *fns[2][2] = {{reinterpret_cast<const void *>(&pt_sweep_kernel<0, false>), reinterpret_cast<const void *>(&pt_sweep_kernel<1, false>)},
                             {reinterpret_cast<const void *>(&pt_sweep_kernel<0, true>), reinterpret_cast<const void *>(&pt_sweep_kernel<1, true>)}};
    if (ws.nspec > 0)
        return hipErrorInvalidValue;  // (the speculative step runs over the task queue: rtmi_launch_pt_queue)
    const void *fn = fns[a.d_partial_x ? 1 : 0][a.rows];
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess)
        return e;
    void *kargs[3] = {&a, &dp, &ws};
    e = hipLaunchCooperativeKernel(fn, dim3((unsigned)(gx * nchains)), dim3((unsigned)threads), kargs, lds_bytes, stream);
    if (e != hipSuccess && verbose) {
        int per_cu = -1, coop = -1, dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds_bytes);
        fprintf(stderr, "rtmi_launch_pt_sweep: cooperative launch of %d x %d threads, %zu B LDS failed (%s); device cooperative %d, "
                        "workgroups per CU %d\n", gx * nchains, threads, lds_bytes, hipGetErrorString(e), coop, per_cu);
    }
    return e;
}

#ifndef RTMI_PT_TEAMS_DEFAULT
#define RTMI_PT_TEAMS_DEFAULT 2  // with four rows per chain (see rtmi_launch_pt_prop); profiles/r04r/pt_teams.txt
#endif
extern "C" hipError_t rtmi_launch_pt_prop(const rtmi::FusedArgs *args, const void *ptdev, size_t ptdev_bytes, const rtmi::PtSweepArgs *d_w,
                                          int nspec, unsigned int epoch, int parity, int nslots_k, int nchains, int max_nlayers,
                                          int num_cus, int waves_per_simd, hipStream_t stream)
{
    using namespace rtmi;
    using rtmi_pt_dev::PtDev;
    if (!args || !ptdev || !d_w || ptdev_bytes != sizeof(PtDev) || !args->d_obs || !args->d_nlayers || args->d_timeP || !args->d_partial_x ||
        nspec < 1 || nspec > PT_SPEC_MAX || nchains < 1 || nchains > num_cus)
        return hipErrorInvalidValue;
    {
        const hipError_t e = prepare_fused_t<double>();
        if (e != hipSuccess)
            return e;
    }
    FusedArgs a = *args;
    const PtDev *dp = static_cast<const PtDev *>(ptdev);  // a DEVICE copy of the handle's view
    a.max_nlayers = max_nlayers;
    a.rows = RTMI_TABLE_FORM(max_nlayers);
    a.skiptab = max_nlayers <= RTMI_SKIP_MAX_NL ? 1 : 0;
    a.refill_min = a.refill_min < 1 ? 1 : (a.refill_min > 64 ? 64 : a.refill_min);
    // one workgroup per CU with as large a record area as a CU holds: a chain with few workgroups takes a row in several rounds
    size_t lds_bytes = 0;
    int threads = 0;
    (void)rtmi_fused_geometry(a.nsrc, max_nlayers, num_cus, num_cus, waves_per_simd, (int)sizeof(double), &threads, &a.slots, &lds_bytes);
    // Two workgroups of half the size per CU (RTMI_PT_TEAMS=2; the default with four rows per chain).  A chain's workgroups idle
    // while its leading workgroup takes the step (a quarter of a pass) and while the slowest of them finishes its share; with
    // workgroups of different chains on a CU the other one's likelihood share runs in that time.  Same waves per SIMD, half the
    // record area each.  Measured on config 3's shape (profiles/r04r, r04s/pt_teams.txt): three rows 0.776 -> 0.774 ms per sweep,
    // four rows 0.787 -> 0.742; four workgroups of 256 lanes per CU are refused by the cooperative launch.
    static const int teams_env = getenv("RTMI_PT_TEAMS") ? atoi(getenv("RTMI_PT_TEAMS")) : 0;
    int teams = teams_env > 0 ? teams_env : (nspec >= 4 ? RTMI_PT_TEAMS_DEFAULT : 1);
    teams = teams >= 2 && threads >= 512 ? 2 : 1;
    int nwg = num_cus;
    if (teams > 1) {
        const size_t slot_bytes = (size_t)3 * sizeof(double) + 10;
        const size_t fixed = lds_bytes - (size_t)a.slots * slot_bytes;
        const size_t share = (size_t)160 * 1024 / teams - 4 * PROP_WORDS - 64 - 16 - 512;  // (512: the kernel's static LDS, generously)
        if (share > fixed + 64 * slot_bytes) {
            const int max_slots = (int)((share - fixed) / slot_bytes / 64 * 64);
            threads /= teams;
            if (a.slots > max_slots)
                a.slots = max_slots;
            lds_bytes = fixed + (size_t)a.slots * slot_bytes;
            nwg = teams * num_cus;
        }
    }
    int lds_ctl = (int)((lds_bytes + 15) / 16 * 16);
    lds_bytes = (size_t)lds_ctl + 4 * PROP_WORDS + 64;  // + the kernel's control words and its diagnostic sums
    if (lds_bytes > 160 * 1024 || threads < 64)
        return hipErrorInvalidValue;
    if (a.rows == 2)
        a.rows = 0;  // (as in rtmi_launch_pt_sweep: form 0 with the body inlined into a loop)
    const void *fn = a.rows == 1 ? reinterpret_cast<const void *>(&pt_sweep_prop_kernel<1>) : reinterpret_cast<const void *>(&pt_sweep_prop_kernel<0>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess)
        return e;
    int max_pass = 2 * nslots_k + 2;  // a pass retires at least one slot of the chain
    void *kargs[7] = {&a, &dp, &d_w, &epoch, &parity, &max_pass, &lds_ctl};
    return hipLaunchCooperativeKernel(fn, dim3((unsigned)nwg), dim3((unsigned)threads), kargs, lds_bytes, stream);
}

extern "C" hipError_t rtmi_launch_fused_f32(const rtmi::FusedArgsT<float> *args, int nmodels, int max_nlayers,
                                            int num_cus, int waves_per_simd, int *grid_x_out, hipStream_t stream)
{
    return launch_fused_t<float>(*args, nmodels, max_nlayers, num_cus, waves_per_simd, grid_x_out, stream);
}

extern "C" hipError_t rtmi_launch_fastmath_check_f32(const float *d_w, const float *d_num, const float *d_den,
                                                     int64_t n, unsigned long long *d_out, hipStream_t stream)
{
    hipLaunchKernelGGL(rtmi::fastmath_check_kernel_f32, dim3(2048), dim3(256), 0, stream, d_w, d_num, d_den, n, d_out);
    return hipGetLastError();
}

extern "C" hipError_t rtmi_launch_fastmath_check(const double *d_w, const double *d_num, const double *d_den,
                                                 int64_t n, unsigned long long *d_out, int extended, hipStream_t stream)
{
    if (extended)
        hipLaunchKernelGGL(rtmi::fastmath_check_kernel<true>, dim3(2048), dim3(256), 0, stream, d_w, d_num, d_den, n, d_out);
    else
        hipLaunchKernelGGL(rtmi::fastmath_check_kernel<false>, dim3(2048), dim3(256), 0, stream, d_w, d_num, d_den, n, d_out);
    return hipGetLastError();
}
