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
// Two paths (DESIGN.md has the measurements):
//
//  (1) trace_kernel -- one launch, one lane per source in input order.  The three solver branches
//      of GetPTime are folded into one per-lane state machine around a single wave-converged
//      "evaluate f and f' at x" loop (one sqrt + two divides per layer; f and f' share the sqrt,
//      solvebst's duplicate costFunc call is evaluated once).  Used for small NSrc and for the
//      many-models batch.  Its cost is set by the slowest lane of each wave: deepest source x most
//      evaluations.
//
//  (2) the binned pipeline for large NSrc -- same arithmetic per source, scheduled so that lanes
//      stay busy:
//        bin_count_kernel    nl per source + histogram of nl
//        bin_scatter_kernel  counting sort by DESCENDING nl (so a wave's layer loop has one trip
//                            count), p0 per source, straight top-layer rays finished here
//        solve_kernel        persistent waves; a lane that finishes its source immediately takes
//                            the next one of its wave's queue (lane refill), so the evaluation loop
//                            runs with nearly all 64 lanes active instead of waiting for the
//                            slowest source
//        time_kernel         t from the final ray parameter, in input order (coalesced)
//      HBM traffic rises from 24 to ~150 B per source, still a few percent of the HBM roof; the
//      kernel is fp64-VALU bound and this trade buys ~3x lane utilisation.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtmi_device.hpp"
#include "rtmi_kernels.h"

namespace rtmi {

// =============================================================================================
// (1) single-launch path
// =============================================================================================

// One source per lane, all lanes of the wave together (wave-uniform loops inside);
// `valid` = false lanes carry a dummy source and their result is discarded.
__device__ __forceinline__ double trace_one(const LayerTable &t, int nlayers, double R, double z, bool valid,
                                            double &p_out)
{
    const int nl = which_layer(t, nlayers, z);
    const bool straight = (nl == 1);  // source in the top layer (:94-97)
    const bool active = valid && !straight;
    const int il = active ? nl - 1 : 1;  // 0-based layer of the source; dummy lanes use layer 1
    Lane s;
    lane_setup(s, t, il, R, z);
    const double p0 = active ? initial_p0(t, il, R, z) : 0.0;
    lane_start(s, p0);
    if (!active || p0 != p0)
        s.st = ST_DONE;
    const int nfull_wave = wave_max_i32(active ? s.nfull : 0);

    while (__any(s.st != ST_DONE)) {
        double f, fp;
        eval_f_fp(t, s.xe, s.nfull, nfull_wave, s.hvl, s.vvl, s.R, f, fp);
        lane_advance(s, f, fp);
    }

    double tt = travel_time(t, s.x, s.nfull, nfull_wave, il, z);
    if (!s.conv)
        tt = -999.0;
    p_out = s.x;
    if (straight) {
        tt = sqrt(z * z + R * R) / t.V[0];
        p_out = __builtin_nan("");
    }
    return tt;
}

// grid.x: source tiles (grid-stride), grid.y: models.
template <bool WANT_P, bool WANT_T, bool WANT_SS>
__global__ __launch_bounds__(RTMI_BLOCK) void trace_kernel(TraceArgs a)
{
    extern __shared__ double lds[];
    const int m = blockIdx.y;
    const int nlayers = a.d_nlayers ? a.d_nlayers[m] : a.nlayers;
    const LayerTable t = carve_table(lds, nlayers);
    stage_model(t, a.d_vels + (int64_t)m * a.ld_model, a.d_depths + (int64_t)m * a.ld_model, nlayers);

    double ss = 0.0;  // this thread's sum of squared residuals (fused logL)
    double *out = a.d_timeP + (int64_t)m * a.nsrc;
    const int64_t ntiles = (a.nsrc + RTMI_BLOCK - 1) / RTMI_BLOCK;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t k = tile * RTMI_BLOCK + threadIdx.x;
        const bool valid = k < a.nsrc;
        const double R = valid ? a.d_src_offset[k] : 1.0;
        const double z = valid ? a.d_src_depth[k] : 1.0;
        double p;
        const double tt = trace_one(t, nlayers, R, z, valid, p);
        if (valid) {
            if (WANT_T)
                out[k] = tt;
            if (WANT_P)
                a.d_p_final[k] = p;
            if (WANT_SS) {
                const double r = a.d_obs[k] - tt;
                ss = ss + r * r;
            }
        }
    }

    if (WANT_SS) {
        // block reduction in a fixed order: lanes by xor-butterfly, then waves 0..3
        __shared__ double wsum[RTMI_BLOCK / 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            ss = ss + __shfl_xor(ss, off, 64);
        if ((threadIdx.x & 63) == 0)
            wsum[threadIdx.x >> 6] = ss;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = wsum[0];
            for (int w = 1; w < RTMI_BLOCK / 64; ++w)
                s = s + wsum[w];
            a.d_partial[(int64_t)m * gridDim.x + blockIdx.x] = s;
        }
    }
}

// logL[m] = -(N/2) ln(2 pi) - ( SS/(2 sd^2) + N ln sd )     loglhood.f90:194-196
__global__ void loglike_kernel(const double *__restrict__ partial, int nparts, const double *__restrict__ sd,
                               int64_t ndat, int nmodels, double *__restrict__ logL)
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
}

// =============================================================================================
// (2) binned pipeline
// =============================================================================================

// Bin b = nl (1 .. nlayers+1).  hist/cursor have nlayers+2 entries, zeroed before the launch.
__global__ __launch_bounds__(RTMI_BLOCK) void bin_count_kernel(BinArgs a)
{
    extern __shared__ double lds[];
    const LayerTable t = carve_table(lds, a.nlayers);
    uint32_t *lh = reinterpret_cast<uint32_t *>(lds + RTMI_TABLE_ARRAYS * (a.nlayers + 1));
    const int nb = a.nlayers + 2;
    for (int i = threadIdx.x; i < nb; i += blockDim.x)
        lh[i] = 0;
    stage_model(t, a.d_vels, a.d_depths, a.nlayers);  // ends with a barrier

    const int64_t ntiles = (a.nsrc + RTMI_BLOCK - 1) / RTMI_BLOCK;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t k = tile * RTMI_BLOCK + threadIdx.x;
        const bool valid = k < a.nsrc;
        const double z = valid ? a.d_src_depth[k] : 0.0;
        const int nl = which_layer(t, a.nlayers, z);
        if (valid) {
            a.key[k] = nl;
            atomicAdd(&lh[nl], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x)
        if (lh[i])
            atomicAdd(&a.hist[i], lh[i]);
}

// Counting sort by descending nl + per-source start values.  One tile of RTMI_BLOCK sources per
// workgroup iteration; positions inside a bin are handed out by atomics (any order is fine: every
// source's arithmetic is independent of where it is processed).
__global__ __launch_bounds__(RTMI_BLOCK) void bin_scatter_kernel(BinArgs a)
{
    extern __shared__ double lds[];
    const LayerTable t = carve_table(lds, a.nlayers);
    const int nb = a.nlayers + 2;
    uint32_t *base = reinterpret_cast<uint32_t *>(lds + RTMI_TABLE_ARRAYS * (a.nlayers + 1));  // bin start
    uint32_t *lc = base + nb;                                                                   // tile count
    uint32_t *gb = lc + nb;                                                                     // tile start
    stage_model(t, a.d_vels, a.d_depths, a.nlayers);
    if (threadIdx.x == 0) {
        uint32_t acc = 0;  // descending nl: deepest bin first, straight rays (nl == 1) last
        for (int b = nb - 1; b >= 0; --b) {
            base[b] = acc;
            acc += a.hist[b];
        }
    }
    for (int i = threadIdx.x; i < nb; i += blockDim.x)
        lc[i] = 0;
    __syncthreads();

    const int64_t ntiles = (a.nsrc + RTMI_BLOCK - 1) / RTMI_BLOCK;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t k = tile * RTMI_BLOCK + threadIdx.x;
        const bool valid = k < a.nsrc;
        int nl = 1;
        uint32_t rank = 0;
        double R = 1.0, z = 1.0;
        if (valid) {
            nl = a.key[k];
            R = a.d_src_offset[k];
            z = a.d_src_depth[k];
            rank = atomicAdd(&lc[nl], 1u);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += blockDim.x) {
            const uint32_t c = lc[i];
            if (c)
                gb[i] = base[i] + atomicAdd(&a.cursor[i], c);
        }
        __syncthreads();
        if (valid) {
            if (nl == 1) {
                // straight ray in the top layer (:94-97): finished here
                a.d_timeP[k] = sqrt(z * z + R * R) / t.V[0];
                if (a.d_pfin)
                    a.d_pfin[k] = __builtin_nan("");
            } else {
                const int il = nl - 1;
                const double p0 = initial_p0(t, il, R, z);
                const uint32_t pos = gb[nl] + rank;
                a.bR[pos] = R;
                a.bz[pos] = z;
                a.bp0[pos] = p0;
                a.bidx[pos] = (uint32_t)k;
                a.bnl[pos] = nl;
                if (p0 != p0) {
                    // the reference never returns for this source (NaN/Inf data); report -999
                    a.d_timeP[k] = -999.0;
                    if (a.d_pfin)
                        a.d_pfin[k] = p0;
                    a.key[k] = 0;
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += blockDim.x)
            lc[i] = 0;
        __syncthreads();
    }
}

// Persistent waves with lane refill.  Wave w owns chunks w, w + nwaves, w + 2 nwaves, ... of 64
// consecutive sources in binned order (interleaving gives every wave the same mix of bins); a lane
// whose source is finished takes the next entry of that queue once `refill_min` lanes are idle.
__global__ __launch_bounds__(RTMI_BLOCK) void solve_kernel(BinArgs a)
{
    extern __shared__ double lds[];
    const LayerTable t = carve_table(lds, a.nlayers);
    stage_model(t, a.d_vels, a.d_depths, a.nlayers);

    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (RTMI_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (RTMI_BLOCK / 64);
    const int64_t n_solve = a.nsrc - (int64_t)a.hist[1];  // everything but the straight rays
    const int64_t nchunks = (n_solve + 63) / 64;
    const int qend = (int)((nchunks + nwaves - 1) / nwaves) * 64;  // queue length of every wave
    int qnext = 0;

    Lane s;
    s.st = ST_DONE;
    s.nfull = 0;
    s.R = s.hvl = s.vvl = s.vmi = s.x = s.xe = s.xb = s.dx = 0.0;
    s.k = 0;
    s.conv = false;
    uint32_t orig = 0;
    int nl = 0;
    int nfull_wave = 0;

    for (;;) {
        const unsigned long long idle = __ballot(s.st == ST_DONE);
        const int nidle = __popcll(idle);
        if (qnext < qend && nidle >= a.refill_min) {
            if (s.st == ST_DONE) {
                const int rank = __popcll(idle & ((1ull << lane) - 1ull));
                const int q = qnext + rank;
                if (q < qend) {
                    const int64_t pos = (wave + (int64_t)(q >> 6) * nwaves) * 64 + (q & 63);
                    if (pos < n_solve) {
                        const double p0 = a.bp0[pos];
                        if (p0 == p0) {  // NaN: already reported by bin_scatter_kernel
                            nl = a.bnl[pos];
                            orig = a.bidx[pos];
                            lane_setup(s, t, nl - 1, a.bR[pos], a.bz[pos]);
                            lane_start(s, p0);
                        }
                    }
                }
            }
            qnext += nidle;
            nfull_wave = wave_max_i32(s.st != ST_DONE ? s.nfull : 0);
        }
        if (__all(s.st == ST_DONE)) {
            if (qnext >= qend)
                break;
            continue;  // nidle == 64 >= refill_min: the next trip refills
        }

        double f, fp;
        eval_f_fp(t, s.xe, s.nfull, nfull_wave, s.hvl, s.vvl, s.R, f, fp);
        const bool was_busy = (s.st != ST_DONE);
        lane_advance(s, f, fp);
        if (was_busy && s.st == ST_DONE) {
            a.d_pfin[orig] = s.x;
            a.key[orig] = nl | (s.conv ? RTMI_KEY_CONV : 0);
        }
    }
}

// t = sum H_i / (V_i sqrt(1 - p^2 V_i^2)) for every traced source, input order.
__global__ __launch_bounds__(RTMI_BLOCK) void time_kernel(BinArgs a)
{
    extern __shared__ double lds[];
    const LayerTable t = carve_table(lds, a.nlayers);
    stage_model(t, a.d_vels, a.d_depths, a.nlayers);
    const int64_t ntiles = (a.nsrc + RTMI_BLOCK - 1) / RTMI_BLOCK;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t k = tile * RTMI_BLOCK + threadIdx.x;
        const bool valid = k < a.nsrc;
        const int kv = valid ? a.key[k] : 0;
        const int nl = kv & RTMI_KEY_NL_MASK;
        const bool traced = nl > 1;  // 1: straight ray, 0: reported by bin_scatter_kernel
        const int il = traced ? nl - 1 : 1;
        const double z = traced ? a.d_src_depth[k] : 1.0;
        const double p = traced ? a.d_pfin[k] : 0.0;
        const int nfull_wave = wave_max_i32(traced ? il : 0);
        double tt = travel_time(t, p, il, nfull_wave, il, z);
        if (!(kv & RTMI_KEY_CONV))
            tt = -999.0;
        if (traced)
            a.d_timeP[k] = tt;
    }
}

}  // namespace rtmi

// ---------------------------------------------------------------------------------------------
// Launchers (host).  No allocation, no synchronisation: graph-capturable.
// ---------------------------------------------------------------------------------------------
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

extern "C" hipError_t rtmi_launch_trace(const rtmi::TraceArgs *args, int nmodels, int max_nlayers, int grid_x,
                                        hipStream_t stream)
{
    using namespace rtmi;
    const TraceArgs a = *args;
    const size_t lds_bytes = (size_t)RTMI_TABLE_ARRAYS * (max_nlayers + 1) * sizeof(double);
    const dim3 grid(grid_x, nmodels, 1), block(RTMI_BLOCK, 1, 1);
    const bool want_p = a.d_p_final != nullptr;
    const bool want_t = a.d_timeP != nullptr;
    const bool want_ss = a.d_obs != nullptr;
    if (want_ss && want_t)
        hipLaunchKernelGGL((trace_kernel<false, true, true>), grid, block, lds_bytes, stream, a);
    else if (want_ss)
        hipLaunchKernelGGL((trace_kernel<false, false, true>), grid, block, lds_bytes, stream, a);
    else if (want_p)
        hipLaunchKernelGGL((trace_kernel<true, true, false>), grid, block, lds_bytes, stream, a);
    else
        hipLaunchKernelGGL((trace_kernel<false, true, false>), grid, block, lds_bytes, stream, a);
    return hipGetLastError();
}

extern "C" hipError_t rtmi_launch_loglike(const double *d_partial, int nparts, const double *d_sd, int64_t ndat,
                                          int nmodels, double *d_logL, hipStream_t stream)
{
    const int block = 64;
    const int grid = (nmodels + block - 1) / block;
    hipLaunchKernelGGL(rtmi::loglike_kernel, dim3(grid), dim3(block), 0, stream, d_partial, nparts, d_sd, ndat,
                       nmodels, d_logL);
    return hipGetLastError();
}

extern "C" size_t rtmi_binned_work_bytes(int64_t nsrc, int nlayers)
{
    // hist | cursor (u32 each, nlayers+2, padded) | key i32 | bidx u32 | bnl i32 | bR bz bp0 pfin f64
    const size_t nb = ((size_t)nlayers + 2 + 63) / 64 * 64;
    const size_t n = ((size_t)nsrc + 63) / 64 * 64;
    return 2 * nb * sizeof(uint32_t) + n * (3 * sizeof(uint32_t) + 4 * sizeof(double));
}

extern "C" hipError_t rtmi_launch_binned(const double *d_vels, const double *d_depths, int nlayers,
                                         const double *d_src_offset, const double *d_src_depth, int64_t nsrc,
                                         double *d_timeP, double *d_p_final, void *d_work, int num_cus,
                                         int waves_per_simd, int refill_min, hipStream_t stream)
{
    using namespace rtmi;
    const size_t nb = ((size_t)nlayers + 2 + 63) / 64 * 64;
    const size_t n = ((size_t)nsrc + 63) / 64 * 64;
    char *w = static_cast<char *>(d_work);
    BinArgs a;
    a.d_vels = d_vels;
    a.d_depths = d_depths;
    a.nlayers = nlayers;
    a.refill_min = refill_min < 1 ? 1 : (refill_min > 64 ? 64 : refill_min);
    a.d_src_offset = d_src_offset;
    a.d_src_depth = d_src_depth;
    a.nsrc = nsrc;
    a.d_timeP = d_timeP;
    a.hist = reinterpret_cast<uint32_t *>(w);
    a.cursor = a.hist + nb;
    w += 2 * nb * sizeof(uint32_t);
    a.bR = reinterpret_cast<double *>(w);
    a.bz = a.bR + n;
    a.bp0 = a.bz + n;
    double *pfin_ws = a.bp0 + n;
    w += 4 * n * sizeof(double);
    a.key = reinterpret_cast<int32_t *>(w);
    a.bidx = reinterpret_cast<uint32_t *>(w) + n;
    a.bnl = reinterpret_cast<int32_t *>(w) + 2 * n;
    a.d_pfin = d_p_final ? d_p_final : pfin_ws;

    hipError_t e = hipMemsetAsync(a.hist, 0, 2 * nb * sizeof(uint32_t), stream);
    if (e != hipSuccess)
        return e;
    const size_t table = (size_t)RTMI_TABLE_ARRAYS * (nlayers + 1) * sizeof(double);
    const size_t bins = (size_t)3 * (nlayers + 2) * sizeof(uint32_t);
    const int64_t ntiles = (nsrc + RTMI_BLOCK - 1) / RTMI_BLOCK;
    int64_t gtiles = (int64_t)num_cus * 8;
    if (gtiles > ntiles)
        gtiles = ntiles;
    const dim3 block(RTMI_BLOCK, 1, 1);
    hipLaunchKernelGGL(bin_count_kernel, dim3((unsigned)gtiles), block, table + bins, stream, a);
    hipLaunchKernelGGL(bin_scatter_kernel, dim3((unsigned)gtiles), block, table + bins, stream, a);
    // solve: waves_per_simd persistent waves on each of the 4 SIMDs of every CU, never more waves
    // than chunks
    const int64_t nchunks = (nsrc + 63) / 64;
    int64_t gsolve = (int64_t)num_cus * waves_per_simd;  // workgroups of 4 waves
    if (gsolve * (RTMI_BLOCK / 64) > nchunks)
        gsolve = (nchunks + RTMI_BLOCK / 64 - 1) / (RTMI_BLOCK / 64);
    if (gsolve < 1)
        gsolve = 1;
    hipLaunchKernelGGL(solve_kernel, dim3((unsigned)gsolve), block, table, stream, a);
    hipLaunchKernelGGL(time_kernel, dim3((unsigned)gtiles), block, table, stream, a);
    return hipGetLastError();
}
