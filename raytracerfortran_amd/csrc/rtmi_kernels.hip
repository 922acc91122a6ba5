// rtmi_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the layered travel-time forward model.
//
// What is computed (reference: /root/reference/subroutineR-quiet.f90, module raymod):
//   per source k:  nl = whichLayer(depths, z_k)                       (:9-32)
//                  stack (H_i, V_i), i <= nl, above the source         (:38-69 InsertLayer)
//                  ray parameter p by the guarded Newton / bisection
//                  hybrid on f(p) = R - sum H_i V_i p / sqrt(1-p^2 V_i^2)   (:77-153, :226-405)
//                  t = sum H_i / (V_i sqrt(1 - p^2 V_i^2)), or -999    (:156-169)
// The reference stops its root-find at |f| < 0.1 m, so t depends on the exact
// iteration sequence: the kernel reproduces the reference's arithmetic operation
// for operation in IEEE fp64.  THIS FILE MUST BE COMPILED WITH -ffp-contract=off
// (build.py does); hipcc's default contraction would fuse 1 - (x*x)*(V*V) and break parity.
//
// Mapping to the machine (DESIGN.md has the numbers):
//   * one lane per source; 64-lane waves; 256-thread workgroups, grid-stride over source tiles;
//   * the model is tiny (<= 2*nlayers+1 doubles) and shared by every lane, so each
//     workgroup derives the layer table once -- thickness H, H*V, V*V, running sum of
//     H/V, running max of V -- into LDS in exactly the reference's rounding order;
//     the per-layer loop reads it with wave-uniform addresses (LDS broadcast), the
//     per-lane entries (layer of the source) with a gather;
//   * the expensive primitive is "evaluate f and f' at x" (one sqrt and two divides
//     per layer).  The three solver branches of GetPTime are folded into one per-lane
//     state machine around a single converged evaluation loop, so a wave never
//     executes the layer loop for one branch while the lanes of another idle;
//     f and f' share their sqrt, and solvebst's duplicate costFunc call (:370,:378)
//     is evaluated once -- same values, fewer instructions;
//   * src_offset / src_depth are read once, coalesced (8 B/lane), timeP written once:
//     24 algorithmic bytes per travel time.  The kernel is fp64-VALU bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtmi_kernels.h"

namespace rtmi {

// Per-lane solver states.  One evaluation of (f, f') separates two transitions.
enum : int {
    ST_INIT = 0,    // evaluated at p0: choose the branch                       (:136-153)
    ST_BIS0 = 1,    // evaluated at x1 = 1e-10: orient the bisection            (:354-365)
    ST_BIS = 2,     // evaluated at xmid: one bisection trip                    (:367-397)
    ST_NEWTON = 3,  // evaluated at x: one Newton iteration                     (:269-311)
    ST_FINAL = 4,   // evaluated at x after 15 updates: the inverted conv test  (:314-330)
    ST_DONE = 5
};

// LDS layout: 8 arrays of (nlayers + 1) doubles.
struct LayerTable {
    double *D;     // D[i]    depth to interface i                       i < nlayers
    double *V;     // V[i]    velocity of layer i                        i <= nlayers
    double *H;     // H[i]    thickness of a fully crossed layer i       i < nlayers
    double *HV;    // HV[i]   H[i]*V[i]                                  i < nlayers
    double *VV;    // VV[i]   V[i]*V[i]                                  i <= nlayers
    double *PS;    // PS[i]   ((H0/V0 + H1/V1) + ...) + Hi/Vi            i < nlayers
    double *VMAX;  // VMAX[i] max(V[0..i])                               i <= nlayers
    double *Q;     // scratch H[i]/V[i]
};

__device__ __forceinline__ LayerTable carve_table(double *lds, int nlayers)
{
    const int n1 = nlayers + 1;
    LayerTable t;
    t.D = lds;
    t.V = lds + n1;
    t.H = lds + 2 * n1;
    t.HV = lds + 3 * n1;
    t.VV = lds + 4 * n1;
    t.PS = lds + 5 * n1;
    t.VMAX = lds + 6 * n1;
    t.Q = lds + 7 * n1;
    return t;
}

// Derive the table of one model.  The roundings are the reference's:
//   H(i) = depths(i) - depths(i-1)            InsertLayer :67
//   (H*V) and (V*V) as separate products      costFunc :195, costFunc_Prime :214-219
//   sum(depths/vp) left to right              GetPTime :112
__device__ __forceinline__ void stage_model(const LayerTable &t, const double *__restrict__ vels,
                                            const double *__restrict__ depths, int nlayers)
{
    for (int i = threadIdx.x; i <= nlayers; i += blockDim.x) {
        const double v = vels[i];
        t.V[i] = v;
        t.VV[i] = v * v;
        if (i < nlayers) {
            const double d = depths[i];
            const double h = (i == 0) ? d : d - depths[i - 1];
            t.D[i] = d;
            t.H[i] = h;
            t.HV[i] = h * v;
            t.Q[i] = h / v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = t.Q[0];
        double m = t.V[0];
        t.PS[0] = s;
        t.VMAX[0] = m;
        for (int i = 1; i <= nlayers; ++i) {
            const double v = t.V[i];
            m = (v > m) ? v : m;  // maxval(): order-free
            t.VMAX[i] = m;
            if (i < nlayers) {
                s = s + t.Q[i];
                t.PS[i] = s;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// costFunc (:184-202) and costFunc_Prime (:206-221) at x for the lane's stack:
// `nfull` fully crossed layers from the table plus the partial layer (hvl, vvl).
// `nfull_wave` = max of nfull over the wave keeps the loop wave-uniform.
__device__ __forceinline__ void eval_f_fp(const LayerTable &t, double x, int nfull, int nfull_wave,
                                          double hvl, double vvl, double R, double &f, double &fp)
{
    const double xx = x * x;
    double sf = 0.0, sp = 0.0;
    for (int i = 0; i < nfull_wave; ++i) {
        const double hv = t.HV[i];
        const double vv = t.VV[i];
        const double s = sqrt(1.0 - xx * vv);
        const double tf = (hv * x) / s;
        const double tp = hv / ((s * s) * s);
        if (i < nfull) {
            sf = sf + tf;
            sp = sp + tp;
        }
    }
    {
        const double s = sqrt(1.0 - xx * vvl);
        sf = sf + (hvl * x) / s;
        sp = sp + hvl / ((s * s) * s);
    }
    f = R - sf;
    fp = -sp;
}

// One source.  Returns the travel time; *p_out gets the final ray parameter.
// All lanes of the wave must call this together (wave-uniform loops inside);
// `valid` = false lanes carry a dummy source and their result is discarded.
__device__ __forceinline__ double trace_one(const LayerTable &t, int nlayers, double R, double z,
                                            bool valid, double &p_out)
{
    // ---- whichLayer (:9-32): first interface strictly below the source -------------------
    int nl = 0;  // 1-based, 0 = not found yet
    for (int i = 0; i < nlayers; ++i) {
        const bool below = t.D[i] > z;  // == (depths(i) - dph > 0)
        if (below && nl == 0)
            nl = i + 1;
        if (__all(nl != 0))
            break;
    }
    if (nl == 0)
        nl = (t.D[nlayers - 1] - z < 0.0) ? nlayers + 1 : nlayers;

    const bool straight = (nl == 1);  // source in the top layer (:94-97)
    const bool active = valid && !straight;

    // ---- InsertLayer (:38-69) as table lookups -------------------------------------------
    const int il = active ? nl - 1 : 1;  // 0-based index of the source's layer; dummy lanes use 1
    const int ilc = il > nlayers ? nlayers : il;
    const int nfull = ilc;               // layers fully crossed
    const double vl = t.V[ilc];
    const double hl = z - t.D[ilc - 1];  // depths_new(nl) - depths_new(nl-1)
    const double hvl = hl * vl;
    const double vvl = t.VV[ilc];
    const double vmax = t.VMAX[ilc];
    const int nfull_wave = wave_max_i32(active ? nfull : 0);

    // ---- GetPTime head (:112-133) -----------------------------------------------------------
    const double c_harmonic = z / (t.PS[ilc - 1] + hl / vl);
    const double cos_t = z / sqrt(R * R + z * z);
    double p0 = cos_t / c_harmonic;
    const double vm1sq = (vmax + 1.0) * (vmax + 1.0);
    // sum(sqrt(1-(p0**2)*(vp+1)**2)) is NaN  <=>  (p0*p0)*(vmax+1)^2 > 1  (monotone in V; the
    // subtraction from 1 has the exact sign).  The reference loops forever for NaN/Inf p0
    // (NaN source data); here such a lane gives up and reports -999.
    bool hopeless = false;
    {
        int guard = 0;
        while (active && !hopeless && !((p0 * p0) * vm1sq <= 1.0)) {
            p0 = p0 / 2.0;
            if (++guard > 2200 || p0 != p0)
                hopeless = true;
        }
    }

    const double vmi = 1.0 / vmax;         // 1/maxval(vp)
    const double safe = vmi - 1e-10;       // 1/maxval(V)-(1d-10)      (:142, :388)

    int st = (active && !hopeless) ? ST_INIT : ST_DONE;
    double xe = p0;   // where the next evaluation happens
    double x = p0;    // Newton iterate
    double xb = 0.0;  // bisection bracket end `x` of solvebst
    double dx = 0.0;
    int k = 0;        // trip counter of the current loop
    bool conv = false;

    while (__any(st != ST_DONE)) {
        double f, fp;
        eval_f_fp(t, xe, nfull, nfull_wave, hvl, vvl, R, f, fp);

        // Exactly one transition per lane and evaluation; no early `continue`, so the
        // wave reconverges at the bottom of the loop before the next __any().
        bool newton_step = (st == ST_NEWTON);
        if (st == ST_INIT) {
            const double check_x = xe - f / fp;
            if (f < 0.0 || check_x < safe) {
                st = ST_NEWTON;  // solve(p0): its first costFunc/costFunc_Prime pair is this one
                k = 1;
                x = xe;
                newton_step = true;
            } else {
                st = ST_BIS0;  // solvebst(1d-10, 1/maxval(vp)-1d-12)
                xe = 1e-10;
            }
        } else if (st == ST_BIS0) {
            const double x1 = 1e-10, x2 = vmi - 1e-12;
            if (f < 0.0) {
                xb = x1;
                dx = x2 - x1;
            } else {
                xb = x2;
                dx = x1 - x2;
            }
            k = 1;
            dx = dx * 0.5;
            xe = xb + dx;
            st = ST_BIS;
        } else if (st == ST_BIS) {
            // xe is xmid, f is fmid (== fx), fp is fxprime
            bool done = false;
            if (f < 0.0)
                xb = xe;
            if (f == 0.0) {
                done = true;
            } else {
                const double check_x = xb - f / fp;
                if (check_x < safe) {
                    xb = xe;
                    done = true;
                } else if (fabs(f) < 0.1) {
                    done = true;
                }
            }
            if (!done) {
                if (k == 20) {
                    done = true;
                } else {
                    ++k;
                    dx = dx * 0.5;
                    xe = xb + dx;
                }
            }
            if (done) {
                st = ST_NEWTON;  // p0 = p_final; call solve(p0, ...)   (:151-152)
                k = 1;
                x = xb;
                xe = xb;
            }
        } else if (st == ST_FINAL) {
            conv = fabs(f) > 0.1;  // the reference's inverted test (:327-330)
            st = ST_DONE;
        }

        if (newton_step) {
            if (fabs(f) < 0.1) {
                conv = true;
                st = ST_DONE;
            } else {
                x = x - f / fp;
                if (x > vmi)
                    x = vmi - 1e-9;
                xe = x;
                if (k == 15)
                    st = ST_FINAL;
                else
                    ++k;
            }
        }
    }

    // ---- travel time (:156, :165-169) ---------------------------------------------------------
    const double pp = x * x;
    double tt = 0.0;
    for (int i = 0; i < nfull_wave; ++i) {
        const double cosv = sqrt(1.0 - pp * t.VV[i]);
        const double ti = t.H[i] / (t.V[i] * cosv);
        if (i < nfull)
            tt = tt + ti;
    }
    {
        const double cosv = sqrt(1.0 - pp * vvl);
        tt = tt + hl / (vl * cosv);
    }
    if (!conv)
        tt = -999.0;
    p_out = x;
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
    const size_t lds_bytes = (size_t)8 * (max_nlayers + 1) * sizeof(double);
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
