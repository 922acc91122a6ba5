// rtmi_device.hpp -- device-side building blocks shared by every kernel of rtmi_kernels.hip.
//
// Everything here restates /root/reference/subroutineR-quiet.f90 (module raymod) operation for
// operation in IEEE fp64; the translation unit MUST be built with -ffp-contract=off.
#ifndef RTMI_DEVICE_HPP
#define RTMI_DEVICE_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

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

// The layer table of one model in LDS: RTMI_TABLE_ARRAYS arrays of (nlayers + 1) doubles.
#define RTMI_TABLE_ARRAYS 9
struct LayerTable {
    double *D;     // D[i]    depth to interface i                       i < nlayers
    double *V;     // V[i]    velocity of layer i                        i <= nlayers
    double *H;     // H[i]    thickness of a fully crossed layer i       i < nlayers
    double *HV;    // HV[i]   H[i]*V[i]                                  i < nlayers
    double *VV;    // VV[i]   V[i]*V[i]                                  i <= nlayers
    double *PS;    // PS[i]   ((H0/V0 + H1/V1) + ...) + Hi/Vi            i < nlayers
    double *VMAX;  // VMAX[i] max(V[0..i])                               i <= nlayers
    double *VMI;   // VMI[i]  1/VMAX[i]                                  i <= nlayers
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
    t.VMI = lds + 7 * n1;
    t.Q = lds + 8 * n1;
    return t;
}

// Derive the table of one model (whole workgroup; ends with a barrier).  Roundings are the reference's:
//   H(i) = depths(i) - depths(i-1)            InsertLayer :67
//   (H*V) and (V*V) as separate products      costFunc :195, costFunc_Prime :214-219
//   sum(depths/vp) left to right              GetPTime :112
//   1/maxval(vp)                              :142, :148, :299, :388
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
    for (int i = threadIdx.x; i <= nlayers; i += blockDim.x)
        t.VMI[i] = 1.0 / t.VMAX[i];
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

// whichLayer (:9-32): 1-based index of the first interface strictly below z; nlayers+1 below the
// last one; nlayers when z sits exactly on the last interface (diff == 0 is neither > 0 nor < 0).
// Wave-uniform loop with early exit; every lane of the wave must call it.
__device__ __forceinline__ int which_layer(const LayerTable &t, int nlayers, double z)
{
    int nl = 0;
    for (int i = 0; i < nlayers; ++i) {
        const bool below = t.D[i] > z;  // == (depths(i) - dph > 0)
        if (below && nl == 0)
            nl = i + 1;
        if (__all(nl != 0))
            break;
    }
    if (nl == 0)
        nl = (t.D[nlayers - 1] - z < 0.0) ? nlayers + 1 : nlayers;
    return nl;
}

// costFunc (:184-202) and costFunc_Prime (:206-221) at x for the lane's stack:
// `nfull` fully crossed layers from the table plus the partial layer (hvl, vvl).
// `nfull_wave` >= max of nfull over the wave keeps the loop wave-uniform.
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

// Solver state of one source (one lane).
struct Lane {
    double R;     // src_offset
    double hvl;   // (z - D[il-1]) * V[il]   partial layer of the source
    double vvl;   // V[il]^2
    double vmi;   // 1/maxval(vp)
    double x;     // Newton iterate / final ray parameter
    double xe;    // where the next (f, f') evaluation happens
    double xb;    // bisection bracket end `x` of solvebst
    double dx;
    int nfull;    // fully crossed layers = il
    int st;
    int k;        // trip counter of the current loop
    bool conv;
};

// The source's layer-dependent constants (InsertLayer :38-69 as table lookups).  il = nl - 1 >= 1.
__device__ __forceinline__ void lane_setup(Lane &s, const LayerTable &t, int il, double R, double z)
{
    const double vl = t.V[il];
    const double hl = z - t.D[il - 1];  // depths_new(nl) - depths_new(nl-1)
    s.R = R;
    s.hvl = hl * vl;
    s.vvl = t.VV[il];
    s.vmi = t.VMI[il];
    s.nfull = il;
}

// GetPTime head (:112-133): the starting ray parameter.  Returns NaN when the reference would loop
// forever (NaN/Inf source data): such a source is reported as -999.
__device__ __forceinline__ double initial_p0(const LayerTable &t, int il, double R, double z)
{
    const double vl = t.V[il];
    const double hl = z - t.D[il - 1];
    const double c_harmonic = z / (t.PS[il - 1] + hl / vl);
    const double cos_t = z / sqrt(R * R + z * z);
    double p0 = cos_t / c_harmonic;
    const double vmax = t.VMAX[il];
    const double vm1sq = (vmax + 1.0) * (vmax + 1.0);
    // sum(sqrt(1-(p0**2)*(vp+1)**2)) is NaN  <=>  (p0*p0)*(vmax+1)^2 > 1  (monotone in V; the
    // subtraction from 1 has the exact sign).
    int guard = 0;
    while (!((p0 * p0) * vm1sq <= 1.0)) {
        p0 = p0 / 2.0;
        if (++guard > 2200 || p0 != p0)
            return __builtin_nan("");
    }
    return p0;
}

__device__ __forceinline__ void lane_start(Lane &s, double p0)
{
    s.x = p0;
    s.xe = p0;
    s.xb = 0.0;
    s.dx = 0.0;
    s.k = 0;
    s.conv = false;
    s.st = ST_INIT;
}

// One transition of the per-lane state machine after (f, f') was evaluated at s.xe.
// GetPTime's three-way branch (:136-153), solvebst (:339-405) and solve (:226-332).
__device__ __forceinline__ void lane_advance(Lane &s, double f, double fp)
{
    const double safe = s.vmi - 1e-10;  // 1/maxval(V)-(1d-10)      (:142, :388)
    // cf/cfp of GetPTime (:138), deltax of solvebst (:384) and of solve (:294) are the same quotient of the
    // same two operands: divide once, ahead of the (lane-divergent) state dispatch.
    const double q = f / fp;
    bool newton_step = (s.st == ST_NEWTON);
    if (s.st == ST_INIT) {
        const double check_x = s.xe - q;
        if (f < 0.0 || check_x < safe) {
            s.st = ST_NEWTON;  // solve(p0): its first costFunc/costFunc_Prime pair is this one
            s.k = 1;
            s.x = s.xe;
            newton_step = true;
        } else {
            s.st = ST_BIS0;  // solvebst(1d-10, 1/maxval(vp)-1d-12)
            s.xe = 1e-10;
        }
    } else if (s.st == ST_BIS0) {
        const double x1 = 1e-10, x2 = s.vmi - 1e-12;
        if (f < 0.0) {
            s.xb = x1;
            s.dx = x2 - x1;
        } else {
            s.xb = x2;
            s.dx = x1 - x2;
        }
        s.k = 1;
        s.dx = s.dx * 0.5;
        s.xe = s.xb + s.dx;
        s.st = ST_BIS;
    } else if (s.st == ST_BIS) {
        // xe is xmid, f is fmid (== fx: the reference evaluates costFunc(xmid) twice), fp is fxprime
        bool done = false;
        if (f < 0.0)
            s.xb = s.xe;
        if (f == 0.0) {
            done = true;
        } else {
            const double check_x = s.xb - q;
            if (check_x < safe) {
                s.xb = s.xe;
                done = true;
            } else if (fabs(f) < 0.1) {
                done = true;
            }
        }
        if (!done) {
            if (s.k == 20) {
                done = true;
            } else {
                ++s.k;
                s.dx = s.dx * 0.5;
                s.xe = s.xb + s.dx;
            }
        }
        if (done) {
            s.st = ST_NEWTON;  // p0 = p_final; call solve(p0, ...)   (:151-152)
            s.k = 1;
            s.x = s.xb;
            s.xe = s.xb;
        }
    } else if (s.st == ST_FINAL) {
        s.conv = fabs(f) > 0.1;  // the reference's inverted test (:327-330)
        s.st = ST_DONE;
    }

    if (newton_step) {
        if (fabs(f) < 0.1) {
            s.conv = true;
            s.st = ST_DONE;
        } else {
            s.x = s.x - q;
            if (s.x > s.vmi)
                s.x = s.vmi - 1e-9;
            s.xe = s.x;
            if (s.k == 15)
                s.st = ST_FINAL;
            else
                ++s.k;
        }
    }
}

// Travel time for a known ray parameter (:156, :165-166).  Wave-uniform loop.
__device__ __forceinline__ double travel_time(const LayerTable &t, double p, int nfull, int nfull_wave, int il,
                                              double z)
{
    const double pp = p * p;
    double tt = 0.0;
    for (int i = 0; i < nfull_wave; ++i) {
        const double cosv = sqrt(1.0 - pp * t.VV[i]);
        const double ti = t.H[i] / (t.V[i] * cosv);
        if (i < nfull)
            tt = tt + ti;
    }
    const double hl = z - t.D[il - 1];
    const double cosv = sqrt(1.0 - pp * t.VV[il]);
    tt = tt + hl / (t.V[il] * cosv);
    return tt;
}

}  // namespace rtmi
#endif
