// rtmi_f32fast.hip -- the single-precision FAST forward model (BASELINE config 5, opt-in): gfx950 kernel.
//
// What it is NOT: the reference's iteration.  rtmi_trace_device_f32 (rtmi_kernels.hip, T = float) repeats
// subroutineR-quiet.f90 operation for operation in float; its guarded Newton / bisection hybrid stops at |f| < 0.1 m,
// which in single precision is the resolution of the arithmetic itself, so near-critical rays wander (sentinels, |dt| up
// to seconds) and the path is slower than double precision.  This kernel solves the SAME equation
//     R = sum_i H_i V_i p / sqrt(1 - p^2 V_i^2)            (costFunc, subroutineR-quiet.f90:184-202)
//     t = sum_i H_i / (V_i sqrt(1 - p^2 V_i^2))            (GetPTime :156, :165-166)
// over the SAME stack of layers (whichLayer :9-32, InsertLayer :38-69) to single-precision roundoff, in a variable in which
// nothing cancels:  s = tan(theta) in the fastest layer of the stack,  p = s / (vmax sqrt(1 + s^2)).  With r_i = V_i / vmax and
// c_i = 1 - r_i^2 = (vmax - V_i)(vmax + V_i) / vmax^2  (tabulated per stack, from double precision):
//     1 - p^2 V_i^2 = (1 + c_i s^2) / (1 + s^2)
//     X(s) = s sum_i (H_i r_i) / sqrt(1 + c_i s^2)          = R
//     t(s) = sqrt(1 + s^2) sum_i (H_i / V_i) / sqrt(1 + c_i s^2)
// X is increasing and CONCAVE in s (linear in the fastest layer, c = 0), so Newton from the lower bound s0 = R / sum H_i r_i
// climbs monotonically to the root: no bracket, no clamp, no state machine -- every lane runs the same few instructions.
// Measured against a fully converged double-precision solution of the same equations: max |dt| 8e-7 s; against the
// reference (which stops 0.1 m short of the root): median 2.4e-7 s, 99.9th percentile 3.4e-5 s = the reference's own
// stopping tolerance (tests/test_gpu_f32fast.py).
//
// Machine mapping: persistent workgroups of 256 lanes; per tile of up to 1280 sources a counting sort by source layer in LDS (so
// that the 64 lanes of a wave walk the same number of layers), one source per lane, the per-stack coefficient rows stored
// as PAIRS of layers -- {c_i, c_i+1, (H r)_i, (H r)_i+1} and {(H / V)_i, (H / V)_i+1} -- so that one ds_read_b128 (broadcast when
// the wave shares a row) feeds two layer terms and the arithmetic on them is packed (v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32: two layers per instruction; the kernel is bound by vector issue), v_rsq_f32 for every 1 / sqrt, results
// returned to input order through LDS so that the stores are coalesced.  12 B of HBM per source.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "rtmi_kernels.h"

namespace rtmi {

#define F32F_BLOCK 256
#ifndef F32F_PER_THREAD
// tiles of up to 1280 sources, four workgroups (16 waves) per CU.  Measured (profiles/r04q/f32fast_tile.txt, one box): at config
// 5's per-GPU share a workgroup's slice is 1221 sources -- ONE tile of five per lane (0.0373 ms) instead of two of 640 with four
// per lane (0.0421) or six (0.0402); at 1e7 sources 0.2244 / 0.2271 / 0.2220.  Tiles of 2048 (three workgroups per CU) were 12 %
// slower in round 3.
#define F32F_PER_THREAD 5
#endif
#define F32F_TILE (F32F_BLOCK * F32F_PER_THREAD)
#define F32F_MAXIT 12
#ifndef F32F_CONV_REL
#define F32F_CONV_REL 1e-4f  // the last step's relative size below which the iterate it lands on is the root to roundoff (see the loop)
#endif

#ifdef F32F_CLOCKS
// instrumented build (profiles/ubench/f32fast_phases.hip): cycles of wave 0 of every workgroup per phase, kept in registers
// and written once when the workgroup leaves
__device__ unsigned long long f32f_clk[8 * 1024];
#define F32F_STAMP(i)                                  \
    do {                                               \
        const unsigned long long now_ = clock64();     \
        clk_acc_[i] += now_ - clk_prev_;               \
        clk_prev_ = now_;                              \
    } while (0)
#else
#define F32F_STAMP(i)
#endif

typedef float v2f __attribute__((ext_vector_type(2)));  // a register pair: the operand of the packed single-precision instructions

__global__ __launch_bounds__(F32F_BLOCK) void f32fast_kernel(F32FastArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int NL = a.nlayers, W = NL + 1;  // source layers il = 0 .. NL
    const int WS = (W + 3) & ~3;           // row stride: the layer loops run four entries at a time over zero-padded rows
    const int WH = WS >> 1;                // pairs of layers per row (even)
    float4 *tabA = reinterpret_cast<float4 *>(lds_raw);          // [W][WH] tabA[il * WH + i / 2] = {c_i, c_i+1, H_i r_i, H_i+1 r_i+1}, zeros from layer il on
    float2 *tabB = reinterpret_cast<float2 *>(tabA + (size_t)W * WH);  // [W][WH] {H_i / V_i, H_i+1 / V_i+1}, likewise
    float4 *diag = reinterpret_cast<float4 *>(tabB + (size_t)W * WH);  // [W] the source's own layer: {c_l, r_l, 1 / V_l, sum_{i < il} H_i r_i}
    float4 *diag2 = diag + W;                                    // [W] {sum of H r over the stack's fastest full layers, sum of H r / sqrt(c)
                                                                 //      over its slower ones, 1 / sqrt(c_l) (0 if the own layer is a fastest one), 0}
    float *vmi = reinterpret_cast<float *>(diag2 + W);           // [W] 1 / vmax of the stack
    float *D = vmi + ((W + 3) & ~3);                             // [NL] interface depths (every array below starts 16-byte aligned)
    float *sz = D + ((NL + 3) & ~3);                             // [TILE] depth, sorted by layer
    float *sR = sz + F32F_TILE;                                  // [TILE] offset
    float *tout = sR + F32F_TILE;                                // [TILE] results in input order
    float *pout = tout + F32F_TILE;                              // [TILE] ray parameters in input order
    uint32_t *hist = reinterpret_cast<uint32_t *>(pout + F32F_TILE);  // [W] counts, then first positions
    uint32_t *sidx = hist + ((W + 3) & ~3);                           // [TILE] local index | layer << 16 of the sorted record
    double *vmaxd = reinterpret_cast<double *>(sidx + F32F_TILE);      // [W] running maximum of the velocities (staging, like the next four)
    double *ivm = vmaxd + W;                                           // [W] 1 / vmax of the stack
    double *ivm2 = ivm + W;                                            // [W] 1 / vmax^2
    double *ivl = ivm2 + W;                                            // [W] 1 / V of the layer
    double *hvd = ivl + W;                                             // [W] H / V of the layer

#ifdef F32F_CLOCKS
    unsigned long long clk_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long clk_prev_ = clock64();
#endif
    // ---- the model's tables (once per workgroup; the model itself goes through LDS: one round trip to memory) ---------------
    float *Vs = tout;  // [W] velocities, staged in the results area (free until the first tile)
    for (int i = threadIdx.x; i < W; i += F32F_BLOCK)
        Vs[i] = a.d_vels[i];
    for (int i = threadIdx.x; i < NL; i += F32F_BLOCK)
        D[i] = a.d_depths[i];
    __syncthreads();
    // this mode's precondition: interfaces in non-decreasing order from a non-negative depth, positive finite velocities
    bool bad = false;
    for (int i = threadIdx.x; i < W; i += F32F_BLOCK) {
        const float v = Vs[i];
        bad = bad || !(v > 0.0f) || !(v < 3.0e38f);
        if (i < NL) {
            const float d = D[i], dp = i > 0 ? D[i - 1] : 0.0f;
            bad = bad || !(d >= dp) || !(d < 3.0e38f);
        }
    }
    if (__syncthreads_or(bad ? 1 : 0)) {
        if (threadIdx.x == 0 && blockIdx.x == 0 && a.d_status)
            atomicOr(a.d_status, 8);
        for (int64_t k = (int64_t)blockIdx.x * F32F_BLOCK + threadIdx.x; k < a.nsrc; k += (int64_t)gridDim.x * F32F_BLOCK) {
            a.d_timeP[k] = __builtin_nanf("");
            if (a.d_p_final)
                a.d_p_final[k] = __builtin_nanf("");
        }
        return;
    }
    // per layer and per stack height, one lane each: the divisions of the whole staging (the pairs below only multiply)
    auto thick = [&](int i) { return i == 0 ? (double)D[0] : (double)D[i] - (double)D[i - 1]; };
    for (int il = threadIdx.x; il < W; il += F32F_BLOCK) {
        float mf = Vs[0];  // (the maximum of floats is a float: no rounding in either precision)
#pragma unroll 4
        for (int i = 1; i <= il; ++i)
            mf = fmaxf(mf, Vs[i]);
        const double m = (double)mf;
        const double im = 1.0 / m, vl = (double)Vs[il];
        vmaxd[il] = m;
        ivm[il] = im;
        ivm2[il] = 1.0 / (m * m);
        ivl[il] = 1.0 / vl;
        hvd[il] = il < NL ? thick(il) / vl : 0.0;
        vmi[il] = (float)im;
    }
    __syncthreads();
    F32F_STAMP(6);
    // The table entries, one (stack, layer) pair per lane, and the three sums over the stack that the start values need: the
    // pairs leave their addends in LDS (the tiles' staging area is free until the first tile), then one lane per stack adds
    // them in layer order (the same bits on every launch).  [Round 3 had one LANE per stack walk its layers with three
    // double-precision divides and a square root per layer, and every workgroup -- four per CU, on the same vector units --
    // does this before its first source: 10 us of a 44 us launch at config 5's per-GPU share
    // (profiles/r04m/f32fast_phases.txt).  Here a pair costs a dozen multiplications: the reciprocals are per stack / per
    // layer (above), and the saturated offset's 1 / sqrt(c) is needed to 1e-6 only (the start value it bounds carries that
    // margin), so the single-precision v_rsq_f32 serves.]
    {
        float *fa = reinterpret_cast<float *>(tabA);
        float *fb = reinterpret_cast<float *>(tabB);
        double2 *addend = reinterpret_cast<double2 *>(sz);  // [rows of a batch][WS] {H r, H r / sqrt(c) or -(H r) for a fastest layer}: over sz, sR (the velocities sit in tout)
        const int rows_per_batch = (int)((size_t)2 * F32F_TILE * sizeof(float) / ((size_t)WS * sizeof(double2)));  // >= 8 (W <= 64)
        for (int r0 = 0; r0 < W; r0 += rows_per_batch) {
            const int nr = W - r0 < rows_per_batch ? W - r0 : rows_per_batch;
            for (int e = threadIdx.x; e < nr * WS; e += F32F_BLOCK) {
                const int il = r0 + e / WS, i = e % WS;
                float c = 0.0f, hr = 0.0f, hv = 0.0f;
                double2 ad = make_double2(0.0, 0.0);
                if (i < il) {
                    const double vm = vmaxd[il], vi = (double)Vs[i], ai = thick(i) * (vi * ivm[il]);
                    c = (float)(((vm - vi) * (vm + vi)) * ivm2[il]);
                    hr = (float)ai;
                    hv = (float)hvd[i];
                    ad.x = ai;
                    ad.y = c > 0.0f ? ai * (double)__builtin_amdgcn_rsqf(c) : -ai;
                }
                const int pr = il * WH + (i >> 1), hf = i & 1;
                fa[4 * pr + hf] = c;
                fa[4 * pr + 2 + hf] = hr;
                fb[2 * pr + hf] = hv;
                addend[e] = ad;
            }
            __syncthreads();
            for (int r = threadIdx.x; r < nr; r += F32F_BLOCK) {
                const int il = r0 + r;
                double afull = 0.0, hc = 0.0, binf = 0.0;
                // (over the whole row, four entries a trip, so that the reads are in flight together: the addends from layer il
                // on are {0, 0}, and x + 0 = x)
                for (int i = 0; i < WS; i += 4) {
                    double2 ad[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        ad[u] = addend[r * WS + i + u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        afull += ad[u].x;
                        binf += ad[u].y >= 0.0 ? ad[u].y : 0.0;
                        hc -= ad[u].y >= 0.0 ? 0.0 : ad[u].y;
                    }
                }
                const double vm = vmaxd[il], vl = (double)Vs[il];
                const float cl = (float)(((vm - vl) * (vm + vl)) * ivm2[il]);
                diag[il] = make_float4(cl, (float)(vl * ivm[il]), (float)ivl[il], (float)afull);
                // rounded so that the start value stays a LOWER bound of the root: hc up, the saturated offset up (the margin
                // covers v_rsq_f32's 1e-7)
                diag2[il] = make_float4((float)(hc * (1.0 + 1e-6)), (float)(binf * (1.0 + 2e-6)), cl > 0.0f ? (float)(1.0 / sqrt((double)cl)) : 0.0f, 0.0f);
            }
            __syncthreads();
        }
    }

    F32F_STAMP(0);  // the model's tables
    const int lane = threadIdx.x & 63;
    // Every workgroup a CONTIGUOUS slice of the sources, cut into equal tiles of at most F32F_TILE.  [Round 3 dealt fixed tiles
    // of 1024 round-robin: at config 5's per-GPU share -- 1.25e6 sources = 1221 tiles for 1024 workgroups -- a fifth of the
    // workgroups took two tiles and the rest one.  Equal slices measured the same (profiles/r04j/f32fast.txt: the four
    // workgroups of a CU even that out between them); kept because a slice is also one contiguous read.]
    const int64_t per = ((a.nsrc + gridDim.x - 1) / gridDim.x + 63) / 64 * 64;
    const int64_t s0 = (int64_t)blockIdx.x * per, s1 = s0 + per < a.nsrc ? s0 + per : a.nsrc;
    const int nsub = s1 > s0 ? (int)((s1 - s0 + F32F_TILE - 1) / F32F_TILE) : 0;
    const int64_t sub = nsub ? ((s1 - s0 + nsub - 1) / nsub + 63) / 64 * 64 : 0;  // <= F32F_TILE
    for (int t = 0; t < nsub; ++t) {
        const int64_t base = s0 + t * sub;
        const int64_t lim = base + sub < s1 ? base + sub : s1;  // this tile's sources: [base, lim)
        // ---- load, locate, count ----------------------------------------------------------------------------------
        for (int i = threadIdx.x; i < W; i += F32F_BLOCK)
            hist[i] = 0;
        __syncthreads();
        float z[F32F_PER_THREAD], R[F32F_PER_THREAD];
        int il[F32F_PER_THREAD];
        uint32_t rank[F32F_PER_THREAD];
#pragma unroll
        for (int j = 0; j < F32F_PER_THREAD; ++j) {
            const int64_t k = base + j * F32F_BLOCK + threadIdx.x;
            const bool valid = k < lim;
            z[j] = valid ? a.d_src_depth[k] : 0.0f;
            R[j] = valid ? a.d_src_offset[k] : 0.0f;
        }
#ifdef F32F_CLOCKS
        if (!(z[0] + R[0] == 12345.678f))  // (waits for the first loads)
#endif
            F32F_STAMP(1);  // sources loaded
        // whichLayer (:9-32): 1-based index of the first interface strictly below z; NL + 1 below the last one; NL when z sits
        // exactly on the last interface (neither > nor <).  With sorted interfaces (this mode's precondition, checked above)
        // that index is 1 + the number of interfaces <= z.  il = index - 1: the source's layer.
        int cnt[F32F_PER_THREAD];
#pragma unroll
        for (int j = 0; j < F32F_PER_THREAD; ++j)
            cnt[j] = 0;
        for (int i = 0; i < NL; ++i) {
            const float di = D[i];  // one (broadcast) LDS read serves the thread's sources
#pragma unroll
            for (int j = 0; j < F32F_PER_THREAD; ++j)
                cnt[j] += di <= z[j] ? 1 : 0;
        }
        const float dlast = D[NL - 1];
#pragma unroll
        for (int j = 0; j < F32F_PER_THREAD; ++j) {
            int nl = cnt[j] + 1;
            if (cnt[j] == NL)
                nl = (dlast - z[j] < 0.0f) ? NL + 1 : NL;
            if (!(z[j] == z[j]))
                nl = NL;  // NaN depth: every comparison of the reference is false
            il[j] = nl - 1;
            const int64_t k = base + j * F32F_BLOCK + threadIdx.x;
            rank[j] = k < lim ? atomicAdd(&hist[il[j]], 1u) : 0u;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            // exclusive prefix of the (<= 64) bins
            const uint32_t c = lane < W ? hist[lane] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off, 64);
                if (lane >= off)
                    incl += o;
            }
            if (lane < W)
                hist[lane] = incl - c;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < F32F_PER_THREAD; ++j) {
            const int loc = j * F32F_BLOCK + threadIdx.x;
            if (base + loc < lim) {
                const uint32_t pos = hist[il[j]] + rank[j];
                sz[pos] = z[j];
                sR[pos] = R[j];
                sidx[pos] = (uint32_t)loc | ((uint32_t)il[j] << 16);
            }
        }
        __syncthreads();
        F32F_STAMP(2);  // located, counted, sorted
        const int nvalid = (int)(lim - base);
        // ---- solve, one sorted record per lane ------------------------------------------------------------------------
        for (int q0 = (threadIdx.x & ~63); q0 < nvalid; q0 += F32F_BLOCK) {
            const int q = q0 + lane;
            const bool valid = q < nvalid;
            // costFunc is odd in p (subroutineR-quiet.f90:184-202), so a source at offset -R has the mirror ray of the one at R:
            // the same travel time, the ray parameter with the opposite sign -- solved for |R| (the tolerances below are relative to it)
            const float Rs = valid ? sR[q] : 1.0f;
            const float zq = valid ? sz[q] : 1.0f, Rq = __builtin_fabsf(Rs);
            const uint32_t rec = valid ? sidx[q] : 0u;
            const int ilq = (int)(rec >> 16);  // the record's layer
            float tt, pp = __builtin_nanf("");
            int ilv = ilq;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                ilv = max(ilv, __shfl_xor(ilv, off, 64));
#ifdef F32F_NO_SOLVE  // (profiles/ubench/f32fast_phases.hip: the kernel without its layer loops, to tell whose LDS traffic the counters show)
            const int ilw = 0;
#else
            const int ilw = __builtin_amdgcn_readfirstlane(ilv);  // the wave's deepest stack, as a scalar
#endif
            if (ilw == 0) {
                tt = __builtin_sqrtf(zq * zq + Rq * Rq) * diag[0].z;  // top layer (:94-97): t = sqrt(z^2 + R^2) / v_1
            } else {
                const int ile = ilq > 0 ? ilq : 1;  // a straight ray in a mixed wave rides along on stack 1 (result replaced below)
                const float4 dg = diag[ile];
                const float hl = zq - D[ile - 1];
                const float cl = dg.x, al = hl * dg.y, bl = hl * dg.z;
                const float4 *rowA = tabA + (size_t)ile * WH;
                const float2 *rowB = tabB + (size_t)ile * WH;
                // two lower bounds of the root: X(s) <= s sum H_i r_i, and X(s) <= hc s + B with hc the fastest layers' share
                // and B what the slower layers can cover at most (s -> infinity): the second one starts a near-critical ray,
                // the kind that would otherwise climb for ten iterations, next to its root
                const float4 d2 = diag2[ile];
                const float hcq = d2.x + (cl > 0.0f ? 0.0f : al), bq = d2.y + al * d2.z;
                float s = Rq * __builtin_amdgcn_rcpf(dg.w + al);
                if (hcq > 0.0f)
                    s = fmaxf(s, (Rq - bq) * __builtin_amdgcn_rcpf(hcq) * 0.999999f);
                // ... and from there three Newton steps on a TWO-layer surrogate that has the stack's slope at s = 0 and its
                // asymptote at infinity -- the fastest layers as they are, all slower ones as one layer of the same total
                // H r and the same saturated offset -- which cost no layer loop and land within a few per cent of the
                // root (on either side: a start to the right of it steps back once, then climbs): 5.3 -> 3.9 iterations of
                // the real thing per wave (profiles/r03g: numpy prototype and the kernel)
                {
                    const float As = fmaxf((dg.w + al) - hcq, 0.0f);
                    const float qe = bq > 0.0f ? As * __builtin_amdgcn_rcpf(bq) : 0.0f;
                    const float ce = qe * qe;
#pragma unroll
                    for (int it = 0; it < 3; ++it) {
                        const float y = __builtin_amdgcn_rsqf(__builtin_fmaf(ce, s * s, 1.0f));
                        const float ay = As * y;
                        const float g = __builtin_fmaf(hcq, s, ay * s) - Rq;
                        const float gp = __builtin_fmaf(ay * y, y, hcq);
                        const float sn = s - g * __builtin_amdgcn_rcpf(gp);
                        s = (sn > 0.0f && sn < 3.0e38f) ? sn : s;  // (a degenerate surrogate leaves the bound as it is)
                    }
                }
                bool conv = !valid || ilq == 0;
                for (int it = 0; it < F32F_MAXIT && __any(!conv); ++it) {
                    const float ss = s * s;
                    // two layers per instruction: the sums as {even layers, odd layers}, added at the end
                    const v2f ss2 = {ss, ss}, one2 = {1.0f, 1.0f};
                    v2f S1v = {0.0f, 0.0f}, S3v = {0.0f, 0.0f};
                    for (int i = 0; i < ilw; i += 4) {  // ilw is a scalar: no lane is masked, rows are zero past the lane's own stack
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const float4 e = rowA[(i >> 1) + u];  // {c_i, c_i+1, (H r)_i, (H r)_i+1}
                            const v2f c = {e.x, e.y}, hr = {e.z, e.w};
                            const v2f w = __builtin_elementwise_fma(c, ss2, one2);
                            const v2f y = {__builtin_amdgcn_rsqf(w.x), __builtin_amdgcn_rsqf(w.y)};
                            const v2f ty = hr * y;
                            S1v += ty;
                            S3v = __builtin_elementwise_fma(ty * y, y, S3v);
                        }
                    }
                    float S1 = S1v.x + S1v.y, S3 = S3v.x + S3v.y;
                    {
                        const float y = __builtin_amdgcn_rsqf(__builtin_fmaf(cl, ss, 1.0f));
                        const float ty = al * y;
                        S1 += ty;
                        S3 = __builtin_fmaf(ty * y, y, S3);
                    }
                    const float ds = (Rq - s * S1) * __builtin_amdgcn_rcpf(S3);
                    const float sn = s + ds;
                    // The iterates climb (X is concave: a Newton step from the left stays left of the root).  With e = root - s the
                    // step is ds = e (1 + O(e / s)) and the next error is e' = |X''(xi)| e^2 / (2 X'(s)), xi in [s, root];
                    // X' = sum a_i y_i^3 decreases and  s |X''| / X' = 3 sum a_i y_i^3 (c_i s^2 y_i^2) / sum a_i y_i^3 < 3,  so
                    // e' / root < 1.5 (e / root)^2: a step of less than 1e-4 s lands within 2e-8 of the root, a quarter of the
                    // single-precision spacing -- the evaluation that would only confirm it is not made [round 3 went on until
                    // the step itself was below 1e-6 s: one more pass over the stack for most waves].  Also true for NaN: such a
                    // lane stops.
                    const bool now = !(ds > F32F_CONV_REL * sn) && !(ds < -1e-4f * s);  // (a step back that is no roundoff: go on)
                    s = conv ? s : sn;  // a converged lane is frozen: its result does not depend on its neighbours
                    conv = conv || now;
                }
                const float ss = s * s;
                const v2f ss2 = {ss, ss}, one2 = {1.0f, 1.0f};
                v2f S1v = {0.0f, 0.0f}, SBv = {0.0f, 0.0f};
                for (int i = 0; i < ilw; i += 4) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float4 e = rowA[(i >> 1) + u];
                        const float2 b = rowB[(i >> 1) + u];
                        const v2f c = {e.x, e.y}, hr = {e.z, e.w}, hv = {b.x, b.y};
                        const v2f w = __builtin_elementwise_fma(c, ss2, one2);
                        const v2f y = {__builtin_amdgcn_rsqf(w.x), __builtin_amdgcn_rsqf(w.y)};
                        S1v = __builtin_elementwise_fma(hr, y, S1v);
                        SBv = __builtin_elementwise_fma(hv, y, SBv);
                    }
                }
                float S1 = S1v.x + S1v.y, SB = SBv.x + SBv.y;
                {
                    const float y = __builtin_amdgcn_rsqf(__builtin_fmaf(cl, ss, 1.0f));
                    S1 = __builtin_fmaf(al, y, S1);
                    SB = __builtin_fmaf(bl, y, SB);
                }
                const float c1 = __builtin_sqrtf(1.0f + ss);
                tt = c1 * SB;
                pp = s * vmi[ile] / c1;
                // the offset the ray found must be the source's (1e-4 relative: thousand times the roundoff); NaN / Inf
                // data, a stack whose fastest layer has no thickness and cannot reach R: the reference's sentinel
                const float miss = __builtin_fabsf(s * S1 - Rq);
                if (!(miss <= 1e-4f * Rq + 1e-3f) || !(tt == tt) || __builtin_isinf(tt))
                    tt = -999.0f;
                if (ilq == 0) {
                    tt = __builtin_sqrtf(zq * zq + Rq * Rq) * diag[0].z;
                    pp = __builtin_nanf("");
                }
            }
            if (valid) {
                const int loc = (int)(rec & 0xFFFFu);
                tout[loc] = tt;
                pout[loc] = Rs < 0.0f ? -pp : pp;
            }
        }
        F32F_STAMP(3);  // wave 0's own solves
        __syncthreads();
        F32F_STAMP(4);  // ... waiting for the other waves
        // ---- results back in input order: coalesced stores ----------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < F32F_PER_THREAD; ++j) {
            const int loc = j * F32F_BLOCK + threadIdx.x;
            const int64_t k = base + loc;
            if (k < lim) {
                a.d_timeP[k] = tout[loc];
                if (a.d_p_final)
                    a.d_p_final[k] = pout[loc];
            }
        }
        __syncthreads();  // before the next tile reuses the staging arrays
        F32F_STAMP(5);  // stores issued
    }
#ifdef F32F_CLOCKS
    if (threadIdx.x == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 8; ++i)
            f32f_clk[blockIdx.x * 8 + i] += clk_acc_[i];
#endif
}

}  // namespace rtmi

extern "C" size_t rtmi_f32fast_lds_bytes(int nlayers)
{
    const size_t W = (size_t)nlayers + 1;
    const size_t WS = (W + 3) & ~(size_t)3;
    size_t b = W * (WS / 2) * (sizeof(float4) + sizeof(float2)) + 2 * W * sizeof(float4) + ((W + 3) & ~(size_t)3) * sizeof(float) + (((size_t)nlayers + 3) & ~(size_t)3) * sizeof(float);
    b += (size_t)4 * F32F_TILE * sizeof(float);                 // sz, sR, tout, pout
    b += ((W + 3) & ~(size_t)3) * sizeof(uint32_t);             // hist
    b += (size_t)F32F_TILE * sizeof(uint32_t);                  // sidx
    b = (b + 7) & ~(size_t)7;
    b += 5 * W * sizeof(double);                                // vmaxd, ivm, ivm2, ivl, hvd
    return b + 16;
}

extern "C" hipError_t rtmi_launch_f32fast(const rtmi::F32FastArgs *args, int num_cus, hipStream_t stream)
{
    using namespace rtmi;
    if (!args || args->nlayers < 1 || args->nlayers > RTMI_F32FAST_MAX_NL)
        return hipErrorInvalidValue;
    const size_t lds = rtmi_f32fast_lds_bytes(args->nlayers);
    static bool attr_set = false;
    if (!attr_set) {
        // (the kernel also has 256 bytes of static LDS -- __syncthreads_or's -- which count against the CU's 160 KiB)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&f32fast_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        if (e != hipSuccess) {
            if (getenv("RTMI_VERBOSE"))
                fprintf(stderr, "rtmi_launch_f32fast: hipFuncSetAttribute: %s\n", hipGetErrorString(e));
            return e;
        }
        attr_set = true;
    }
    const int64_t ntiles = (args->nsrc + F32F_TILE - 1) / F32F_TILE;
    // as many workgroups per CU as their LDS allows (at most four: 16 waves), each walking tiles b, b + grid, ...
    int per_cu = (int)((size_t)159 * 1024 / (lds + 256));
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    int64_t grid = (int64_t)num_cus * per_cu;
    if (grid > ntiles)
        grid = ntiles;
    if (grid < 1)
        grid = 1;
    hipLaunchKernelGGL(f32fast_kernel, dim3((unsigned)grid), dim3(F32F_BLOCK), lds, stream, *args);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && getenv("RTMI_VERBOSE"))
        fprintf(stderr, "rtmi_launch_f32fast: grid %lld, lds %zu, nlayers %d: %s\n", (long long)grid, lds, args->nlayers, hipGetErrorString(e));
    return e;
}
