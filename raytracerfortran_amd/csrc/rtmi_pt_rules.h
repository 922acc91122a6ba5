// rtmi_pt_rules.h -- the rules of the reversible-jump tempering sampler as plain functions that compile for the host
// and for the device (the sweep kernels of rtmi_pt.hip run them on the GPU; the C ABI's host-side helpers and the CPU
// tests run the very same code).  Reference: ray_tracing_sampling/prjmh_temper_rf.f90, configuration NPL = 2,
// ENOS = 0, IPOIPR = 1, ICOV = 1, I_VARPAR = 0 (test_1); every rule cites its source line.
#ifndef RTMI_PT_RULES_H
#define RTMI_PT_RULES_H

#include <math.h>
#include <stdint.h>

#include "../../include/rtmi.h"

#if defined(__HIPCC__)
#define RTMI_HD __host__ __device__ inline
#else
#define RTMI_HD inline
#endif

#define RTMI_PT_MAX_K 64  // nodes per chain the sweep kernels hold in registers / scratch (kmax <= this)

namespace rtmi_pt_rules {

// ---- counter-based random numbers: Philox4x32-10 (Salmon et al., SC'11; public algorithm) ----------------------
// One stream per (seed, stream id, chain): key = seed, counter = (draw index lo, draw index hi, chain, stream id).
// A chain's state is its 64-bit draw index, kept in HBM next to the chain; the swap decisions use the stream
// (seed, 0xFFFFFFFF, round, pair), which every rank derives identically.
struct Philox {
    uint32_t key[2];
    uint32_t ctr_hi[2];  // words 2, 3 of the counter (chain, stream)
    uint64_t index;      // words 0, 1: the next draw
};

RTMI_HD void philox_round(uint32_t (&c)[4], uint32_t (&k)[2])
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
}

RTMI_HD Philox philox_make(uint64_t seed, uint32_t stream, uint32_t chain, uint64_t index)
{
    Philox p;
    p.key[0] = (uint32_t)seed;
    p.key[1] = (uint32_t)(seed >> 32);
    p.ctr_hi[0] = chain;
    p.ctr_hi[1] = stream;
    p.index = index;
    return p;
}

// uniform in [0, 1) with 53 random bits; advances the stream by one draw
RTMI_HD double philox_uniform(Philox &p)
{
    uint32_t c[4] = {(uint32_t)p.index, (uint32_t)(p.index >> 32), p.ctr_hi[0], p.ctr_hi[1]};
    uint32_t k[2] = {p.key[0], p.key[1]};
    for (int r = 0; r < 10; ++r)
        philox_round(c, k);
    ++p.index;
    const uint64_t bits = ((uint64_t)c[0] << 32) | c[1];
    return (double)(bits >> 11) * (1.0 / 9007199254740992.0);
}

// Box-Muller, as the reference's GASDEVJ (prjmh_temper_rf.f90:2007)
RTMI_HD double philox_normal(Philox &p)
{
    double u1 = philox_uniform(p);
    const double u2 = philox_uniform(p);
    if (u1 < 1e-300)
        u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

// ---- the sampler's draws, INDEXED BY WHAT THEY ARE FOR ---------------------------------------------------------
// A chain's draw is a pure function of (seed, stream, chain; sweep, slot, purpose): Philox counter word 1 = the number of
// the sweep, word 0 = slot * 8 + purpose, slot = 0 birth/death, 2 ivo + ipar for the perturbation of (node ivo, parameter
// ipar), SLOT_SIGMA for the sigma move.  No draw depends on how many draws were taken before it -- in particular not on
// whether the proposals of earlier slots fell inside the prior -- so the proposals of ALL the slots of a window are known at
// once and a group of lanes each can work them out side by side (rtmi_pt_dev.hpp::pt_step_spec).  [Rounds 1-3 took the
// draws from a running position of the chain's stream; the laws are the same, the streams are not.]
enum : int { DRAW_MOVE = 0, DRAW_A = 1, DRAW_B = 2, DRAW_ACCEPT = 3, DRAW_N1 = 4, DRAW_N2 = 5, SLOT_SIGMA = 255 };

RTMI_HD double draw_uniform(Philox &p, uint64_t sweep, int slot, int purpose)
{
    p.index = (sweep << 32) | (uint64_t)(uint32_t)(slot * 8 + purpose);
    return philox_uniform(p);
}

// Box-Muller from the two uniforms (slot, DRAW_N1), (slot, DRAW_N2), as philox_normal
RTMI_HD double draw_normal(Philox &p, uint64_t sweep, int slot)
{
    double u1 = draw_uniform(p, sweep, slot, DRAW_N1);
    const double u2 = draw_uniform(p, sweep, slot, DRAW_N2);
    if (u1 < 1e-300)
        u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

// ---- prior, likelihood, bounds ---------------------------------------------------------------------------------
RTMI_HD double log_pk(const rtmi_pt_prior &p, int k)  // Poisson prior on k (read_input.f90:78-81)
{
    return -p.lambda + k * log(p.lambda) - lgamma(k + 1.0);
}

RTMI_HD double gaussian_loglike(double ss, int64_t n, double sd)  // loglhood.f90:194-196, overflow-free normalisation
{
    return -0.5 * (double)n * 1.8378770664093453 - (ss / (2.0 * sd * sd) + (double)n * log(sd));
}

// nodes sorted by depth (stable insertion sort, k <= RTMI_PT_MAX_K): INTERPLAYER_novar, loglhood.f90:214-260
RTMI_HD void sorted_order(const double *z, int k, int *ord)
{
    for (int i = 0; i < k; ++i)
        ord[i] = i;
    for (int i = 1; i < k; ++i) {
        const int o = ord[i];
        int j = i - 1;
        while (j >= 0 && z[ord[j]] > z[o]) {
            ord[j + 1] = ord[j];
            --j;
        }
        ord[j + 1] = o;
    }
}

// CHECKBOUNDS (prjmh_temper_rf.f90:1639-1679); leaves the depth order in `ord` when k > 1
RTMI_HD bool inside_bounds(const double *z, const double *v, int k, const rtmi_pt_prior &p, int *ord)
{
    for (int i = 0; i < k; ++i)
        if (v[i] < p.vmin || v[i] > p.vmax)
            return false;
    if (k > 1) {
        for (int i = 1; i < k; ++i)
            if (z[i] < 0.0 || z[i] > p.hmx)
                return false;
        sorted_order(z, k, ord);
        double prev = 0.0;
        for (int i = 1; i < k; ++i) {
            const double zi = z[ord[i]];
            if (zi - prev < p.hmin || zi > p.hmx)
                return false;
            prev = zi;
        }
    }
    return true;
}

// LOGLHOOD_RT's call arguments (loglhood.f90:130-144) into one model row: V[0..k-1] velocities top -> down,
// D[0..k-2] interface depths; returns NLayers.  k = 1 is passed as (v, v) over one interface at 9999.9 m.
RTMI_HD int pack_model(const double *z, const double *v, int k, const int *ord, double *V, double *D)
{
    if (k == 1) {
        V[0] = V[1] = v[0];
        D[0] = 9999.9;
        return 1;
    }
    for (int i = 0; i < k; ++i)
        V[i] = v[ord[i]];
    for (int i = 1; i < k; ++i)
        D[i - 1] = z[ord[i]];
    return k - 1;
}

// ---- temperature exchange (TEMPSWP_MH, prjmh_temper_rf.f90:1329-1357) ------------------------------------------
// Every rank holds logL of ALL chains (one all-gather) and the replicated level[] (temperature level of every
// global chain).  Chains are paired by adjacent temperature level, alternating even and odd pairs from round to
// round, and each pair swaps its LEVELS with probability min(1, exp((beta2 - beta1)(logL1 - logL2))) -- the
// reference swaps the chain states and leaves beta with the worker (:1351-1357), which is the same move with no
// model crossing the fabric.  Deterministic: the random number of pair a in round r is the first draw of the Philox
// stream (seed, 0xFFFFFFFF, r, a), so every rank takes identical decisions without talking.  `order` is scratch of
// nchains ints.  Returns accepted swaps; *proposed += pairs tried.  The reference's pairing -- the first two workers
// that report to rank 0 (:331-349) -- depends on MPI timing and is deliberately not reproduced.
// The two steps of a swap round, usable one chain / one pair at a time (the device runs them with one thread each):
// position of chain i when the chains are ordered by (level, chain index) ...
RTMI_HD int swap_position(int nchains, const int *level, int i)
{
    int pos = 0;
    const int li = level[i];
    for (int j = 0; j < nchains; ++j)
        pos += (level[j] < li) || (level[j] == li && j < i);
    return pos;
}

// ... and the decision for the pair at positions (a, a + 1) of that order: true = the two chains swap their levels.
RTMI_HD bool swap_pair(int a, const int *order, const double *logl, const int *level, const double *beta_of_level,
                       uint64_t seed, uint64_t round)
{
    const int c1 = order[a], c2 = order[a + 1];
    const double b1 = beta_of_level[level[c1]], b2 = beta_of_level[level[c2]];
    Philox p = philox_make(seed, 0xFFFFFFFFu, (uint32_t)round, ((uint64_t)(round >> 32) << 32) | (uint32_t)a);
    const double u = philox_uniform(p);
    if (b1 == b2)
        return false;
    double lr = (b2 - b1) * (logl[c1] - logl[c2]);
    if (lr > 50.0)
        lr = 50.0;
    return u <= exp(lr);
}

RTMI_HD int swap_decide(int nchains, const double *logl, int *level, const double *beta_of_level, uint64_t seed,
                        uint64_t round, int *order, long long *proposed)
{
    for (int i = 0; i < nchains; ++i)
        order[swap_position(nchains, level, i)] = i;
    int acc = 0;
    for (int a = (int)(round & 1ull); a + 1 < nchains; a += 2) {  // the pairs are disjoint: any order of evaluation
        if (proposed)
            *proposed += 1;
        if (swap_pair(a, order, logl, level, beta_of_level, seed, round)) {
            const int c1 = order[a], c2 = order[a + 1];
            const int t = level[c1];
            level[c1] = level[c2];
            level[c2] = t;
            ++acc;
        }
    }
    return acc;
}

}  // namespace rtmi_pt_rules

#endif
