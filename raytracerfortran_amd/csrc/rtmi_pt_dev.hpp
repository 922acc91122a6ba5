// rtmi_pt_dev.hpp -- device side of the tempering sampler that more than one translation unit needs: the handle's device
// view (PtDev) and the step of one chain (decide / sigma / propose), run by a group of lanes.
// Reference: EXPLORE_MH_NOVARPAR prjmh_temper_rf.f90:637-761, EXPLORE_MH :521-634; rules in rtmi_pt_rules.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/rtmi.h"
#include "rtmi_device.hpp"
#include "rtmi_kernels.h"
#include "rtmi_pt_rules.h"

namespace rtmi_pt_dev {

enum : int { PT_NONE = -1, PT_BD = 0, PT_MH = 1, PT_EVAL = 2 };

// Everything the kernels need; passed by value.  Pointers are device pointers owned by the handle.
struct PtDev {
    int nchains, K, ld_model, gx;
    int64_t ndata;
    rtmi_pt_prior pr;
    uint64_t seed;
    uint32_t stream_id;
    const double *logpk;  // [kmax + 2] Poisson prior, log
    // chain state
    int *k;
    double *z, *v;        // [nchains][K]
    double *sd, *logL, *ss, *beta;
    unsigned long long *ctr;
    // proposal of the current slot
    int *pk, *pactive;    // pactive: 0 none, 1 proposal awaiting its decision, 2 evaluation of the current state,
                          // 3 proposal that changes the model only below every source: its SS is the current one, nothing was launched
    double *pz, *pv, *plp;
    // batch rows of the likelihood launch and its per-workgroup residual sums
    double *V, *D;        // [nchains][ld_model]
    int32_t *NL;          // 0 = inactive row
    const double *partial;  // [nchains][gx] (the plain kernel's sums: small source sets)
    const unsigned long long *partial_x;  // [nchains][gx][4] or NULL: exact sums of the fused kernel's sampler form (Acc192)
    long long *stats;     // [nchains][8]: acc/prop birth-death, acc/prop perturbation, acc/prop sigma, models evaluated, -
    // incremental likelihood (rtmi_kernels.h::PtInc): what the step tells the likelihood about each model row's proposal
    rtmi::PtInc *inc;     // [nchains * nspec]
    int *cache_valid;     // [nchains] r2cur holds the residuals of the chain's current model (cleared by rtmi_pt_set_chain)
    double zmax_src;      // the deepest source of the handle's set
    int incremental;      // the incremental likelihood is on (RTMI_PT_INCREMENTAL)
    // the handle's sources are sorted by depth; a histogram of their depths bounds, with one load, the prefix of the set
    // that lies above a given depth: sources [0, cdf[bin(z)]) have depths < z (PtInc::n0, pn0)
    const int *cdf;       // [ncdf + 1] cdf[g] = sources whose bin is < g
    int ncdf;
    double cdf_z0, cdf_inv;  // bin(z) = clamp(floor((z - cdf_z0) * cdf_inv), 0, ncdf - 1)
    int64_t nsrc;
    // ---- speculative proposals (pt_step_spec below): nspec model rows per chain --------------------------------------
    // Rows m = chain * nspec + e, e < nspec: pk, pactive, plp, pz, pv, V, D, NL, inc and the residual sums are indexed by ROW
    // (nspec = 1: by chain, the non-speculative step).
    int nspec;
    int spec_step;        // the handle runs pt_step_spec (RTMI_PT_SPECULATE >= 1); 0 = the one-at-a-time step pt_step_group
    int spec_prop;        // 1: a window holds rows in proportion to the chain's slots (nspec at kmax nodes), see pt_step_spec
    unsigned long long *dbg;  // NULL, or [8] diagnostic (RTMI_PT_QSTATS=3): 100 MHz ticks of pt_step_spec's phases, summed over the calls
    int part_stride;      // residual sums of row m: partial_x[(m * part_stride + j) * 4], j < rowwg[m]
    int *cur;             // [nchains] the chain's next undecided slot of the current sweep (0 = birth/death, s = (node s>>1, parameter s&1))
    int *fin;             // [nchains] the sweep's sigma move is done: nothing left to do until the next sweep
    int *nwin;            // [nchains] proposals of the window that await their decisions (entries e < nwin); -1 = no window
    int *wend;            // [nchains] the slot after the window (where the chain goes on when every entry is rejected)
    unsigned long long *wctr;  // [nchains] ... and the position of its random stream there
    int *winv;            // [nchains] bit j: the window's slot wbase + j drew a proposal outside the prior; slots covered << 16
    int *pslot;           // [rows] the slot of entry e | (a trivial move was accepted on the path before it) << 16
    int *pinv;            // [rows] proposals outside the prior between the entry before and this one (same packing)
    unsigned long long *pctr;  // [rows] position of the random stream after the entry's proposal was drawn (= its acceptance draw)
    int *rowwg;           // [rows] workgroups that evaluate the row (= partial sums to add up)
    int *nact;            // [nchains] entries that go through the likelihood in this pass ...
    int *alist;           // [rows]   ... and which (alist[chain * nspec + a] = e of the a-th)
    int *wbuf;            // [nchains] which of the two sets of proposal rows (of the residual slab) the chain's rows write next
    // chain-level record of the pending accepted proposal: pend_i[4 c] = pending, pend_i[4 c + 1] = its PtPend record + 1
    double *wz, *wv;      // [nchains][K] the nodes at the end of the window's predicted path (pt_step_spec)
    int *pend_i;          // [nchains][4]: + [4 c + 2] the window's trivial moves (bit j: slot wbase + j; bit 16 + j: accepted),
                          // [4 c + 3] k at the end of the path
    rtmi::PtPend *pinfo;  // [2][nchains * nspec] what a later pass needs to know about an accepted proposal (rtmi_kernels.h)
    unsigned long long *acc;  // [rows][part_stride][4] = partial_x, writable: the step zeroes the accumulator of every row it proposes
    int gx_launch;        // per-slot engine: workgroups per row of the batched launch (rowwg of every evaluated row)
    int *walloc;          // [2][nchains] weights (slots of the chain's sweep, 2 k + 1) by which pt_sweep_prop_kernel deals its
                          // workgroups to the chains; an evaluation (rtmi_pt_evaluate) refreshes both halves
};

// Sources [0, prefix_above(d, z)) of the handle's depth-sorted set have depths < z: one load from the depth histogram (a
// multiple of 64, conservative; 0 when z is not above the shallowest bin, or NaN).  bin() is the host's formula (rtmi_pt.hip).
__device__ __forceinline__ int prefix_above(const PtDev &d, double z)
{
    if (!d.cdf || !(z > d.cdf_z0))
        return 0;
    const double t = (z - d.cdf_z0) * d.cdf_inv;
    const int g = t >= (double)(d.ncdf - 1) ? d.ncdf - 1 : (int)t;
    return d.cdf[g] & ~63;
}

#define PT_STEP_THREADS 256

// The sampler's step: [decide the previous slot] [sigma move] [propose the next slot], one GROUP of G lanes per chain
// (G = 16, 32 or 64 >= K, groups never straddle a wave).  Lane i of a group holds node i of its chain in registers;
// everything scalar (the residual sum, the likelihood, the random draws, the acceptance tests) is computed by all G
// lanes alike -- in SIMD that costs what one lane costs and needs no broadcast -- and everything per node (the bound
// checks, the depth order, packing the model row, committing an accepted proposal) is one step wide instead of a loop
// of dependent LDS round trips: the depth order is a rank count over lane shuffles (stable: ties keep node order,
// as the insertion sort of rtmi_pt_rules.h::sorted_order), CHECKBOUNDS' spacing test reads the predecessor in depth
// from the lane that holds rank - 1.  A one-thread-per-chain version of the same rules spent 60 % of its 12 us in the
// sort and the packing loops (profiles/r02k/pt_step_phases.txt).  The rules and the random streams are unchanged:
// chains advance bit for bit as before (tests/test_tempering.py compares with the host rules).
// pt_step_group<G>(d, c, ...): called by the G lanes of chain c's group (lane index within the wave = threadIdx.x & 63;
// the group starts at a multiple of G).  Used by pt_step_kernel (rtmi_pt.hip: one launch per slot) and by the persistent
// sweep kernel (rtmi_kernels.hip: the leading workgroup of a chain, between two likelihood evaluations).
template <int G>
__device__ __forceinline__ void pt_step_group(const PtDev &d, const int c, int decide, int sigma, int propose, int ivo, int ipar)
{
    namespace R = rtmi_pt_rules;
    const int lane = threadIdx.x & 63, gbase = lane & ~(G - 1);
    const int i = lane - gbase;               // the node this lane holds
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << gbase;
    const int K = d.K;
    const rtmi_pt_prior &p = d.pr;
    const bool lead = i == 0;
    const size_t row = (size_t)c * K + (i < K ? i : 0);
    // ---- one round trip to HBM: this lane's node (current and proposed), the chain's scalars, the Poisson table ----
    double zi = d.z[row], vi = d.v[row];
    const double pzi = d.pz[row], pvi = d.pv[row];
    const double lpk_lo = d.logpk[i < K + 2 ? i : K + 1], lpk_hi = d.logpk[i + G < K + 2 ? i + G : K + 1];
    int k = d.k[c];
    const int act = d.pactive[c];
    const int pk = d.pk[c];
    const unsigned long long nsw = d.ctr[c];  // sweeps this chain has completed: the random draws are indexed by (sweep, slot, purpose)
    const int pslot = d.pslot[c];             // the slot whose proposal awaits its decision
    double sd = d.sd[c], logL = d.logL[c], ss = d.ss[c];
    const double beta = d.beta[c], plp = d.plp[c];
    const rtmi::PtInc ic = d.inc[c];  // of the proposal that has just been evaluated (if any)
    int cache_ok = d.cache_valid[c];
    double part = 0.0;  // SUM(DresRT**2) over the workgroups' sums
    if (decide != PT_NONE && act == 3) {
        part = ss;  // no source is affected: every residual, hence the (exact) sum, is the current model's
    } else if (decide != PT_NONE && act != 0) {
        if (d.partial_x) {  // exact: any order gives the same words
            rtmi::Acc192 tot = rtmi::acc_zero();
            for (int j = 0; j < d.gx; ++j) {
                const unsigned long long *w = d.partial_x + 4 * ((size_t)c * d.gx + j);
                rtmi::Acc192 o;
                o.w0 = w[0];
                o.w1 = w[1];
                rtmi::acc_add(tot, o);
            }
            part = rtmi::acc_to_double(tot);
        } else {
            for (int j = 0; j < d.gx; ++j)  // in a fixed order
                part = part + d.partial[(size_t)c * d.gx + j];
        }
    }
    long long *st = d.stats + (size_t)c * 8;
    R::Philox rng = R::philox_make(d.seed, d.stream_id, (uint32_t)c, 0);

    // ---- decision of the slot whose likelihood has just been evaluated (prjmh_temper_rf.f90:697-706, :742-751) ----
    if (decide != PT_NONE && act != 0) {
        // a NaN likelihood (NaN data, an unusable model row) is the reference's -HUGE(1._RP): never accepted (loglhood.f90:197-200)
        double logl_new = R::gaussian_loglike(part, d.ndata, sd);
        if (logl_new != logl_new)
            logl_new = -1.7976931348623157e308;
        bool took = act == 2;  // the evaluated model is (now) the chain's current model
        if (act == 2) {
            ss = part;
            logL = logl_new;
        } else {
            double lr = plp + (logl_new - logL) * beta;
            if (lr > 50.0)
                lr = 50.0;
            if (lead)
                st[2 * decide + 1] += 1;
            const double u_acc = R::draw_uniform(rng, nsw, pslot, R::DRAW_ACCEPT);
            // accept unless u >= exp(logPr + beta (logL' - logL)); a NaN ratio (-HUGE against -HUGE) rejects
            if (lr == lr && !(u_acc >= exp(lr))) {
                k = pk;
                ss = part;
                logL = logl_new;
                if (lead) {
                    st[2 * decide] += 1;
                    d.k[c] = k;
                }
                if (i < k) {  // the proposal becomes the state
                    zi = pzi;
                    vi = pvi;
                    d.z[row] = zi;
                    d.v[row] = vi;
                }
                took = true;
            }
        }
        if (lead)
            d.pactive[c] = 0;
        if (lead && act != 3) {  // (act == 3: nothing was launched, nothing moved, no residuals of its own)
            // The launch that evaluated this model has moved the residuals of an earlier accepted proposal to r2cur.  If
            // this one is taken, ITS residuals (in r2prop, for the sources it affected) are the pending ones now; a model
            // that was evaluated in full makes the cache valid.
            rtmi::PtInc *o = d.inc + c;
            o->pending = took ? 1 : 0;
            if (took) {
                o->pzthr = ic.zthr;
                o->pdl1 = ic.dl1;
                o->pdl2 = ic.dl2;
                o->pn0 = ic.n0;
                if (ic.zthr == -INFINITY && !cache_ok)
                    d.cache_valid[c] = 1;
            }
        }
        if (took && act != 3 && ic.zthr == -INFINITY)
            cache_ok = 1;
    }

    // ---- sigma (EXPLORE_MH :521-575, PROPOSAL_SDRT :1616-1630); same rays, so the residual sum is reused ----------
    if (sigma) {
        if (R::draw_uniform(rng, nsw, R::SLOT_SIGMA, R::DRAW_MOVE) >= 0.10) {
            const double step = (p.sdmax - p.sdmin) / p.pertsdsdsc;
            const double sd_new = sd + step * R::draw_normal(rng, nsw, R::SLOT_SIGMA);
            if (lead)
                st[5] += 1;
            if (sd_new >= p.sdmin && sd_new <= p.sdmax) {
                double logl_new = R::gaussian_loglike(ss, d.ndata, sd_new);
                if (logl_new != logl_new)
                    logl_new = -1.7976931348623157e308;  // as above
                double lr = (logl_new - logL) * beta;
                if (lr > 50.0)
                    lr = 50.0;
                const double u_acc = R::draw_uniform(rng, nsw, R::SLOT_SIGMA, R::DRAW_ACCEPT);
                if (lr == lr && !(u_acc >= exp(lr))) {
                    sd = sd_new;
                    logL = logl_new;
                    if (lead)
                        st[4] += 1;
                }
            }
        }
    }
    if (lead) {
        d.sd[c] = sd;
        d.logL[c] = logL;
        d.ss[c] = ss;
    }
    if (propose == PT_NONE) {
        if (lead && sigma)
            d.ctr[c] = nsw + 1ull;  // (the step with the sigma move ends the chain's sweep)
        return;  // launch-uniform
    }
    const int slot = propose == PT_BD ? 0 : 2 * ivo + ipar;

    // ---- proposal of the next slot --------------------------------------------------------------------------------
    // Cross-lane steps (shuffles, ballots) are taken by every lane of the wave, outside the chain-dependent branches.
    const int up = lane + 1 < 64 ? lane + 1 : 63;
    const double z_up = __shfl(zi, up, 64), v_up = __shfl(vi, up, 64);  // node i + 1 (death shifts the nodes down)
    const int kc = k < 1 ? 1 : k;
    const double lpk_k = kc < G ? __shfl(lpk_lo, gbase + (kc & (G - 1)), 64) : __shfl(lpk_hi, gbase + ((kc - G) & (G - 1)), 64);
    const double lpk_kp = kc + 1 < G ? __shfl(lpk_lo, gbase + ((kc + 1) & (G - 1)), 64)
                                     : __shfl(lpk_hi, gbase + ((kc + 1 - G) & (G - 1)), 64);
    const double lpk_km = kc - 1 < G ? __shfl(lpk_lo, gbase + ((kc - 1) & (G - 1)), 64)
                                     : __shfl(lpk_hi, gbase + ((kc - 1 - G) & (G - 1)), 64);
    // Incremental likelihood: the proposal changes the model from depth zthr downwards (the new node, the removed node,
    // the shallower of a moved node's two depths, the top of the layer whose velocity changes); dl1 / dl2 = the last
    // interface of the current / the proposed model (9999.9 for a half-space, as LOGLHOOD_RT passes it).
    auto group_max = [&](double x) {
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1)
            x = fmax(x, __shfl_xor(x, off, 64));
        return x;
    };
    const double zi0 = zi;  // this lane's node depth before the move
    const double dl_cur = k == 1 ? 9999.9 : group_max((i >= 1 && i < k) ? zi : -INFINITY);
    const double zmin_cur = -group_max(i < k ? -zi : -INFINITY);  // the shallowest node: its velocity is the TOP layer's
    int moved_node = 0;     // death: the removed node; perturbation: the node; birth: the new node
    double zthr = -INFINITY;
    int qk = k;
    double lp = 0.0;
    bool have = false;
    if (propose == PT_BD) {
        // birth / death, 1/3 each, none at the k limits (prjmh_temper_rf.f90:669-713)
        if (p.kmin != p.kmax) {
            const double u = R::draw_uniform(rng, nsw, 0, R::DRAW_MOVE);
            int move;
            if (k == p.kmax)
                move = (u <= 0.3333) ? 2 : 0;
            else if (k == p.kmin)
                move = (u <= 0.3333) ? 1 : 0;
            else
                move = (u <= 0.3333) ? 1 : ((u > 0.6666) ? 2 : 0);
            if (move == 1) {  // BIRTH_FULL (:997-1095): znew = maxpert(1)*u, Vp = minlim(2) + maxpert(2)*u
                const double zn = (p.hmx - p.hmin) * R::draw_uniform(rng, nsw, 0, R::DRAW_A);
                const double vn = p.vmin + (p.vmax - p.vmin) * R::draw_uniform(rng, nsw, 0, R::DRAW_B);
                if (i == k) {
                    zi = zn;
                    vi = vn;
                }
                qk = k + 1;
                lp = lpk_kp - lpk_k;
                have = true;
                moved_node = k;
                zthr = zn;
            } else if (move == 2 && k > 1) {  // DEATH_FULL (:917-995): one of nodes 2..k
                int idel = 1 + (int)(R::draw_uniform(rng, nsw, 0, R::DRAW_A) * (k - 1));
                if (idel > k - 1)
                    idel = k - 1;
                if (i >= idel && i + 1 < k) {
                    zi = z_up;
                    vi = v_up;
                }
                qk = k - 1;
                lp = lpk_km - lpk_k;
                have = true;
                moved_node = idel;
                zthr = INFINITY;  // = the removed node's depth: taken from its lane below
            }
        }
    } else if (ivo < k) {
        // perturbation of (node ivo, parameter ipar), Cauchy step of scale (max - min)/pertsdsc (:721-759, PROPOSAL :1405)
        const double scale = (ipar == 0 ? (p.hmx - p.hmin) : (p.vmax - p.vmin)) / p.pertsdsc;
        const double step = scale * tan(3.141592653589793 * (R::draw_uniform(rng, nsw, slot, R::DRAW_MOVE) - 0.5));
        if (i == ivo) {
            if (ipar == 0)
                zi += step;
            else
                vi += step;
        }
        have = true;
        moved_node = ivo;
        zthr = INFINITY;      // = min(old, new depth) of the node (a velocity leaves the depth as it is): from its lane below
    }
    const double z_old_moved = __shfl(zi0, gbase + (moved_node & (G - 1)), 64);
    {
        const double z_new = __shfl(zi, gbase + (moved_node & (G - 1)), 64);
        if (zthr == INFINITY)
            zthr = (propose == PT_BD) ? z_old_moved : fmin(z_old_moved, z_new);
        if (!cache_ok || !(zthr == zthr))
            zthr = -INFINITY;  // no valid cache yet (or a NaN depth): evaluate in full
    }
    // CHECKBOUNDS (prjmh_temper_rf.f90:1639-1679; rtmi_pt_rules.h::inside_bounds) over the qk proposed nodes
    const bool in = have && i < qk;
    bool bad = in && (vi < p.vmin || vi > p.vmax);
    bad = bad || (in && qk > 1 && i >= 1 && (zi < 0.0 || zi > p.hmx));
    int r = 0;  // rank of this node in depth (stable)
    for (int j = 0; j < K; ++j) {
        const double zj = __shfl(zi, gbase + (j & (G - 1)), 64);
        r += (in && j < qk && (zj < zi || (zj == zi && j < i))) ? 1 : 0;
    }
    double prev = 0.0;  // the node above in depth; the rule measures the second node's spacing from 0
    for (int j = 0; j < K; ++j) {
        const double zj = __shfl(zi, gbase + (j & (G - 1)), 64);
        const int rj = __shfl(r, gbase + (j & (G - 1)), 64);
        if (j < qk && r >= 2 && rj == r - 1)
            prev = zj;
    }
    bad = bad || (in && qk > 1 && r >= 1 && (zi - prev < p.hmin || zi > p.hmx));
    // The node of rank 0 gives the top layer its velocity and its depth is not an interface (pack_model): a move that
    // involves the shallowest node of the current model (its removal, its velocity, its depth) or makes a node the
    // shallowest of the proposed one changes the model from the surface downwards, whatever that node's depth is --
    // with the usual node 0 at depth 0 this is zthr = 0 anyway; a chain set up with its first node deeper, or with a node
    // above it, gets a full evaluation instead of a cache that is silently wrong.
    {
        const int r_moved = __shfl(r, gbase + (moved_node & (G - 1)), 64);  // (birth, perturbation: rank in the proposed model)
        const bool removed = propose == PT_BD && qk < k;
        const bool touches_top = have && ((moved_node < k && z_old_moved <= zmin_cur) || (!removed && r_moved == 0));
        if (touches_top)
            zthr = -INFINITY;
    }
    const double dl_all = group_max((in && i >= 1) ? zi : -INFINITY);
    const double dl_new = qk == 1 ? 9999.9 : dl_all;
    const bool ok = have && (__ballot(bad) & gmask) == 0ull;
    // A proposal that changes the model only below the deepest source (and leaves both models' last interfaces below it:
    // whichLayer's special case) affects no source: its sum of squared residuals IS the current one, bit for bit (the sums
    // are exact), and no likelihood is launched for it -- about a third of the prior's nodes lie below the data.
    const bool trivial = d.incremental && cache_ok && zthr > d.zmax_src && dl_cur > d.zmax_src && dl_new > d.zmax_src;
    int nl = 0;
    if (ok) {
        // LOGLHOOD_RT's call arguments (rtmi_pt_rules.h::pack_model) and the proposal awaiting its decision
        double *Vr = d.V + (size_t)c * d.ld_model, *Dr = d.D + (size_t)c * d.ld_model;
        if (qk == 1) {
            if (lead) {
                Vr[0] = Vr[1] = vi;
                Dr[0] = 9999.9;
            }
            nl = 1;
        } else {
            if (in) {
                Vr[r] = vi;
                if (r >= 1)
                    Dr[r - 1] = zi;
            }
            nl = qk - 1;
        }
        if (in) {
            d.pz[row] = zi;
            d.pv[row] = vi;
        }
        if (lead) {
            d.pk[c] = qk;
            d.plp[c] = lp;
            d.pactive[c] = trivial ? 3 : 1;
            st[6] += 1;  // a model that goes through the likelihood
            if (!trivial) {
                rtmi::PtInc *o = d.inc + c;
                o->zthr = zthr;
                o->dl1 = dl_cur;
                o->dl2 = dl_new;
                o->n0 = prefix_above(d, fmin(zthr, fmin(dl_cur, dl_new)));  // (the rows of the residual slab never change here)
            }
        }
        if (trivial)
            nl = 0;  // an inactive row: the workgroups of this chain have nothing to do in this slot
    } else if (have && lead) {
        st[2 * propose + 1] += 1;  // proposed, rejected outside the prior (:707-711)
    }
    if (lead) {
        d.NL[c] = nl;
        d.pslot[c] = slot;
    }
}


// =====================================================================================================================
// The SPECULATIVE step (exact: chains advance bit for bit as with pt_step_group).
//
// The chain's draws are indexed by (sweep, slot, purpose) (rtmi_pt_rules.h), so what a chain does in the slots s, s + 1, ...
// can be worked out ahead of the likelihoods along a PREDICTED PATH:
//   * a proposal outside the prior is rejected whatever happens (no model row, :707-711);
//   * a proposal that changes the model only below the deepest source ("trivial": a node down there moved, was born or
//     died) leaves every residual, hence SS, as it is -- its acceptance ratio is the prior's alone, its acceptance draw is
//     known, so its OUTCOME IS KNOWN when it is drawn: it is decided on the spot, and the path goes on from the state it
//     leaves.  On BASELINE config 3 these are one slot in six, nearly all of them accepted -- three quarters of everything a
//     cold chain accepts (profiles/r04u/acc_by_level.txt) -- and up to round 4's first form each of them ended the window;
//   * a proposal that changes travel times needs its likelihood: it gets a model row, and the path ASSUMES IT IS REJECTED
//     (a cold chain accepts 3 % of them, the hottest 11 %).
// One step therefore lays out a WINDOW of up to PT_WIN_MAX slots holding up to `nspec` evaluated proposals (rows
// m = chain * nspec + e), one likelihood pass evaluates the rows side by side, and the next step takes their decisions in
// slot order with the very acceptance draws the one-at-a-time sampler would take: the first accepted entry becomes the
// state (it was drawn from the path's state at its slot, so it carries the trivial moves before it), everything behind it --
// rows, trivial moves, proposals outside the prior -- is void (drawn from a state that no longer is the chain's) and the chain
// goes on from the slot after it; if none is accepted the state at the END of the path becomes the chain's and the whole
// window is retired.  Same draws, same sums (they are exact integers), same decisions: prjmh_temper_rf.f90:721-759
// (sequential perturbations), :669-713 (birth / death), :561-569 (sigma) are run as written, only the likelihoods are
// evaluated ahead of time.  Statistics are taken when the window is decided, for the slots up to the accepted one.
//
// ONE WAVE PER CHAIN, SLOTS SIDE BY SIDE: the wave's 64 lanes are 64 / G groups of G lanes (G = 16, 32, 64 >= K); every
// group holds the path's nodes (lane i of a group = node i) and works on a slot of its own -- proposal, bounds, depth order,
// model row when the window is made; likelihood, acceptance draw, ratio when it comes back.  A round of the groups takes the
// next 64 / G slots from the path's state; the round is cut after the first group whose trivial move is accepted (the groups
// behind it drew from the wrong state: they go again in the next round) and before the group that would need one row more
// than the window has.  A proposal costs a lone wave about 6 us of dependent instructions whatever the number of groups at
// work (profiles/r04a/queue_v3_timings.txt).
//
// Per chain: `cur` = its next undecided slot (0 = birth/death, s = (node s >> 1, parameter s & 1); a chain with k nodes has
// the slots 0 .. 2k - 1, then the sigma move), `fin` = the sweep is complete; of the window: wend (the slot after it), winv
// (bit j: slot wbase + j drew a proposal outside the prior; slots covered << 16), pend_i[4 c + 2] (bit j: slot wbase + j was
// a trivial proposal; bit 16 + j: and accepted), pend_i[4 c + 3] (k at the end of the path) and wz / wv (its nodes), which
// count only when a trivial move was accepted.  Per row: pslot = slot | (a trivial move was accepted before it) << 16.
// Incremental likelihood: a trivial move changes no residual, so the rows of a window all read the chain's current
// residuals (r2cur, or the pending accepted proposal's row) and write rows of their own; the two sets of proposal rows
// alternate whenever a proposal is accepted, so no row is read and written in the same pass, and exactly one row of a pass
// (the first) moves the pending residuals to r2cur.
// begin: first step of a sweep (2: and the chain's temperature may still change, see below).  eval: decide an evaluation of the
// current state (rtmi_pt_evaluate) and propose nothing.
// wg_total: > 0 workgroups the chain's rows share in the next pass (they are dealt to the evaluated rows round-robin),
// 0 = every row has d.gx_launch of its own (per-slot engine: one batched launch over all rows).
// Called by ALL 64 lanes of a wave for one chain.
#define PT_SPEC_MAX 8    // evaluated rows of a window
#define PT_WIN_MAX 16    // slots of a window

template <int G>
__device__ __forceinline__ bool pt_step_spec(const PtDev &d, const int c, const int begin, const int eval, const int wg_total,
                                             const double beta_alt = __builtin_nan(""))
{
    namespace R = rtmi_pt_rules;
    constexpr int NG = 64 / G;                              // slots worked on side by side
    constexpr int NR = (PT_SPEC_MAX + NG - 1) / NG;         // rounds the decisions of a window can take
    const int lane = threadIdx.x & 63, gbase = lane & ~(G - 1);
    const int i = lane - gbase;               // the node this lane holds
    const int g = lane / G;                   // this lane's group
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << gbase;
    // one bit per group: its leading lane
    unsigned long long leads = 0ull;
#pragma unroll
    for (int q = 0; q < NG; ++q)
        leads |= 1ull << (q * G);
    const int K = d.K, D = d.nspec;
    const rtmi_pt_prior &p = d.pr;
    const bool lead = i == 0;
    const int ic_ = i < K ? i : 0;
    const size_t row = (size_t)c * K + ic_;
    const int m0 = c * D;                     // the chain's first model row
    // ---- one round trip to HBM: the chain, its window, the Poisson table -----------------------------------------------
    double zi = d.z[row], vi = d.v[row];
    const double wzi = d.wz[row], wvi = d.wv[row];  // the end of the window's path (counts when a trivial move was accepted)
    const double lpk_lo = d.logpk[i < K + 2 ? i : K + 1], lpk_hi = d.logpk[i + G < K + 2 ? i + G : K + 1];
    int k = d.k[c];
    unsigned long long nsw = d.ctr[c];        // sweeps completed: the draws are indexed by (sweep, slot, purpose)
    double sd = d.sd[c], logL = d.logL[c], ss = d.ss[c];
    const double beta = d.beta[c];
    int cache_ok = d.cache_valid[c];
    int cur = begin ? 0 : d.cur[c];
    int fin = begin ? 0 : d.fin[c];
    const int nwin = begin ? -1 : d.nwin[c];  // -1: no window awaits its decisions; 0: one without rows (its other slots still count)
    const int wend = d.wend[c], winv = d.winv[c], wtr = d.pend_i[4 * c + 2], wk = d.pend_i[4 * c + 3];
    const int nact_prev = d.nact[c];
    int wbuf = d.wbuf[c];
    int pending = d.pend_i[4 * c + 0], pend_idx = d.pend_i[4 * c + 1];
    // this group's entries of the window (one per round): everything the decision needs
    int e_act[NR], e_pk[NR], e_slot[NR];
    double e_lp[NR], e_pz[NR], e_pv[NR], e_part[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int e = r * NG + g;
        const bool on = e < D && e < nwin;
        const int m = m0 + (on ? e : 0);
        e_act[r] = on ? d.pactive[m] : 0;
        e_pk[r] = on ? d.pk[m] : 0;
        e_slot[r] = on ? d.pslot[m] : 0;
        e_lp[r] = on ? d.plp[m] : 0.0;
        e_pz[r] = on ? d.pz[(size_t)m * K + ic_] : 0.0;
        e_pv[r] = on ? d.pv[(size_t)m * K + ic_] : 0.0;
        // the row's exact sum: three limbs + flags, added up by the row's workgroups with 64-bit atomics
        const unsigned long long *w = d.partial_x + 4 * (size_t)m * d.part_stride;
        const unsigned long long a0 = on ? w[0] : 0ull, a1 = on ? w[1] : 0ull, a2 = on ? w[2] : 0ull, a3 = on ? w[3] : 0ull;
        rtmi::Acc192 tot;
        tot.w0 = a0 + (a1 << 48);
        tot.w1 = ((a1 >> 16) + (a2 << 32) + (tot.w0 < a0 ? 1ull : 0ull)) | (a3 & RTMI_ACC_FLAGS);
        e_part[r] = rtmi::acc_to_double(tot);
    }
    long long *st = d.stats + (size_t)c * 8;
    unsigned long long tk0 = 0ull, tk1 = 0ull;
    if (d.dbg) {
        tk0 = __builtin_amdgcn_s_memrealtime();
        if (e_part[0] + zi + wzi + sd + lpk_lo == 1.2345e301)  // (waits for the loads)
            tk0 += 1ull;
        tk1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0)
            atomicAdd(d.dbg + 0, tk1 - tk0);   // phase 0: the loads' round trip (from the first use)
    }
    R::Philox rng = R::philox_make(d.seed, d.stream_id, (uint32_t)c, 0);
    // what an accepted trivial move leaves as logL: the likelihood of the unchanged SS, as the one-at-a-time step computes it
    auto loglike_of_ss = [&](double s_, double sd_) {
        double l = R::gaussian_loglike(s_, d.ndata, sd_);
        if (l != l)
            l = -1.7976931348623157e308;  // a NaN likelihood is the reference's -HUGE(1._RP): never accepted (loglhood.f90:197-200)
        return l;
    };
    const double lstar_prev = loglike_of_ss(ss, sd);

    // beta_alt (not NaN): the chain's exchange of the sweep before is posted, not completed (rtmi_kernels.hip::prop_lead): its
    // temperature is `beta` or `beta_alt`.  The step goes on WITHOUT the partner as long as every decision it takes comes out the
    // same under both -- a proposal that is rejected by e^-30 under one temperature is rejected under its neighbour's as well --
    // and otherwise writes nothing and returns true: the caller completes the exchange and calls again.
    const bool two_betas = beta_alt == beta_alt;
    bool diverged = false;
    bool write_cache_valid = false, write_state = false;   // (what the decisions change is written after that check)
    // ---- decisions of the window (prjmh_temper_rf.f90:697-706, :742-751): every group its entry, then the first in order ---
    bool taken = false;
    int acc_slot = 1 << 30;   // slot of the accepted entry (the window's other slots count up to it)
    int n_eval_upto = 0;      // evaluated entries up to and including the accepted one
    long long d_st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int e = r * NG + g;
        const bool mine = !taken && e < D && e < nwin;      // (group-uniform)
        const int act = e_act[r];
        const int eslot = e_slot[r] & 0xFFFF;
        const bool after = (e_slot[r] >> 16) != 0;          // a trivial move was accepted on the path before this entry
        const double part = act == 3 ? ss : e_part[r];      // (3: no source is affected -- the sum IS the current one)
        const double logl_new = loglike_of_ss(part, sd);
        double lr = e_lp[r] + (logl_new - (after ? lstar_prev : logL)) * beta;
        if (lr > 50.0)
            lr = 50.0;
        const double u_acc = R::draw_uniform(rng, nsw, eslot, R::DRAW_ACCEPT);
        // accept unless u >= exp(logPr + beta (logL' - logL)); a NaN ratio (-HUGE against -HUGE) rejects; an evaluation
        // of the current state (2) is always "taken"
        const bool acc = mine && (act == 2 || (lr == lr && !(u_acc >= exp(lr))));
        const unsigned long long macc = __ballot(acc) & leads;
        const int gfirst = macc ? (__ffsll((long long)macc) - 1) / G : NG;   // the first group (= entry, = slot) that accepts
        if (two_betas) {  // the same under the other temperature the chain may turn out to have?
            double lr2 = e_lp[r] + (logl_new - (after ? lstar_prev : logL)) * beta_alt;
            if (lr2 > 50.0)
                lr2 = 50.0;
            const bool acc2 = mine && (act == 2 || (lr2 == lr2 && !(u_acc >= exp(lr2))));
            const unsigned long long macc2 = __ballot(acc2) & leads;
            const int gfirst2 = macc2 ? (__ffsll((long long)macc2) - 1) / G : NG;
            diverged = diverged || gfirst2 != gfirst;
        }
        // entries decided in this round: the groups up to the accepting one
        const bool decided = mine && g <= gfirst;
        const unsigned long long mdec = __ballot(decided && act != 2) & leads;
        const unsigned long long mdec_bd = __ballot(decided && act != 2 && eslot == 0) & leads;
        const unsigned long long mdec_ev = __ballot(decided && (act == 1 || act == 2)) & leads;
        d_st[1] += __popcll(mdec_bd);
        d_st[3] += __popcll(mdec) - __popcll(mdec_bd);
        d_st[6] += __popcll(mdec);   // models that went through the likelihood and were decided
        n_eval_upto += __popcll(mdec_ev);
        if (macc) {
            // the accepting group's entry becomes the state: its scalars and this lane's node from the lane that holds it
            const int src = gfirst * G;
            const int a_act = __shfl(act, src, 64), a_pk = __shfl(e_pk[r], src, 64), a_slot = __shfl(eslot, src, 64);
            const double a_part = __shfl(part, src, 64), a_logl = __shfl(logl_new, src, 64);
            const double a_pz = __shfl(e_pz[r], src + i, 64), a_pv = __shfl(e_pv[r], src + i, 64);
            ss = a_part;
            logL = a_logl;
            taken = true;
            if (a_act != 2) {
                k = a_pk;
                acc_slot = a_slot;
                cur = a_slot + 1;
                d_st[a_slot == 0 ? 0 : 2] += 1;
                if (i < k) {  // the proposal becomes the state
                    zi = a_pz;
                    vi = a_pv;
                }
            }
            if (a_act != 3) {
                // ITS residuals (in its row, for the sources it affected) are the pending ones now, and the chain's next
                // rows write the other set; a model evaluated in full makes the cache valid
                pending = 2;  // (set below, after the pass that evaluated this window is accounted for)
                pend_idx = (wbuf * d.nchains + c) * D + (r * NG + gfirst) + 1;  // its record was written when it was proposed
            }
        }
    }
    if (nwin >= 0) {
        // the window's slots without a row, up to the accepted slot: proposals outside the prior (:707-711) and trivial moves
        const int nsl = winv >> 16, wbase = wend - nsl;
        int upto = acc_slot - wbase;  // window slots that count
        upto = upto > nsl ? nsl : (upto < 0 ? 0 : upto);
        const unsigned int mcount = upto >= 32 ? 0xFFFFFFFFu : ((1u << upto) - 1u);
        const unsigned int binv = (unsigned)winv & 0xFFFFu & mcount, btr = (unsigned)wtr & 0xFFFFu & mcount,
                           bta = ((unsigned)wtr >> 16) & 0xFFFFu & mcount;
        const unsigned int bd0 = wbase == 0 ? 1u : 0u;  // window slot 0 is the birth/death slot
        d_st[1] += __popc((binv | btr) & bd0);
        d_st[3] += __popc((binv | btr) & ~bd0);
        d_st[6] += __popc(btr);      // (the one-at-a-time step counts a trivial proposal as a model that was decided)
        d_st[0] += __popc(bta & bd0);
        d_st[2] += __popc(bta & ~bd0);
        if (!taken) {
            cur = wend;  // the whole window is retired: the state at the end of its path is the chain's
            if (bta) {
                k = wk;
                if (i < k) {
                    zi = wzi;
                    vi = wvi;
                }
                logL = lstar_prev;
            }
        }
        d_st[7] += nact_prev - n_eval_upto > 0 && taken ? nact_prev - n_eval_upto : 0;  // evaluated for nothing
        // The pass that evaluated this window has moved the residuals of an earlier accepted proposal to r2cur (its mover row)
        if (nact_prev > 0 && pending == 1)
            pending = 0;
        if (pending == 2) {
            pending = 1;
            wbuf ^= 1;
            if (!cache_ok) {  // (without a valid cache every proposal is evaluated in full: this one was)
                cache_ok = 1;
                write_cache_valid = true;
            }
        }
        write_state = taken || bta != 0u;
    }

    if (d.dbg) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && (cur + k) != -12345)
            atomicAdd(d.dbg + 1, t - tk1);   // phase 1: the decisions
        tk1 = t;
    }
    // ---- sigma (EXPLORE_MH :521-575, PROPOSAL_SDRT :1616-1630) once every slot of the sweep is decided ------------------
    if (!eval && !fin && cur >= 2 * k) {
        if (R::draw_uniform(rng, nsw, R::SLOT_SIGMA, R::DRAW_MOVE) >= 0.10) {
            const double step = (p.sdmax - p.sdmin) / p.pertsdsdsc;
            const double sd_new = sd + step * R::draw_normal(rng, nsw, R::SLOT_SIGMA);
            d_st[5] += 1;
            if (sd_new >= p.sdmin && sd_new <= p.sdmax) {
                const double logl_new = loglike_of_ss(ss, sd_new);
                double lr = (logl_new - logL) * beta;
                if (lr > 50.0)
                    lr = 50.0;
                const double u_acc = R::draw_uniform(rng, nsw, R::SLOT_SIGMA, R::DRAW_ACCEPT);
                const bool acc_sd = lr == lr && !(u_acc >= exp(lr));
                if (two_betas) {
                    double lr2 = (logl_new - logL) * beta_alt;
                    if (lr2 > 50.0)
                        lr2 = 50.0;
                    diverged = diverged || acc_sd != (lr2 == lr2 && !(u_acc >= exp(lr2)));
                }
                if (acc_sd) {
                    sd = sd_new;
                    logL = logl_new;
                    d_st[4] += 1;
                }
            }
        }
        nsw += 1ull;  // the chain's sweep is complete
        fin = 1;
    }
    // a trivial move's ratio is lp + beta (L* - logL); logL is always the likelihood of (SS, sd) by this very formula, so the
    // difference is 0 and beta drops out -- checked, not assumed, while the temperature is open
    if (two_betas && !eval && !fin && !(loglike_of_ss(ss, sd) == logL))
        diverged = true;
    if (diverged)
        return true;   // nothing was written: the caller completes the exchange and calls again
    if (write_cache_valid && lane == 0)
        d.cache_valid[c] = 1;
    if (write_state && g == 0) {
        if (lead)
            d.k[c] = k;
        if (i < k && i < K) {
            d.z[row] = zi;
            d.v[row] = vi;
        }
    }
    if (lane == 0) {
        d.sd[c] = sd;
        d.logL[c] = logL;
        d.ss[c] = ss;
        d.ctr[c] = nsw;
        d.cur[c] = cur;
        d.fin[c] = fin;
        d.wbuf[c] = wbuf;
        d.pend_i[4 * c + 0] = pending;
        d.pend_i[4 * c + 1] = pend_idx;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (d_st[j])
                st[j] += d_st[j];
    }

    if (d.dbg) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        if (lane == 0)
            atomicAdd(d.dbg + 2, t - tk1);   // phase 2: sigma, the chain's words written
        tk1 = t;
    }
    // ---- the next window: rounds of the lane groups along the predicted path ------------------------------------------------
    // Cross-lane steps (shuffles, ballots) are taken by every lane of the wave, outside the group-dependent branches.
    double zi0 = zi, vi0 = vi;          // the path's state (starts as the chain's)
    int kp = k;
    double lpath = logL;                // ... and its logL
    const double lstar = loglike_of_ss(ss, sd);
    // begin == 2: the chain's temperature is not known yet (its exchange is posted, not completed: rtmi_kernels.hip::prop_lead).
    // Nothing of a window depends on it but a trivial move's ratio lp + beta (L* - logL) -- and at the start of a sweep
    // L* - logL is exactly 0 (logL was last set by this very formula), so beta drops out.  Should it not be (never seen), the
    // window is left to the next call, which comes after the exchange.
    const bool defer = begin == 2 && !(lstar == lpath);
    const bool propose_on = !eval && !fin && !defer;
    // Rows of this window.  A sweep lasts as long as the chain with the most nodes takes, a pass costs a fixed latency (the step
    // and the slowest sources of a share) on top of its throughput, and the launch's workgroups are dealt to the chains in
    // proportion to their slots: with windows in the same proportion every chain needs about the same NUMBER of passes and a
    // row gets about the same number of workgroups whatever its chain (nspec rows at kmax nodes; profiles/r04y/pt_spec.txt).
    int Dw = D;
    if (d.spec_prop && D > 2) {
        Dw = (D * (2 * k + 1) + 2 * p.kmax) / (2 * p.kmax + 1);
        Dw = Dw < 2 ? 2 : (Dw > D ? D : Dw);
    }
    int nslots = 0, nent = 0;           // slots the window covers, rows it holds
    unsigned int invbits = 0u, trivbits = 0u, taccbits = 0u;
    bool any_tacc = false;
    double my_zq = INFINITY;            // lane e < nent: the shallowest depth entry e changes (its prefix is looked up after the rounds)
    const int up = lane + 1 < 64 ? lane + 1 : 63;
    auto group_max = [&](double x) {
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1)
            x = fmax(x, __shfl_xor(x, off, 64));
        return x;
    };
#pragma unroll 1
    for (int round = 0; round < PT_WIN_MAX && propose_on; ++round) {
        int navail = 2 * kp - (cur + nslots);    // slots left in the path's sweep ...
        navail = navail > NG ? NG : navail;
        navail = navail > PT_WIN_MAX - nslots ? PT_WIN_MAX - nslots : navail;   // ... and in the window
        if (navail <= 0 || nent >= Dw)
            break;
        const double z_up = __shfl(zi0, up, 64), v_up = __shfl(vi0, up, 64);  // node i + 1 (death shifts the nodes down)
        const int kc = kp < 1 ? 1 : kp;
        const double lpk_k = kc < G ? __shfl(lpk_lo, gbase + (kc & (G - 1)), 64) : __shfl(lpk_hi, gbase + ((kc - G) & (G - 1)), 64);
        const double lpk_kp = kc + 1 < G ? __shfl(lpk_lo, gbase + ((kc + 1) & (G - 1)), 64)
                                         : __shfl(lpk_hi, gbase + ((kc + 1 - G) & (G - 1)), 64);
        const double lpk_km = kc - 1 < G ? __shfl(lpk_lo, gbase + ((kc - 1) & (G - 1)), 64)
                                         : __shfl(lpk_hi, gbase + ((kc - 1 - G) & (G - 1)), 64);
        const double dl_cur = kp == 1 ? 9999.9 : group_max((i >= 1 && i < kp) ? zi0 : -INFINITY);
        const double zmin_cur = -group_max(i < kp ? -zi0 : -INFINITY);  // the shallowest node: its velocity is the TOP layer's
        const int s = cur + nslots + g;  // this group's slot
        const bool on = g < navail;
        const int propose = s == 0 ? (int)PT_BD : (int)PT_MH;
        const int ivo = s >> 1, ipar = s & 1;
        zi = zi0;
        vi = vi0;
        int moved_node = 0;     // death: the removed node; perturbation: the node; birth: the new node
        double zthr = -INFINITY;
        int qk = kp;
        double lp = 0.0;
        bool have = false;
        if (on) {
            if (propose == PT_BD) {
                // birth / death, 1/3 each, none at the k limits (prjmh_temper_rf.f90:669-713)
                if (p.kmin != p.kmax) {
                    const double u = R::draw_uniform(rng, nsw, 0, R::DRAW_MOVE);
                    int move;
                    if (kp == p.kmax)
                        move = (u <= 0.3333) ? 2 : 0;
                    else if (kp == p.kmin)
                        move = (u <= 0.3333) ? 1 : 0;
                    else
                        move = (u <= 0.3333) ? 1 : ((u > 0.6666) ? 2 : 0);
                    if (move == 1) {  // BIRTH_FULL (:997-1095): znew = maxpert(1)*u, Vp = minlim(2) + maxpert(2)*u
                        const double zn = (p.hmx - p.hmin) * R::draw_uniform(rng, nsw, 0, R::DRAW_A);
                        const double vn = p.vmin + (p.vmax - p.vmin) * R::draw_uniform(rng, nsw, 0, R::DRAW_B);
                        if (i == kp) {
                            zi = zn;
                            vi = vn;
                        }
                        qk = kp + 1;
                        lp = lpk_kp - lpk_k;
                        have = true;
                        moved_node = kp;
                        zthr = zn;
                    } else if (move == 2 && kp > 1) {  // DEATH_FULL (:917-995): one of nodes 2..k
                        int idel = 1 + (int)(R::draw_uniform(rng, nsw, 0, R::DRAW_A) * (kp - 1));
                        if (idel > kp - 1)
                            idel = kp - 1;
                        if (i >= idel && i + 1 < kp) {
                            zi = z_up;
                            vi = v_up;
                        }
                        qk = kp - 1;
                        lp = lpk_km - lpk_k;
                        have = true;
                        moved_node = idel;
                        zthr = INFINITY;  // = the removed node's depth: taken from its lane below
                    }
                }
            } else {
                // perturbation of (node ivo, parameter ipar), Cauchy step of scale (max - min)/pertsdsc (:721-759, PROPOSAL :1405)
                const double scale = (ipar == 0 ? (p.hmx - p.hmin) : (p.vmax - p.vmin)) / p.pertsdsc;
                const double step = scale * tan(3.141592653589793 * (R::draw_uniform(rng, nsw, s, R::DRAW_MOVE) - 0.5));
                if (i == ivo) {
                    if (ipar == 0)
                        zi += step;
                    else
                        vi += step;
                }
                have = true;
                moved_node = ivo;
                zthr = INFINITY;      // = min(old, new depth) of the node (a velocity leaves the depth as it is): from its lane below
            }
        }
        const double z_old_moved = __shfl(zi0, gbase + (moved_node & (G - 1)), 64);
        {
            const double z_new = __shfl(zi, gbase + (moved_node & (G - 1)), 64);
            if (zthr == INFINITY)
                zthr = (propose == PT_BD) ? z_old_moved : fmin(z_old_moved, z_new);
            if (!cache_ok || !(zthr == zthr))
                zthr = -INFINITY;  // no valid cache yet (or a NaN depth): evaluate in full
        }
        // CHECKBOUNDS (prjmh_temper_rf.f90:1639-1679; rtmi_pt_rules.h::inside_bounds) over the qk proposed nodes
        const bool in = have && i < qk;
        bool bad = in && (vi < p.vmin || vi > p.vmax);
        bad = bad || (in && qk > 1 && i >= 1 && (zi < 0.0 || zi > p.hmx));
        int rk_ = 0;  // rank of this node in depth (stable)
        for (int q = 0; q < K; ++q) {
            const double zq = __shfl(zi, gbase + (q & (G - 1)), 64);
            rk_ += (in && q < qk && (zq < zi || (zq == zi && q < i))) ? 1 : 0;
        }
        double prev = 0.0;  // the node above in depth; the rule measures the second node's spacing from 0
        for (int q = 0; q < K; ++q) {
            const double zq = __shfl(zi, gbase + (q & (G - 1)), 64);
            const int rq = __shfl(rk_, gbase + (q & (G - 1)), 64);
            if (q < qk && rk_ >= 2 && rq == rk_ - 1)
                prev = zq;
        }
        bad = bad || (in && qk > 1 && rk_ >= 1 && (zi - prev < p.hmin || zi > p.hmx));
        // (see pt_step_group: a move that involves the shallowest node changes the model from the surface downwards)
        {
            const int r_moved = __shfl(rk_, gbase + (moved_node & (G - 1)), 64);
            const bool removed = propose == PT_BD && qk < kp;
            const bool touches_top = have && ((moved_node < kp && z_old_moved <= zmin_cur) || (!removed && r_moved == 0));
            if (touches_top)
                zthr = -INFINITY;
        }
        const double dl_all = group_max((in && i >= 1) ? zi : -INFINITY);
        const double dl_new = qk == 1 ? 9999.9 : dl_all;
        const bool ok = have && (__ballot(bad) & gmask) == 0ull;
        const bool trivial = ok && d.incremental && cache_ok && zthr > d.zmax_src && dl_cur > d.zmax_src && dl_new > d.zmax_src;
        // a trivial move is decided here (prjmh_temper_rf.f90:697-706, :742-751 with an unchanged SS): logL' is the likelihood
        // of the chain's SS as the one-at-a-time step would compute it after its empty pass
        bool tacc = false;
        if (trivial) {
            double lr = lp + (lstar - lpath) * beta;
            if (lr > 50.0)
                lr = 50.0;
            const double u_acc = R::draw_uniform(rng, nsw, s, R::DRAW_ACCEPT);
            tacc = lr == lr && !(u_acc >= exp(lr));
        }
        const unsigned long long mev = __ballot(ok && !trivial) & leads, mtriv = __ballot(trivial) & leads,
                                 mtacc = __ballot(tacc) & leads, minv = __ballot(have && !ok) & leads;
        // the groups that count, a prefix in slot order: up to the first accepted trivial move (the groups behind it drew from
        // a state the path has left), not beyond the window's rows
        const int gt = mtacc ? (__ffsll((long long)mtacc) - 1) / G : NG;
        int ncons = 0, nev = 0;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const bool isev = ((mev >> (q * G)) & 1ull) != 0ull;
            const bool can = q == ncons && q < navail && q <= gt && (!isev || nent + nev < Dw);
            ncons += can ? 1 : 0;
            nev += (can && isev) ? 1 : 0;
        }
        const bool consumed = g < ncons;
        const unsigned long long below = (1ull << gbase) - 1ull;
        const int e = nent + __popcll(mev & below);   // this group's row, if it holds an evaluated proposal
        const bool mine_ev = consumed && ok && !trivial;
#pragma unroll
        for (int q = 0; q < NG; ++q)
            if (q < ncons) {
                const unsigned int bit = 1u << (nslots + q);
                invbits |= ((minv >> (q * G)) & 1ull) ? bit : 0u;
                trivbits |= ((mtriv >> (q * G)) & 1ull) ? bit : 0u;
                taccbits |= ((mtacc >> (q * G)) & 1ull) ? bit : 0u;
            }
        if (mine_ev) {
            // LOGLHOOD_RT's call arguments (rtmi_pt_rules.h::pack_model) and the proposal awaiting its decision: row m0 + e
            const int m = m0 + e;
            double *Vr = d.V + (size_t)m * d.ld_model, *Dr = d.D + (size_t)m * d.ld_model;
            int nl;
            if (qk == 1) {
                if (lead) {
                    Vr[0] = Vr[1] = vi;
                    Dr[0] = 9999.9;
                }
                nl = 1;
            } else {
                if (in) {
                    Vr[rk_] = vi;
                    if (rk_ >= 1)
                        Dr[rk_ - 1] = zi;
                }
                nl = qk - 1;
            }
            if (in && i < K) {
                d.pz[(size_t)m * K + i] = zi;
                d.pv[(size_t)m * K + i] = vi;
            }
            if (lead) {
                d.pk[m] = qk;
                d.plp[m] = lp;
                d.pactive[m] = 1;
                d.pslot[m] = s | (any_tacc ? 1 << 16 : 0);
                d.NL[m] = nl;
                const int prow = d.nchains + (wbuf * d.nchains + c) * D + e;
                rtmi::PtInc *o = d.inc + m;
                o->zthr = zthr;
                o->dl1 = dl_cur;
                o->dl2 = dl_new;
                o->pending = pending;
                o->pend_idx = pend_idx;
                o->mover = e == 0 ? 1 : 0;
                o->cur_row = c;
                o->prop_row = prow;
                rtmi::PtPend *pp = d.pinfo + (size_t)(wbuf * d.nchains + c) * D + e;  // should it be accepted
                pp->zthr = zthr;
                pp->dl1 = dl_cur;
                pp->dl2 = dl_new;
                pp->prop_row = prow;
                unsigned long long *ac = d.acc + 4 * (size_t)m * d.part_stride;   // the row's sums start from zero
                ac[0] = ac[1] = ac[2] = ac[3] = 0ull;
                d.alist[m] = e;
            }
        }
        // lane e of the wave keeps what entry e's prefix lookup needs
        {
            const double zq = fmin(zthr, fmin(dl_cur, dl_new));
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const double zq_q = __shfl(zq, q * G, 64);
                const int e_q = nent + __popcll(mev & ((1ull << (q * G)) - 1ull));
                if (q < ncons && ((mev >> (q * G)) & 1ull) && lane == e_q)
                    my_zq = zq_q;
            }
        }
        // the path goes on from the state an accepted trivial move leaves (it is the last group that counts)
        if (gt < ncons) {
            const double nz = __shfl(zi, gt * G + i, 64), nv = __shfl(vi, gt * G + i, 64);
            kp = __shfl(qk, gt * G, 64);
            zi0 = nz;
            vi0 = nv;
            lpath = lstar;
            any_tacc = true;
        }
        nent += nev;
        nslots += ncons;
    }
    if (d.dbg) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && nent + nslots != -12345) {
            atomicAdd(d.dbg + 3, t - tk1);   // phase 3: the window's rounds
            atomicAdd(d.dbg + 6, 1ull);      // calls
            atomicAdd(d.dbg + 7, (unsigned long long)nslots);
        }
        tk1 = t;
    }
    // every row learns how many workgroups it gets and which prefix of the sorted sources lies above everything its proposal
    // changes; rows beyond the window are inactive; the end of the path is kept for the decision
    if (!eval) {
        if (lane < nent) {
            const int n0 = prefix_above(d, my_zq);
            int nw = d.gx_launch;                        // per-slot engine: every row has the batched launch's workgroups
            if (wg_total > 0) {                          // a chain's own workgroups, dealt round-robin to its rows
                nw = (wg_total - lane + nent - 1) / nent;
                nw = nw < 1 ? 1 : nw;                    // (fewer workgroups than rows: they take several rows each)
            }
            d.rowwg[m0 + lane] = nw;
            d.inc[m0 + lane].n0 = n0;
            d.pinfo[(size_t)(wbuf * d.nchains + c) * D + lane].n0 = n0;
        }
        if (lane < D && lane >= nent)
            d.NL[m0 + lane] = 0;
        if (any_tacc && g == 0 && i < kp && i < K) {
            d.wz[row] = zi0;
            d.wv[row] = vi0;
        }
        if (lane == 0) {
            d.nwin[c] = fin ? -1 : nent;
            d.wend[c] = cur + nslots;
            d.winv[c] = (int)(invbits | ((unsigned)nslots << 16));
            d.pend_i[4 * c + 2] = (int)(trivbits | (taccbits << 16));
            d.pend_i[4 * c + 3] = kp;
            d.nact[c] = nent;
        }
    } else if (lane == 0) {
        d.nwin[c] = -1;
        d.nact[c] = 0;
        d.walloc[c] = d.walloc[d.nchains + c] = 2 * k + 1;
    }
    if (d.dbg) {
        __threadfence();
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            atomicAdd(d.dbg + 4, t - tk1);   // phase 4: prefix lookups, the window's words, until the stores are out
            atomicAdd(d.dbg + 5, t - tk0);   // the whole call
        }
    }
    return false;
}

}  // namespace rtmi_pt_dev
