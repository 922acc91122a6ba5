// rtmi_pt.cpp -- native Metropolis-Hastings sweep of the reversible-jump sampler over the batched GPU likelihood.
//
// What the reference does per MCMC step and chain (one MPI rank each):
//   EXPLORE_MH_NOVARPAR   birth / death / perturbation of every (node, parameter)   prjmh_temper_rf.f90:637-761
//   EXPLORE_MH            sigma proposal                                            prjmh_temper_rf.f90:521-634
// with LOGLHOOD -> TraceRays behind every decision (loglhood.f90:128-199).  Here the chains of one process advance
// in lock-step: for every proposal slot the proposed models of all chains are packed into one pinned buffer and
// evaluated by ONE rtmi_loglhood_batch launch; proposals, bounds and accept/reject run on the host in this file.
// Proposal, prior, bound and acceptance rules are the reference's for NPL = 2, ENOS = 0, IPOIPR = 1, ICOV = 1,
// I_VARPAR = 0 (the test_1 configuration); each rule cites its source line below.  The temperature exchange between
// chains (possibly on other GPUs) stays with the caller (raytracerfortran_amd/tempering.py): it needs the collective.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rtmi.h"

namespace {

// xoshiro256** (public-domain algorithm by Blackman & Vigna): one independent stream per chain.
struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x)
    {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    void seed(uint64_t a, uint64_t b)
    {
        uint64_t x = a * 0xD1342543DE82EF95ull + b;
        for (auto &v : s)
            v = splitmix(x);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0, 1)
    double normal()
    {
        // Box-Muller (the reference's GASDEVJ is the same transform, prjmh_temper_rf.f90:2007)
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300)
            u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

struct Chain {
    int k = 1;
    std::vector<double> z, v;  // node depth / Vp, node 0 = surface node (depth pinned at 0)
    double sd = 0.01, logL = -INFINITY, ss = INFINITY, beta = 1.0;
    Rng rng;
};

}  // namespace

struct rtmi_pt {
    rtmi_pt_prior pr;
    int nchains, ld;
    int64_t ndata;
    std::vector<Chain> ch;
    // packed batch of proposals (host side of rtmi_loglhood_batch)
    std::vector<double> V, D, one, ll, ss;
    std::vector<int32_t> NL;
    std::vector<int> idx;
    std::vector<double> lp;
    std::vector<Chain> prop;
    int64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // acc/prop bd, acc/prop mh, acc/prop sd, launches, models
};

namespace {

const double LOG_2PI = 1.8378770664093453;

double log_pk(const rtmi_pt_prior &p, int k)  // Poisson prior on k (read_input.f90:78-81)
{
    return -p.lambda + k * std::log(p.lambda) - std::lgamma(k + 1.0);
}

double gaussian_loglike(double ss, int64_t n, double sd)  // loglhood.f90:194-196
{
    return -0.5 * (double)n * LOG_2PI - (ss / (2.0 * sd * sd) + (double)n * std::log(sd));
}

// nodes sorted by depth (stable): INTERPLAYER_novar, loglhood.f90:214-260
void sorted_order(const Chain &c, std::vector<int> &ord)
{
    ord.resize(c.k);
    for (int i = 0; i < c.k; ++i)
        ord[i] = i;
    for (int i = 1; i < c.k; ++i) {  // insertion sort: k <= 30
        const int o = ord[i];
        int j = i - 1;
        while (j >= 0 && c.z[ord[j]] > c.z[o]) {
            ord[j + 1] = ord[j];
            --j;
        }
        ord[j + 1] = o;
    }
}

// CHECKBOUNDS (prjmh_temper_rf.f90:1639-1679)
bool inside_bounds(const Chain &c, const rtmi_pt_prior &p, std::vector<int> &ord)
{
    for (int i = 0; i < c.k; ++i)
        if (c.v[i] < p.vmin || c.v[i] > p.vmax)
            return false;
    if (c.k > 1) {
        for (int i = 1; i < c.k; ++i)
            if (c.z[i] < 0.0 || c.z[i] > p.hmx)
                return false;
        sorted_order(c, ord);
        double prev = 0.0;
        for (int i = 1; i < c.k; ++i) {
            const double zi = c.z[ord[i]];
            if (zi - prev < p.hmin || zi > p.hmx)
                return false;
            prev = zi;
        }
    }
    return true;
}

// LOGLHOOD_RT's call arguments (loglhood.f90:130-144) into row m of the batch
void pack_model(rtmi_pt *h, int m, const Chain &c, std::vector<int> &ord)
{
    double *V = &h->V[(size_t)m * h->ld], *D = &h->D[(size_t)m * h->ld];
    if (c.k == 1) {
        V[0] = V[1] = c.v[0];
        D[0] = 9999.9;
        h->NL[m] = 1;
        return;
    }
    sorted_order(c, ord);
    for (int i = 0; i < c.k; ++i)
        V[i] = c.v[ord[i]];
    for (int i = 1; i < c.k; ++i)
        D[i - 1] = c.z[ord[i]];
    h->NL[m] = c.k - 1;
}

int evaluate_batch(rtmi_pt *h, int nm)
{
    if (nm == 0)
        return RTMI_OK;
    h->stats[6] += 1;
    h->stats[7] += nm;
    return rtmi_loglhood_batch(h->V.data(), h->D.data(), h->NL.data(), h->ld, nm, h->one.data(), h->ll.data(),
                               h->ss.data());
}

// MH rule (prjmh_temper_rf.f90:697-706, :742-751): accept unless u >= exp(logPr + beta (logL' - logL))
int decide(rtmi_pt *h, int kind)
{
    const int nm = (int)h->idx.size();
    std::vector<int> ord;
    for (int m = 0; m < nm; ++m)
        pack_model(h, m, h->prop[m], ord);
    const int rc = evaluate_batch(h, nm);
    if (rc)
        return rc;
    for (int m = 0; m < nm; ++m) {
        Chain &c = h->ch[h->idx[m]];
        const double logl_new = gaussian_loglike(h->ss[m], h->ndata, c.sd);
        double lr = h->lp[m] + (logl_new - c.logL) * c.beta;
        if (lr > 50.0)
            lr = 50.0;
        h->stats[2 * kind + 1] += 1;
        if (!(c.rng.uniform() >= std::exp(lr))) {
            c.k = h->prop[m].k;
            c.z = h->prop[m].z;
            c.v = h->prop[m].v;
            c.ss = h->ss[m];
            c.logL = logl_new;
            h->stats[2 * kind] += 1;
        }
    }
    return RTMI_OK;
}

void begin_slot(rtmi_pt *h)
{
    h->idx.clear();
    h->lp.clear();
    h->prop.clear();
}

int one_sweep(rtmi_pt *h)
{
    const rtmi_pt_prior &p = h->pr;
    std::vector<int> ord;
    // --- birth / death (prjmh_temper_rf.f90:669-713) ---
    begin_slot(h);
    if (p.kmin != p.kmax) {
        for (int i = 0; i < h->nchains; ++i) {
            Chain &c = h->ch[i];
            const double u = c.rng.uniform();
            int move = 0;
            if (c.k == p.kmax)
                move = (u <= 0.3333) ? 2 : 0;
            else if (c.k == p.kmin)
                move = (u <= 0.3333) ? 1 : 0;
            else
                move = (u <= 0.3333) ? 1 : ((u > 0.6666) ? 2 : 0);
            Chain q = c;
            double lp = 0.0;
            if (move == 1) {  // BIRTH_FULL (:997-1095): znew = maxpert(1)*u, Vp = minlim(2) + maxpert(2)*u
                q.z.push_back((p.hmx - p.hmin) * c.rng.uniform());
                q.v.push_back(p.vmin + (p.vmax - p.vmin) * c.rng.uniform());
                q.k = c.k + 1;
                lp = log_pk(p, c.k + 1) - log_pk(p, c.k);
            } else if (move == 2 && c.k > 1) {  // DEATH_FULL (:917-995): one of nodes 2..k
                const int idel = 1 + (int)(c.rng.uniform() * (c.k - 1));
                q.z.erase(q.z.begin() + idel);
                q.v.erase(q.v.begin() + idel);
                q.k = c.k - 1;
                lp = log_pk(p, c.k - 1) - log_pk(p, c.k);
            } else {
                continue;
            }
            if (inside_bounds(q, p, ord)) {
                h->idx.push_back(i);
                h->lp.push_back(lp);
                h->prop.push_back(std::move(q));
            } else {
                h->stats[1] += 1;  // rejected outside the prior (:707-711)
            }
        }
    }
    int rc = decide(h, 0);
    if (rc)
        return rc;
    // --- perturbation of every (node, parameter) in turn (:721-759), Cauchy step (PROPOSAL :1405) ---
    const double pert[2] = {(p.hmx - p.hmin) / p.pertsdsc, (p.vmax - p.vmin) / p.pertsdsc};
    int kmax_now = 0;
    for (const Chain &c : h->ch)
        kmax_now = c.k > kmax_now ? c.k : kmax_now;
    for (int ivo = 0; ivo < kmax_now; ++ivo) {
        for (int ipar = 0; ipar < 2; ++ipar) {
            if (ivo == 0 && ipar == 0)
                continue;  // the surface node's depth is not a parameter (:727)
            begin_slot(h);
            for (int i = 0; i < h->nchains; ++i) {
                Chain &c = h->ch[i];
                if (ivo >= c.k)
                    continue;
                Chain q = c;
                const double step = pert[ipar] * std::tan(3.141592653589793 * (c.rng.uniform() - 0.5));
                (ipar == 0 ? q.z[ivo] : q.v[ivo]) += step;
                if (inside_bounds(q, p, ord)) {
                    h->idx.push_back(i);
                    h->lp.push_back(0.0);
                    h->prop.push_back(std::move(q));
                } else {
                    h->stats[3] += 1;
                }
            }
            rc = decide(h, 1);
            if (rc)
                return rc;
        }
    }
    // --- sigma (EXPLORE_MH :521-575, PROPOSAL_SDRT :1616-1630); same rays: the residual sum is reused ---
    const double step = (p.sdmax - p.sdmin) / p.pertsdsdsc;
    for (Chain &c : h->ch) {
        if (c.rng.uniform() >= 0.10) {
            const double sd_new = c.sd + step * c.rng.normal();
            h->stats[5] += 1;
            if (sd_new >= p.sdmin && sd_new <= p.sdmax) {
                const double logl_new = gaussian_loglike(c.ss, h->ndata, sd_new);
                double lr = (logl_new - c.logL) * c.beta;
                if (lr > 50.0)
                    lr = 50.0;
                if (!(c.rng.uniform() >= std::exp(lr))) {
                    c.sd = sd_new;
                    c.logL = logl_new;
                    h->stats[4] += 1;
                }
            }
        }
    }
    return RTMI_OK;
}

}  // namespace

extern "C" {

rtmi_pt *rtmi_pt_create(int nchains, int ld_model, int64_t ndata, const rtmi_pt_prior *prior, uint64_t seed,
                        uint64_t stream_id)
{
    if (nchains < 1 || ld_model < 2 || ld_model - 1 > RTMI_MAX_LAYERS || !prior || ndata < 1 ||
        prior->kmax + 1 > ld_model || prior->kmin < 1 || prior->kmin > prior->kmax)
        return nullptr;
    rtmi_pt *h = new rtmi_pt;
    h->pr = *prior;
    h->nchains = nchains;
    h->ld = ld_model;
    h->ndata = ndata;
    h->ch.resize(nchains);
    for (int i = 0; i < nchains; ++i) {
        h->ch[i].z.assign(1, 0.0);
        h->ch[i].v.assign(1, 0.5 * (prior->vmin + prior->vmax));
        h->ch[i].rng.seed(seed, stream_id * 1000003ull + (uint64_t)i);
    }
    h->V.assign((size_t)nchains * ld_model, 0.0);
    h->D.assign((size_t)nchains * ld_model, 0.0);
    h->NL.assign(nchains, 1);
    h->one.assign(nchains, 1.0);
    h->ll.assign(nchains, 0.0);
    h->ss.assign(nchains, 0.0);
    return h;
}

void rtmi_pt_destroy(rtmi_pt *h) { delete h; }

int rtmi_pt_set_chain(rtmi_pt *h, int chain, int k, const double *depths, const double *vels, double sd, double beta)
{
    if (!h || chain < 0 || chain >= h->nchains || k < 1 || k + 1 > h->ld || !depths || !vels)
        return RTMI_ERR_BAD_ARG;
    Chain &c = h->ch[chain];
    c.k = k;
    c.z.assign(depths, depths + k);
    c.v.assign(vels, vels + k);
    c.sd = sd;
    c.beta = beta;
    return RTMI_OK;
}

int rtmi_pt_set_beta(rtmi_pt *h, const double *beta)
{
    if (!h || !beta)
        return RTMI_ERR_BAD_ARG;
    for (int i = 0; i < h->nchains; ++i)
        h->ch[i].beta = beta[i];
    return RTMI_OK;
}

int rtmi_pt_get_chain(const rtmi_pt *h, int chain, int *k, double *depths, double *vels, double *sd, double *logL,
                      double *ss)
{
    if (!h || chain < 0 || chain >= h->nchains)
        return RTMI_ERR_BAD_ARG;
    const Chain &c = h->ch[chain];
    if (k)
        *k = c.k;
    if (depths)
        memcpy(depths, c.z.data(), c.k * sizeof(double));
    if (vels)
        memcpy(vels, c.v.data(), c.k * sizeof(double));
    if (sd)
        *sd = c.sd;
    if (logL)
        *logL = c.logL;
    if (ss)
        *ss = c.ss;
    return RTMI_OK;
}

int rtmi_pt_evaluate(rtmi_pt *h)
{
    if (!h)
        return RTMI_ERR_BAD_ARG;
    std::vector<int> ord;
    for (int i = 0; i < h->nchains; ++i)
        pack_model(h, i, h->ch[i], ord);
    const int rc = evaluate_batch(h, h->nchains);
    if (rc)
        return rc;
    for (int i = 0; i < h->nchains; ++i) {
        h->ch[i].ss = h->ss[i];
        h->ch[i].logL = gaussian_loglike(h->ss[i], h->ndata, h->ch[i].sd);
    }
    return RTMI_OK;
}

int rtmi_pt_sweep(rtmi_pt *h, int nsweeps)
{
    if (!h || nsweeps < 0)
        return RTMI_ERR_BAD_ARG;
    for (int s = 0; s < nsweeps; ++s) {
        const int rc = one_sweep(h);
        if (rc)
            return rc;
    }
    return RTMI_OK;
}

int rtmi_pt_stats(const rtmi_pt *h, int64_t *out8)
{
    if (!h || !out8)
        return RTMI_ERR_BAD_ARG;
    memcpy(out8, h->stats, sizeof h->stats);
    return RTMI_OK;
}

}  // extern "C"
