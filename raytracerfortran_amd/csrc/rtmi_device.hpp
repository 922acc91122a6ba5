// rtmi_device.hpp -- device-side building blocks shared by every kernel of rtmi_kernels.hip.
//
// Everything here restates /root/reference/subroutineR-quiet.f90 (module raymod) operation for
// operation in IEEE fp64; the translation unit MUST be built with -ffp-contract=off.
#ifndef RTMI_DEVICE_HPP
#define RTMI_DEVICE_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

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

// The layer table of one model in LDS: RTMI_TABLE_ARRAYS arrays of RTMI_TABLE_STRIDE(nlayers) elements; the pair
// array HW comes first (16-byte aligned for the fp64 build) and takes two of them.
#define RTMI_TABLE_ARRAYS 16  // + 3 (SA, SC, HC) for models of up to RTMI_SKIP_MAX_NL interfaces: RTMI_TABLE_NARRAYS
#define RTMI_TRIP_CLASSES 3  // classes of trip_class (the fused kernel's solve order puts the Newton-only class after them)
#ifndef RTMI_TERM_GROUP
#define RTMI_TERM_GROUP 3  // layer terms evaluated side by side in eval_terms' main loop
#endif
#ifndef RTMI_PREDICT
#define RTMI_PREDICT 1  // 0: every evaluation of solve() is made (A/B builds; see newton_lands_converged)
#endif
// Table forms of the fused kernel (a template parameter, chosen per launch from its largest model):
//   0  stride nlayers + 2, decided at run time                                   (any model)
//   1  row tables (below) and a CONSTANT stride for RTMI_ROWS_MAX_NL interfaces   (every model <= 20 interfaces)
//   2  a CONSTANT stride for RTMI_FIXED_MAX_NL interfaces, no row tables          (every model <= 32: the sampler's k <= 30)
// With a constant stride the arrays sit at COMPILE-TIME offsets from the table's base -- in the immediate offsets of the
// LDS instructions instead of a dozen scalar registers that the solve loop has not got (it spills them) -- and so does
// everything of fixed size that is laid out behind the table.
#define RTMI_FIXED_MAX_NL 32
#define RTMI_TABLE_FORM(max_nlayers) ((max_nlayers) <= RTMI_ROWS_MAX_NL ? 1 : ((max_nlayers) <= RTMI_FIXED_MAX_NL ? 2 : 0))
#define RTMI_TABLE_STRIDE_OF(nlayers, tf) ((tf) == 1 ? RTMI_ROWS_MAX_NL + 2 : ((tf) == 2 ? RTMI_FIXED_MAX_NL + 2 : (nlayers) + 2))
#define RTMI_TABLE_STRIDE(nlayers) ((nlayers) + 2)
// Models of up to RTMI_ROWS_MAX_NL interfaces also get the pair table once per stack height ("rows", see LayerTable);
// `rows` is decided per launch (from the largest model of the launch), so that the size is monotone in nlayers.
#define RTMI_ROWS_MAX_NL 20
#define RTMI_HINT_MAX_NL 512  // deeper models: no ordering of the bisection sources by expected evaluations
#define RTMI_SKIP_MAX_NL 64   // deeper models: no skipping of provably idle bisection trips (the bound's table is O(nlayers^2))
// `skiptab` (like `rows`) is decided per LAUNCH from its largest model, so that the size is monotone in nlayers
#define RTMI_TABLE_NARRAYS(skiptab) (RTMI_TABLE_ARRAYS + ((skiptab) ? 3 : 0))
#define RTMI_TABLE_ELEMS(nlayers, tf, skiptab) \
    (((size_t)RTMI_TABLE_NARRAYS(skiptab) * RTMI_TABLE_STRIDE_OF(nlayers, (int)(tf)) + 1) / 2 * 2 + \
     ((int)(tf) == 1 ? (size_t)2 * (RTMI_ROWS_MAX_NL + 1) * (RTMI_ROWS_MAX_NL + 1) : 0))
template <typename T>
struct alignas(2 * sizeof(T)) LayerPair {
    T hv, vv;
};
template <typename T>
struct LayerTable {
    LayerPair<T> *HW;  // HW[i] = {HV[i], VV[i]} i < nlayers: what one layer term reads; HW[nul] = {0, 0} is the
                       // "no layer" entry: its term is exactly 0 for every finite x
    int nul;           // = nlayers + 1
    LayerPair<T> *HWR; // rows (or nullptr): HWR[r * row_len + i] = HW[i] for i < r, {0, 0} for r <= i < row_len, one
                       // row per number r of fully crossed layers.  A lane reads its own row at a wave-uniform
                       // column, so the layer loop needs no per-lane index arithmetic at all.
    int row_len;       // = nlayers + 1
    T *D;     // D[i]    depth to interface i                       i < nlayers
    T *V;     // V[i]    velocity of layer i                        i <= nlayers
    T *H;     // H[i]    thickness of a fully crossed layer i       i < nlayers
    T *HV;    // HV[i]   H[i]*V[i]                                  i < nlayers
    T *VV;    // VV[i]   V[i]*V[i]                                  i <= nlayers
    T *PS;    // PS[i]   ((H0/V0 + H1/V1) + ...) + Hi/Vi            i < nlayers
    T *VMAX;  // VMAX[i] max(V[0..i])                               i <= nlayers
    T *VMI;   // VMI[i]  1/VMAX[i]                                  i <= nlayers
    T *Q;     // scratch: H[i]/V[i] while the table is built, then the index of the fastest layer of the stack 0..i
    T *PA;    // } scheduling hint only (trip_class): the offset the rest of the stack covers when the ray runs
    T *PC;    // } critical in its fastest layer, PA[il] + hl * PC[il], and that layer's thickness PH[il] (0: it
    T *PH;    // } is the source's own layer, thickness hl)                             1 <= il <= nlayers
    T *HVOK;  // HVOK[i] 1.0 if H and H*V of layers 0..i are 0 or of moderate magnitude and V of moderate magnitude
              // (fast-path guards), else 0.0
    T *PSA;   // PSA[i]  |HV[0]| + ... + |HV[i]|: bounds costFunc's sum at the bisection's first point (lane_setup)  i < nlayers
    T *SA;    // } bound of costFunc's sum near the critical ray parameter (bisection_skip): for the stack whose
    T *SC;    // } source layer is il, the full layers slower than its fastest contribute at most SA[il], the
    T *HC;    // } partial layer hl * SC[il] (0 if it is the fastest), the fastest layers HC[il] * x / sqrt(1 - x^2 vmax^2)
};

template <int TF = 0, typename T>
__device__ __forceinline__ LayerTable<T> carve_table(T *lds, int nlayers, bool rows, bool skiptab)
{
    constexpr bool ROWS = TF == 1;
    const int n1 = RTMI_TABLE_STRIDE_OF(nlayers, TF);  // forms 1 and 2: a constant
    if (TF != 0) {
        skiptab = true;  // RTMI_FIXED_MAX_NL <= RTMI_SKIP_MAX_NL: the launcher sets both (a constant for the offsets below)
        rows = ROWS;
    }
    LayerTable<T> t;
    t.HW = reinterpret_cast<LayerPair<T> *>(lds);
    t.nul = nlayers + 1;
    t.D = lds + 2 * n1;
    t.V = lds + 3 * n1;
    t.H = lds + 4 * n1;
    t.HV = lds + 5 * n1;
    t.VV = lds + 6 * n1;
    t.PS = lds + 7 * n1;
    t.VMAX = lds + 8 * n1;
    t.VMI = lds + 9 * n1;
    t.Q = lds + 10 * n1;
    t.HVOK = lds + 11 * n1;
    t.PA = lds + 12 * n1;
    t.PC = lds + 13 * n1;
    t.PH = lds + 14 * n1;
    t.PSA = lds + 15 * n1;
    t.SA = skiptab ? lds + 16 * n1 : nullptr;
    t.SC = skiptab ? lds + 17 * n1 : nullptr;
    t.HC = skiptab ? lds + 18 * n1 : nullptr;
    t.HWR = rows ? reinterpret_cast<LayerPair<T> *>(lds + ((size_t)RTMI_TABLE_NARRAYS(skiptab) * n1 + 1) / 2 * 2) : nullptr;
    t.row_len = ROWS ? RTMI_ROWS_MAX_NL + 1 : nlayers + 1;
    return t;
}

// the IEEE divide of the kernels (defined below, next to the guarded fast forms)
__device__ __forceinline__ double div_ieee(double n, double d);
__device__ __forceinline__ float div_ieee(float n, float d);

// v == 0 or |v| inside (1e-30, 1e30): safe as one factor of a numerator of div_guarded (the product of two
// such factors stays far from the exponent ranges where v_div_scale / v_div_fixup would act).
__device__ __forceinline__ bool mag_ok(float v)  // single precision: the same idea over a narrower exponent range
{
    const float a = fabsf(v);
    return v == 0.0f || (a > 1e-9f && a < 1e9f);
}
__device__ __forceinline__ bool mag_ok(double v)
{
    const double a = fabs(v);
    return v == 0.0 || (a > 1e-30 && a < 1e30);
}

__device__ __forceinline__ bool mag_nz(float v)
{
    const float a = fabsf(v);
    return a > 1e-9f && a < 1e9f;
}
__device__ __forceinline__ bool mag_nz(double v)  // as mag_ok, zero excluded: safe as a factor of a denominator
{
    const double a = fabs(v);
    return a > 1e-30 && a < 1e30;
}

// The value lane `j` (wave-uniform) holds, in every lane.
__device__ __forceinline__ double wave_readlane(double v, int j)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), j), __builtin_amdgcn_readlane(__double2loint(v), j));
}
__device__ __forceinline__ float wave_readlane(float v, int j)
{
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), j));
}

// Derive the table of one model (whole workgroup; ends with a barrier).  Roundings are the reference's:
//   H(i) = depths(i) - depths(i-1)            InsertLayer :67
//   (H*V) and (V*V) as separate products      costFunc :195, costFunc_Prime :214-219
//   sum(depths/vp) left to right              GetPTime :112
//   1/maxval(vp)                              :142, :148, :299, :388
template <bool WITH_HINT = false, typename T>
__device__ __forceinline__ void stage_model(const LayerTable<T> &t, const T *__restrict__ vels,
                                            const T *__restrict__ depths, int nlayers)
{
    for (int i = threadIdx.x; i <= nlayers; i += blockDim.x) {
        const T v = vels[i];
        t.V[i] = v;
        t.VV[i] = v * v;
        if (i < nlayers) {
            const T d = depths[i];
            const T h = (i == 0) ? d : d - depths[i - 1];
            t.D[i] = d;
            t.H[i] = h;
            t.HV[i] = h * v;
            t.Q[i] = div_ieee(h, v);
            t.HW[i].hv = h * v;
            t.HW[i].vv = v * v;
        }
    }
    if (threadIdx.x == 0) {
        t.HW[t.nul].hv = T(0.0);
        t.HW[t.nul].vv = T(0.0);
        t.H[t.nul] = T(0.0);
        t.V[t.nul] = T(1.0);
        t.VV[t.nul] = T(0.0);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        // The sequential part -- running maximum, left-to-right sum, running flags -- over the lanes of wave 0:
        // every lane loads its layer, then one scalar loop walks the lanes with readlane (no LDS round trip per
        // layer); the running values are the same in every lane, lane j keeps what they are after layer j.
        const int lane = threadIdx.x;
        T s = T(0.0), m = T(0.0), sa = T(0.0);
        int jm = 0;           // first layer that carries the maximum (for the scheduling hint below)
        bool okrun = true;    // H, H*V of layers 0..i are 0 or of moderate magnitude and V of moderate magnitude
        bool sorted = true;   // interfaces in non-decreasing order (no NaN): which_layer may search by bisection
        bool posv = true;     // every velocity > 0 (no NaN): with sorted interfaces and D[0] >= 0 no term of f, f' is negative
        for (int base = 0; base <= nlayers; base += 64) {
            const int i = base + lane;
            const bool in = i <= nlayers, inl = i < nlayers;
            const T v = in ? t.V[i] : T(0.0), q = inl ? t.Q[i] : T(0.0), ahv = inl ? fabs(t.HV[i]) : T(0.0);
            const bool ok_l = !inl || (mag_ok(t.HV[i]) && mag_ok(t.H[i]) && mag_nz(v));
            const bool so_l = !(inl && i >= 1) || (t.D[i] >= t.D[i - 1]);
            T vmax_l = T(0.0), ps_l = T(0.0), psa_l = T(0.0);
            int jm_l = 0;
            const int cnt = nlayers + 1 - base < 64 ? nlayers + 1 - base : 64;
            for (int j = 0; j < cnt; ++j) {
                const T vj = wave_readlane(v, j), qj = wave_readlane(q, j);
                sa = sa + wave_readlane(ahv, j);
                const int ij = base + j;
                if (ij == 0) {
                    m = vj;
                    s = qj;
                } else {
                    jm = (vj > m) ? ij : jm;
                    m = (vj > m) ? vj : m;  // maxval(): order-free
                    if (ij < nlayers)
                        s = s + qj;         // sum(depths/vp) left to right
                }
                if (lane == j) {
                    vmax_l = m;
                    ps_l = s;
                    psa_l = sa;
                    jm_l = jm;
                }
            }
            const unsigned long long bad = __ballot(!ok_l);
            const int firstbad = bad ? __ffsll((long long)bad) - 1 : 64;
            const bool hvok_l = okrun && lane < firstbad;
            okrun = okrun && bad == 0ull;
            sorted = sorted && __ballot(!so_l) == 0ull;
            posv = posv && __ballot(in && !(v > T(0.0))) == 0ull;
            if (in) {
                t.VMAX[i] = vmax_l;
                t.Q[i] = (T)jm_l;  // Q[i] has served: it now holds the index of the fastest layer of the stack 0..i
            }
            if (inl) {
                t.PS[i] = ps_l;
                t.PSA[i] = psa_l;
                t.HVOK[i] = hvok_l ? T(1.0) : T(0.0);
            }
        }
        if (lane == 0)  // spare entry: 0 = unsorted interfaces, 1 = sorted, 2 = sorted, no negative thickness, every velocity positive
            t.HVOK[nlayers] = (sorted && t.D[0] == t.D[0]) ? ((posv && t.D[0] >= T(0.0)) ? T(2.0) : T(1.0)) : T(0.0);
    }
    __syncthreads();
    if (t.HWR) {
        // one row per stack height a lane can have (r <= nlayers): with the constant row length of table form 1 that is
        // (nlayers + 1) x 21 entries, not all 441 -- the sampler's k <= 10 models pay a quarter of the fill per staged model,
        // and no entry of HW beyond the model's own is read
        const int nrows = nlayers + 1 < t.row_len ? nlayers + 1 : t.row_len;
        const int n = nrows * t.row_len;
        for (int e = threadIdx.x; e < n; e += blockDim.x) {
            const int r = e / t.row_len, i = e - r * t.row_len;
            LayerPair<T> v;
            v.hv = i < r ? t.HW[i].hv : T(0.0);
            v.vv = i < r ? t.HW[i].vv : T(0.0);
            t.HWR[e] = v;
        }
    }
    for (int i = threadIdx.x; i <= nlayers; i += blockDim.x)
        t.VMI[i] = div_ieee(T(1.0), t.VMAX[i]);
    if constexpr (WITH_HINT) {
        // Where the fused kernel's solve order comes from (trip_class): a ray that needs many evaluations is one
        // that runs almost horizontally in the fastest layer j of its stack (p close to 1/vmax).  There the
        // offset is  R ~ A + H_j / sqrt(1 - (p vmax)^2)  with A the offset covered by the other layers at
        // p = 1/vmax, so (R - A) / H_j tells how close to critical the solution lies.  Single precision: this
        // only orders the work, no result depends on it.
        // The table costs O(nlayers^2) operations per workgroup (approximate single-precision reciprocal and
        // reciprocal square root: a few instructions per pair of layers).
        const bool hint_unused = t.SA != nullptr;  // see trip_class
        for (int il = 1 + (int)threadIdx.x; il <= nlayers && !hint_unused; il += blockDim.x) {
            if (nlayers > RTMI_HINT_MAX_NL) {
                // very deep models: one class for all bisection sources (measured at 1024 interfaces: ordering by
                // expected evaluations mixes the layer counts inside a wave, which costs more than it returns)
                t.PA[il] = T(INFINITY);
                t.PC[il] = T(0.0);
                t.PH[il] = T(0.0);
                continue;
            }
            const float inv_vmax = __builtin_amdgcn_rcpf((float)t.VMAX[il]);
            const int j = (int)t.Q[il];  // first layer of the stack (own layer = il included) with the maximum velocity
            float a = 0.0f, c = 0.0f;
            for (int i = 0; i <= il; ++i) {
                if (i == j)
                    continue;
                const float pv = fminf((float)t.V[i] * inv_vmax, 0.999999f);
                const float r = pv * __builtin_amdgcn_rsqf(1.0f - pv * pv);  // tan of the ray angle in layer i at p = 1/vmax
                if (i < il)
                    a += (float)t.H[i] * r;
                else
                    c = r;
            }
            t.PA[il] = (T)a;
            t.PC[il] = (T)c;
            t.PH[il] = j < il ? t.H[j] : T(0.0);
        }
    }
    {
        // The bound behind bisection_skip.  For the stack of a source in layer il (fastest velocity vmax, vmi = 1/vmax)
        // and any x <= vmi, a layer slower than vmax contributes at most its term at x = vmi,
        // H V vmi / sqrt(1 - vmi^2 V^2); the layers AT vmax contribute (sum of their H V) x / sqrt(1 - x^2 vmax^2).
        // Approximate arithmetic (v_rsq + one Newton step) is enough: the bound is used with a margin (1e-6 relative in
        // double, 1e-4 in single precision) and decides only which evaluations need not be made, never a value.
        // +Inf = "no bound" (negative thicknesses, unsorted interfaces).
        const bool usable = t.HVOK[nlayers] != T(0.0);
        for (int il = 1 + (int)threadIdx.x; il <= nlayers && t.SA; il += blockDim.x) {
            T sa = T(0.0), hc = T(0.0), sc = T(0.0);
            if (!usable) {
                sa = T(INFINITY);
            } else {
                const T vmax = t.VMAX[il], vmi = div_ieee(T(1.0), vmax);
                for (int i = 0; i <= il; ++i) {
                    const T vi = t.V[i];
                    const T hv = i < il ? t.HV[i] : T(1.0);      // own layer: coefficient of its thickness hl times ...
                    const T vfac = i < il ? T(1.0) : vi;          // ... its velocity
                    if (i < il && !(t.H[i] >= T(0.0)))
                        sa = T(INFINITY);
                    if (vi >= vmax) {
                        if (i < il)
                            hc += hv;
                        continue;
                    }
                    const T y = vmi * vi, w = T(1.0) - y * y;
                    T r;
                    if constexpr (std::is_same<T, double>::value)
                        r = __builtin_amdgcn_rsq(w);
                    else
                        r = __builtin_amdgcn_rsqf(w);
                    r = r * (T(1.5) - T(0.5) * w * r * r);
                    const T term = hv * vfac * vmi * r;
                    if (i < il)
                        sa += term;
                    else
                        sc = term;
                }
            }
            t.SA[il] = sa;
            t.SC[il] = sc;
            t.HC[il] = hc;
        }
    }
    __syncthreads();
}

// Class of a source that starts with the bisection, by the evaluations it is expected to take (about
// 5.5 + 2 log2((R - A) / H_j), see stage_model): RTMI_TRIP_CLASSES - 1 for the cheapest, one less per threshold
// passed.  Three classes are in use (measured: finer classes finish the lanes of a wave closer together but mix
// the layer counts inside a wave more; seven classes were 2 % slower, two 1 % slower).  A scheduling hint: it
// decides when the source is handed to a lane, nothing else.
template <typename T>
__device__ __forceinline__ int trip_class(const LayerTable<T> &t, int il, T R, T z)
{
    // Models that carry the bound of bisection_skip (<= RTMI_SKIP_MAX_NL interfaces) enter the
    // bisection a trip or two before it ends, whatever the geometry: one class (measured: -9 % on a tempering slot
    // against the three classes below, which sort by the trips that are no longer made).
    if (t.SA)
        return RTMI_TRIP_CLASSES - 1;
    const T hl = z - t.D[il - 1];
    const T num = R - (t.PA[il] + hl * t.PC[il]);
    const T hj = t.PH[il];
    const T den = hj > T(0.0) ? hj : hl;
    int c = RTMI_TRIP_CLASSES - 1;
    c -= (num > den * T(9.51));    // 2^3.25: about 12 evaluations or more
    c -= (num > den * T(64.0));    // 2^6:    about 17.5 or more
    return c;
}

// Maximum over the wave of a NON-NEGATIVE value, as a scalar.  DPP row shifts and row broadcasts (no LDS
// traffic): lanes without a source lane contribute 0.
__device__ __forceinline__ int wave_max_i32(int v)
{
#define RTMI_DPP_MAX(ctrl, row_mask)                                                  \
    {                                                                                 \
        const int o = __builtin_amdgcn_update_dpp(0, v, ctrl, row_mask, 0xf, false);  \
        v = o > v ? o : v;                                                            \
    }
    RTMI_DPP_MAX(0x111, 0xf)  // row_shr:1
    RTMI_DPP_MAX(0x112, 0xf)  // row_shr:2
    RTMI_DPP_MAX(0x114, 0xf)  // row_shr:4
    RTMI_DPP_MAX(0x118, 0xf)  // row_shr:8   -> lane 15 of every row holds the row's maximum
    RTMI_DPP_MAX(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    RTMI_DPP_MAX(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
#undef RTMI_DPP_MAX
    return __builtin_amdgcn_readlane(v, 63);
}

// whichLayer (:9-32): 1-based index of the first interface strictly below z; nlayers+1 below the
// last one; nlayers when z sits exactly on the last interface (diff == 0 is neither > 0 nor < 0).
// Every lane of the wave must call it (wave-uniform loops).  With the interfaces in non-decreasing order
// (the flag stage_model leaves in HVOK[nlayers]) the first interface below z is found by bisection:
// "first i with depths(i) > z" is then the number of interfaces <= z.  Unsorted interfaces take the
// reference's own scan.
template <typename T>
__device__ __forceinline__ int which_layer(const LayerTable<T> &t, int nlayers, T z)
{
    int nl = 0;
    if (t.HVOK[nlayers] != T(0.0)) {
        int lo = 0, hi = nlayers;  // the count c of interfaces <= z lies in [lo, hi]
        while (__any(lo < hi)) {
            const bool act = lo < hi;
            const int mid = (lo + hi) >> 1;
            const bool le = t.D[act ? mid : 0] <= z;
            lo = (act && le) ? mid + 1 : lo;
            hi = (act && !le) ? mid : hi;
        }
        nl = lo < nlayers ? lo + 1 : 0;
        if (z != z)
            nl = 0;  // NaN depth: every comparison of the reference is false
    } else {
        for (int i = 0; i < nlayers; ++i) {
            const bool below = t.D[i] > z;  // == (depths(i) - dph > 0)
            if (below && nl == 0)
                nl = i + 1;
            if (__all(nl != 0))
                break;
        }
    }
    if (nl == 0)
        nl = (t.D[nlayers - 1] - z < T(0.0)) ? nlayers + 1 : nlayers;
    return nl;
}

// ---------------------------------------------------------------------------------------------
// Correctly rounded sqrt and divide without the range handling the general expansions carry.
//
// hipcc expands an fp64 `sqrt(w)` to  v_rsq_f64 + the Goldschmidt/Newton steps below, wrapped in a
// pre/post scaling for w < 2^-767 and a class test for 0/inf, and an fp64 `n / d` to
// v_div_scale x2 + v_rcp_f64 + the steps below + v_div_fmas + v_div_fixup, where the scale/fixup
// instructions only act on operands or quotients near the ends of the exponent range (and on 0, inf,
// NaN).  The fast forms are the SAME instruction sequences with those no-op wrappers removed, so for
// operands inside the guarded ranges they return the same bits as `sqrt` and `/` (checked on the
// GPU by tests/test_gpu_parity.py::test_fast_math_equals_ieee against the plain operators on the
// operands of real solves and on random operands).  eval_f_fp only takes them when every operand
// of the wave is inside the guard, and otherwise re-evaluates with the plain operators.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double sqrt_guarded(double w)  // w in (1e-60, 1e60)
{
    const double y = __builtin_amdgcn_rsq(w);
    double g = w * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, w);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, w);
    g = __builtin_fma(d, h, g);
    return g;
}

// ---------------------------------------------------------------------------------------------
// The IEEE divide.  hipcc's expansion of `n / d` (v_rcp + two Newton steps + one residual correction + scale/fixup)
// is FAITHFUL but not always correctly rounded on gfx950: for a divisor a few ulps below a power of two (mantissa
// 2^53 - M, M = 5, 11, 13 ... observed) the refined reciprocal lands on the wrong side of a rounding midpoint
// (Markstein's exception, widened by the error the refinement starts from), and quotients that sit next to a
// midpoint themselves come out one ulp off -- e.g. 0x1.6666666666663p-130 / 0x1.ffffffffffffbp-62 returns
// ...666p-69 where IEEE 754 (and the reference's x86 divide) gives ...667p-69 (tests/hardcases.py builds such
// operands; profiles/r02c/gpu_div_vs_cpu.txt).  One wrong ulp flips a solver branch, so every division the kernels
// perform outside their guarded fast forms goes through div_ieee: the compiler's quotient plus ONE repair step --
// the exact remainders of q and of its neighbour on the remainder's side decide which of the two is the correctly
// rounded quotient (a faithful q is at most one ulp off, and n - d*q of a faithful q is exactly representable).
// Operands outside the comfortable exponent range (tiny n, sub-normal q, Inf, NaN) keep the compiler's result.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double div_ieee(double n, double d)
{
    const double q = n / d;
    const double rem = __builtin_fma(-d, q, n);
    const bool up = (rem > 0.0) == (d > 0.0);                        // the exact quotient lies above q
    const long long step = (up == (q > 0.0)) ? 1ll : -1ll;            // ... i.e. one ulp further from / closer to zero
    const double qa = __longlong_as_double(__double_as_longlong(q) + step);
    const double rema = __builtin_fma(-d, qa, n);
    const bool safe = fabs(n) > 1e-200 && fabs(q) > 1e-290 && fabs(q) < 1e300;
    return (safe && fabs(rema) < fabs(rem)) ? qa : q;
}
// Single precision needs no repair: hipcc's fp32 divide (v_div_scale / v_rcp / fma chain in the denormal-preserving
// mode / v_div_fmas / v_div_fixup) returned the IEEE result for a midpoint numerator of EVERY float divisor of the
// kernels' range (3.3e8 constructed quotients, profiles/r02e/gpu_div_vs_cpu_f32.txt; tests/test_gpu_fastmath_hard.py
// re-checks a thirteenth of them against numpy on every run).
__device__ __forceinline__ float div_ieee(float n, float d) { return n / d; }

// n / d for d in (1e-90, 1e30), n == 0 or |n| in (1e-60, 1e60): v_rcp_f64 refined by THREE Newton steps is the
// correctly rounded reciprocal (the third step starts from within an ulp, whatever the accuracy of v_rcp_f64) for
// every divisor but the all-ones mantissa, so the residual correction returns the correctly rounded quotient
// (Markstein 1990).  `ok` is cleared for a divisor whose low mantissa word is all ones (the caller then takes div_ieee).
__device__ __forceinline__ double div_guarded(double n, double d, bool &ok)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = n * r;
    const double res = __builtin_fma(-d, q, n);
    ok = ok && (uint32_t)__double2loint(d) != 0xFFFFFFFFu;
    return __builtin_fma(res, r, q);
}

// One layer term of costFunc and costFunc_Prime with a shared reciprocal:  tf = a / s,  tp = hv / ((s*s)*s),
// s = sqrt(w).  The square-root iteration already carries h ~ 1/(2s); 2h refined by TWO Newton steps against
// the final s is the correctly rounded reciprocal r of s (Markstein 1990; one step is not enough, see below), which
// serves the first quotient directly and -- cubed, then refined once against the ROUNDED cube s3 the reference
// divides by -- the second.  Each quotient ends with the same residual correction q' = fma(fma(-d, q, n), r, q) as the plain
// divide, which returns the correctly rounded n/d when r is the correctly rounded reciprocal.  The one family
// of divisors for which a refined reciprocal is NOT correctly rounded is the all-ones mantissa (the refinement
// lands exactly on a rounding tie); there, quotients of "short" numerators (powers of two, small integers) come
// out one ulp low (`profiles/allones_check.py` finds them).  term_guarded_n therefore also returns whether
// any of its divisors s, s3 has an all-ones low mantissa word -- one integer compare each, a necessary
// condition hit by 1 ordinary operand in 2^32 -- and the caller treats that like any other operand outside
// the guards: the wave re-evaluates with the plain operators.  Two v_rcp_f64 (quarter rate) and two fma per
// term are saved against two independent divides.  Bitwise equality with `sqrt` and `/` is checked on the
// GPU (test_fast_math_equals_ieee, incl. the all-ones family with short numerators) and by every parity test.
//
// N independent layer terms, statement by statement side by side: the same operations as N calls of
// term_guarded, written interleaved so that a wave that has its SIMD to itself (the end of a round) has N
// dependency chains in flight.
template <int N>
__device__ __forceinline__ bool term_guarded_n(const double (&w)[N], const double (&a)[N], const double (&hv)[N],
                                               double (&tf)[N], double (&tp)[N])
{
    double y[N], g[N], h[N], e[N], d[N], r[N], q[N], s3[N], r3[N];
#define RTMI_EACH for (int j = 0; j < N; ++j)
#pragma unroll
    RTMI_EACH y[j] = __builtin_amdgcn_rsq(w[j]);
#pragma unroll
    RTMI_EACH g[j] = w[j] * y[j];
#pragma unroll
    RTMI_EACH h[j] = y[j] * 0.5;
#pragma unroll
    RTMI_EACH e[j] = __builtin_fma(-h[j], g[j], 0.5);
#pragma unroll
    RTMI_EACH g[j] = __builtin_fma(g[j], e[j], g[j]);
#pragma unroll
    RTMI_EACH h[j] = __builtin_fma(h[j], e[j], h[j]);
#pragma unroll
    RTMI_EACH d[j] = __builtin_fma(-g[j], g[j], w[j]);
#pragma unroll
    RTMI_EACH g[j] = __builtin_fma(d[j], h[j], g[j]);
#pragma unroll
    RTMI_EACH d[j] = __builtin_fma(-g[j], g[j], w[j]);
#pragma unroll
    RTMI_EACH g[j] = __builtin_fma(d[j], h[j], g[j]);
#pragma unroll
    RTMI_EACH r[j] = h[j] + h[j];
    // two Newton steps: 2h carries the error of the coupled iteration (measured: up to ~13 units of 2^-53; the ISA
    // only promises v_rsq_f64 to ~2^-23, i.e. up to ~190 such units after one coupled step), and ONE step from k
    // units rounds 1/s to the wrong neighbour whenever the mantissa of s lies within k of all-ones (1/s then sits
    // k^2 2^-106 below ... M^2 2^-106 above a rounding midpoint, M = 2^53 - mantissa) -- which the first evaluation
    // of every bisection (x1 = 1e-10: s = 1 - M 2^-53 with M ~ 5e-5 v^2) meets systematically for slow layers.  The
    // second step starts from within one ulp and leaves exactly Markstein's exception, M = 1, to the guard below.
#pragma unroll
    RTMI_EACH e[j] = __builtin_fma(-g[j], r[j], 1.0);
#pragma unroll
    RTMI_EACH r[j] = __builtin_fma(r[j], e[j], r[j]);
#pragma unroll
    RTMI_EACH e[j] = __builtin_fma(-g[j], r[j], 1.0);
#pragma unroll
    RTMI_EACH r[j] = __builtin_fma(r[j], e[j], r[j]);
#pragma unroll
    RTMI_EACH q[j] = a[j] * r[j];
#pragma unroll
    RTMI_EACH tf[j] = __builtin_fma(__builtin_fma(-g[j], q[j], a[j]), r[j], q[j]);
#pragma unroll
    RTMI_EACH s3[j] = (g[j] * g[j]) * g[j];
    // r correctly rounded => (r*r)*r is within 7 units of 2^-53 of 1/s3 (three factors, two roundings, two more in s3):
    // one Newton step rounds 1/s3 correctly unless the mantissa of s3 lies within 7 of all-ones -- the guard below.
#pragma unroll
    RTMI_EACH r3[j] = (r[j] * r[j]) * r[j];
#pragma unroll
    RTMI_EACH e[j] = __builtin_fma(-s3[j], r3[j], 1.0);
#pragma unroll
    RTMI_EACH r3[j] = __builtin_fma(r3[j], e[j], r3[j]);
#pragma unroll
    RTMI_EACH q[j] = hv[j] * r3[j];
#pragma unroll
    RTMI_EACH tp[j] = __builtin_fma(__builtin_fma(-s3[j], q[j], hv[j]), r3[j], q[j]);
    // the guard: no s with an all-ones low mantissa word, no s3 whose low word is within 7 of all-ones (necessary
    // conditions of the two exceptions above; one ordinary operand in 2^29 trips them): two integer maxima per group
    uint32_t mx = 0, mx3 = 0;
#pragma unroll
    RTMI_EACH
    {
        mx = max(mx, (uint32_t)__double2loint(g[j]));
        mx3 = max(mx3, (uint32_t)__double2loint(s3[j]));
    }
#undef RTMI_EACH
    return mx != 0xFFFFFFFFu && mx3 < 0xFFFFFFF8u;
}

// The single-precision form of term_guarded_n: v_rsq_f32 is good to an ulp, so one coupled iteration and one
// correction give the correctly rounded square root (checked over EVERY float in the guarded range by
// rtmi_selftest_fastmath_f32), the rest as in double precision.  w in (1e-12, 1], numerators 0 or in (1e-18, 1e18).
template <int N>
__device__ __forceinline__ bool term_guarded_n(const float (&w)[N], const float (&a)[N], const float (&hv)[N],
                                               float (&tf)[N], float (&tp)[N])
{
    float y[N], g[N], h[N], e[N], d[N], r[N], q[N], s3[N], r3[N];
#define RTMI_EACH for (int j = 0; j < N; ++j)
#pragma unroll
    RTMI_EACH y[j] = __builtin_amdgcn_rsqf(w[j]);
#pragma unroll
    RTMI_EACH g[j] = w[j] * y[j];
#pragma unroll
    RTMI_EACH h[j] = y[j] * 0.5f;
#pragma unroll
    RTMI_EACH e[j] = __builtin_fmaf(-h[j], g[j], 0.5f);
#pragma unroll
    RTMI_EACH g[j] = __builtin_fmaf(g[j], e[j], g[j]);
#pragma unroll
    RTMI_EACH h[j] = __builtin_fmaf(h[j], e[j], h[j]);
#pragma unroll
    RTMI_EACH d[j] = __builtin_fmaf(-g[j], g[j], w[j]);
#pragma unroll
    RTMI_EACH g[j] = __builtin_fmaf(d[j], h[j], g[j]);
#pragma unroll
    RTMI_EACH d[j] = __builtin_fmaf(-g[j], g[j], w[j]);
#pragma unroll
    RTMI_EACH g[j] = __builtin_fmaf(d[j], h[j], g[j]);
#pragma unroll
    RTMI_EACH r[j] = h[j] + h[j];
    // two refinements: the first leaves r faithful, the second correctly rounded (which the quotient below needs)
#pragma unroll
    RTMI_EACH e[j] = __builtin_fmaf(-g[j], r[j], 1.0f);
#pragma unroll
    RTMI_EACH r[j] = __builtin_fmaf(r[j], e[j], r[j]);
#pragma unroll
    RTMI_EACH e[j] = __builtin_fmaf(-g[j], r[j], 1.0f);
#pragma unroll
    RTMI_EACH r[j] = __builtin_fmaf(r[j], e[j], r[j]);
#pragma unroll
    RTMI_EACH q[j] = a[j] * r[j];
#pragma unroll
    RTMI_EACH tf[j] = __builtin_fmaf(__builtin_fmaf(-g[j], q[j], a[j]), r[j], q[j]);
#pragma unroll
    RTMI_EACH s3[j] = (g[j] * g[j]) * g[j];
#pragma unroll
    RTMI_EACH r3[j] = (r[j] * r[j]) * r[j];
#pragma unroll
    RTMI_EACH e[j] = __builtin_fmaf(-s3[j], r3[j], 1.0f);
#pragma unroll
    RTMI_EACH r3[j] = __builtin_fmaf(r3[j], e[j], r3[j]);
#pragma unroll
    RTMI_EACH e[j] = __builtin_fmaf(-s3[j], r3[j], 1.0f);
#pragma unroll
    RTMI_EACH r3[j] = __builtin_fmaf(r3[j], e[j], r3[j]);
#pragma unroll
    RTMI_EACH q[j] = hv[j] * r3[j];
#pragma unroll
    RTMI_EACH tp[j] = __builtin_fmaf(__builtin_fmaf(-s3[j], q[j], hv[j]), r3[j], q[j]);
    // no divisor with an all-ones mantissa
    uint32_t mx = 0;
#pragma unroll
    RTMI_EACH mx = max(mx, max(__float_as_uint(g[j]) & 0x7FFFFFu, __float_as_uint(s3[j]) & 0x7FFFFFu));
#undef RTMI_EACH
    return mx != 0x7FFFFFu;
}

// the word of a value that orders positive numbers like integers
__device__ __forceinline__ int hi_word(double v) { return __double2hiint(v); }
__device__ __forceinline__ int hi_word(float v) { return (int)__float_as_uint(v); }

// ---- order-independent accumulation of squared residuals (the sampler's likelihood) ---------------------------------------
// A 126-bit fixed-point accumulator in two words, least significant bit 2^-64 s^2: w0 holds the fraction, w1 the integer part
// (62 bits) and two flags.  An addend (>= 0, below 2^32 s^2) is ROUNDED to that grid -- a fixed function of the addend alone,
// |error| <= 2^-65 = 2.7e-20 s^2, unbiased -- and added as an integer, and integer addition is associative: the sum of a
// chain's squared residuals does not depend on the order in which lanes, waves, workgroups or launches of different geometry
// add them up, nor on whether a residual was computed in this launch or taken from the chain's cache.  The two top bits of
// the high word are flags (the sum itself stays below 2^62 s^2: 2^30 addends of the largest size): bit 63 = a NaN was added
// (the sum is NaN, as the reference's SUM would be), bit 62 = an addend >= 2^32 s^2 (the sum is reported as +Inf).
// [Round 3 had the grid at 2^-76 with addends below 2^20 s^2 -- which the reference's own -999 sentinel exceeds as soon as an
// observed time is above 25 s: (t_obs + 999)^2 >= 2^20.  The reference sums that residual as a finite 1e6 (loglhood.f90:191
// with GetPTime's -999), and so do the batch API and the host sampler; with the grid here the sentinel's residual stays
// finite for every observed time below 64 000 s.  Four registers per lane; the 192-bit form of round 2, exact down to
// 2^-128, took seven and pushed the sampler's kernel into scratch.]
struct Acc192 {  // (the name of the round-2 form is kept)
    unsigned long long w0, w1;
};
#define RTMI_ACC_NAN (1ull << 63)
#define RTMI_ACC_INF (1ull << 62)
#define RTMI_ACC_FLAGS (RTMI_ACC_NAN | RTMI_ACC_INF)

__device__ __forceinline__ Acc192 acc_zero()
{
    Acc192 a;
    a.w0 = a.w1 = 0ull;
    return a;
}

// a += b (flags are OR-ed: the value bits of w1 never carry into them)
__device__ __forceinline__ void acc_add(Acc192 &a, const Acc192 &b)
{
    const unsigned long long fl = (a.w1 | b.w1) & RTMI_ACC_FLAGS;
    const unsigned long long t = a.w0 + b.w0;
    const unsigned long long c = t < b.w0 ? 1ull : 0ull;
    a.w0 = t;
    a.w1 = (((a.w1 & ~RTMI_ACC_FLAGS) + (b.w1 & ~RTMI_ACC_FLAGS) + c) & ~RTMI_ACC_FLAGS) | fl;
}

// a += x for a squared residual x (>= 0, or NaN)
__device__ __forceinline__ void acc_add_double(Acc192 &a, double x)
{
    if (!(x >= 0.0)) {
        a.w1 |= RTMI_ACC_NAN;
        return;
    }
    if (x >= 4294967296.0) {  // 2^32
        a.w1 |= RTMI_ACC_INF;
        return;
    }
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const int e = (int)((bits >> 52) & 0x7FFull);
    if (e == 0)
        return;  // zero or subnormal: below the grid
    unsigned long long m = (bits & 0xFFFFFFFFFFFFFull) | 0x10000000000000ull;  // x = m 2^(e - 1075)
    int shift = e - 1011;                                                     // position of m's lowest bit above 2^-64
    if (shift < 0) {
        if (shift < -53)
            return;  // below half a grid step
        m = (m + (1ull << (-shift - 1))) >> -shift;  // round to the grid (ties up)
        shift = 0;
    }
    // x < 2^32: shift <= 43, so m << shift ends below bit 96
    const unsigned long long lo = m << shift;
    const unsigned long long hi = shift ? m >> (64 - shift) : 0ull;
    const unsigned long long t = a.w0 + lo;
    const unsigned long long c = t < lo ? 1ull : 0ull;
    a.w0 = t;
    a.w1 += hi + c;  // no carry reaches the flag bits (see above)
}

// the accumulated sum as a double: a fixed function of the integer
__device__ __forceinline__ double acc_to_double(const Acc192 &a)
{
    if (a.w1 & RTMI_ACC_NAN)
        return __builtin_nan("");
    if (a.w1 & RTMI_ACC_INF)
        return __builtin_inf();
    return __ull2double_rn(a.w1) + __ull2double_rn(a.w0) * 5.421010862427522e-20;  // 2^-64
}

// lower end of the guarded range of w = 1 - p^2 v^2
template <typename T>
__device__ __forceinline__ constexpr T guard_w()
{
    return std::is_same<T, double>::value ? T(1e-60) : T(1e-12);  // fp32: sqrt(w)^3 must stay a normal number
}

// One layer term: tf = (hv*x)/s, tp = hv/((s*s)*s), s = sqrt(1 - xx*vv).  Returns the guard of the fast form.
template <bool FAST, typename T>
__device__ __forceinline__ bool layer_term(T xx, T x, T hv, T vv, T &tf, T &tp)
{
    const T w = T(1.0) - xx * vv;
    if constexpr (FAST) {
        const T wa[1] = {w}, aa[1] = {hv * x}, ha[1] = {hv};
        T f1[1], p1[1];
        const bool sound = term_guarded_n<1>(wa, aa, ha, f1, p1);
        tf = f1[0];
        tp = p1[0];
        return sound && w > guard_w<T>();
    } else {
        const T s = sqrt(w);
        tf = div_ieee(T(hv * x), s);
        tp = div_ieee(hv, T((s * s) * s));
        return true;
    }
}

// N layer terms at once (independent chains), added to (sf, sp) in index order.  Returns the fast form's guard.
template <bool FAST, int N, typename T>
__device__ __forceinline__ bool add_terms(T xx, T x, const T (&hv)[N], const T (&vv)[N], T &sf, T &sp)
{
    T tf[N], tp[N];
    bool ok = true;
    if constexpr (FAST) {
        T w[N], a[N];
        // every w above the lower end of the guarded range: for positive numbers the order of the values is the
        // order of their (high) words as integers, so one compare on the minimum serves the group (a negative w
        // has the sign bit set and fails; a NaN w can only come from a NaN x or velocity, which the caller's
        // guards on x and on the table catch)
        int mn = 0x7FFFFFFF;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            w[j] = T(1.0) - xx * vv[j];
            a[j] = hv[j] * x;
            mn = min(mn, hi_word(w[j]));
        }
        ok = mn > hi_word(guard_w<T>());
        ok = term_guarded_n<N>(w, a, hv, tf, tp) && ok;
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j)
            layer_term<false>(xx, x, hv[j], vv[j], tf[j], tp[j]);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        sf = sf + tf[j];
        sp = sp + tp[j];
    }
    return ok;
}

// costFunc (:184-202) and costFunc_Prime (:206-221) at x for the lane's stack:
// `nfull` fully crossed layers from the table plus the partial layer (hvl, vvl).
// `nfull_wave` >= max of nfull over the wave keeps the loop wave-uniform.
// Terms are evaluated two at a time (two independent sqrt/divide chains in flight: the evaluation is a long
// dependent chain, and in the drain a wave has its SIMD to itself) and ADDED strictly in layer order.
template <bool FAST, bool ROWS, typename T>
__device__ __forceinline__ bool eval_terms(const LayerTable<T> &t, T x, int nfull, int nfull_wave, T hvl,
                                           T vvl, T R, T &f, T &fp)
{
    // ROWS: the lane reads row `nfull` of the row table at the wave-uniform column i (zeros past its stack);
    // otherwise it picks entry i or the "no layer" entry of the one shared table.
    const LayerPair<T> *row = ROWS ? t.HWR + nfull * t.row_len : t.HW;
    auto entry = [&](int i) -> LayerPair<T> { return ROWS ? row[i] : t.HW[i < nfull ? i : t.nul]; };
    // No lane is masked inside the loop: a lane whose stack ends before layer i reads the table's "no layer"
    // entry {0, 0} instead, whose term is (0*x)/sqrt(1) = 0 and 0/1 = 0, and sf + 0 == sf bit for bit (sf is never
    // -0: it starts at +0).  That keeps the loop free of EXEC masks and branches, so the two chains of a pair are
    // interleaved by the scheduler.
    const T xx = x * x;
    T sf = T(0.0), sp = T(0.0);
    bool ok = true;
    int i = 0;
    if constexpr (!FAST) {
        // the plain operators are the rare path (operands outside the guards): one term per trip, small code
#pragma unroll 1
        for (; i < nfull_wave; ++i) {
            const LayerPair<T> e = entry(i);
            const T hv[1] = {e.hv}, vv[1] = {e.vv};
            add_terms<false, 1>(xx, x, hv, vv, sf, sp);
        }
        const T hv[1] = {hvl}, vv[1] = {vvl};
        add_terms<false, 1>(xx, x, hv, vv, sf, sp);
        f = R - sf;
        fp = -sp;
        return true;
    }
    for (; i + RTMI_TERM_GROUP - 1 < nfull_wave; i += RTMI_TERM_GROUP) {
        T hv[RTMI_TERM_GROUP], vv[RTMI_TERM_GROUP];
#pragma unroll
        for (int j = 0; j < RTMI_TERM_GROUP; ++j) {
            const LayerPair<T> e = entry(i + j);
            hv[j] = e.hv;
            vv[j] = e.vv;
        }
        ok = add_terms<FAST, RTMI_TERM_GROUP>(xx, x, hv, vv, sf, sp) && ok;
    }
    for (; i + 1 < nfull_wave; i += 2) {
        const LayerPair<T> e0 = entry(i);
        const LayerPair<T> e1 = entry(i + 1);
        const T hv[2] = {e0.hv, e1.hv}, vv[2] = {e0.vv, e1.vv};
        ok = add_terms<FAST, 2>(xx, x, hv, vv, sf, sp) && ok;
    }
    if (i < nfull_wave) {  // wave-uniform: the odd full layer together with the source's own partial layer
        const LayerPair<T> e0 = entry(i);
        const T hv[2] = {e0.hv, hvl}, vv[2] = {e0.vv, vvl};
        ok = add_terms<FAST, 2>(xx, x, hv, vv, sf, sp) && ok;
    } else {
        const T hv[1] = {hvl}, vv[1] = {vvl};
        ok = add_terms<FAST, 1>(xx, x, hv, vv, sf, sp) && ok;
    }
    f = R - sf;
    fp = -sp;
    return ok;
}

// `numer_ok`: lane-constant part of the guard (table products and hvl are 0 or of moderate magnitude).
// `busy`: the lane carries a live source (idle lanes never veto the fast path).
template <bool ROWS = false, typename T>
__device__ __forceinline__ void eval_f_fp(const LayerTable<T> &t, T x, int nfull, int nfull_wave, T hvl,
                                          T vvl, T R, bool numer_ok, bool busy, T &f, T &fp)
{
    const bool ok = eval_terms<true, ROWS>(t, x, nfull, nfull_wave, hvl, vvl, R, f, fp) && numer_ok && mag_ok(x);
    if (__any(busy && !ok))  // rare: some operand near the ends of the range, 0, Inf or NaN
        eval_terms<false, false>(t, x, nfull, nfull_wave, hvl, vvl, R, f, fp);
}

// f / f' of the state machine (cf/cfp of GetPTime :138, deltax of solvebst :384 and of solve :294): the guarded divide
// when every live lane's operands are inside its range (f 0 or of moderate magnitude, f' of moderate magnitude and
// not an all-ones mantissa), the repaired IEEE divide for the whole wave otherwise.  Every lane of the wave calls it.
template <typename T>
__device__ __forceinline__ T solver_quotient(T f, T fp, bool busy)
{
    if constexpr (std::is_same<T, double>::value) {
        bool ok = mag_ok(f) && mag_nz(fp);
        const T q = div_guarded(f, fp, ok);
        if (!__any(busy && !ok))
            return q;
    }
    return div_ieee(f, fp);
}

// Solver state of one source (one lane).
template <typename T>
struct Lane {
    T R;     // src_offset
    T hvl;   // (z - D[il-1]) * V[il]   partial layer of the source
    T vvl;   // V[il]^2
    T vmi;   // 1/maxval(vp)
    T safe;  // 1/maxval(vp) - 1d-10   (:142, :388)   } per-source constants of the state machine, kept in
    T clampv;  // 1/maxval(vp) - 1d-9  (:299-304)     } registers instead of being recomputed every trip
    T x;     // Newton iterate / final ray parameter
    T xe;    // where the next (f, f') evaluation happens
    T xb;    // bisection bracket end `x` of solvebst
    T dx;
    int nfull;    // fully crossed layers = il
    int st;
    int k;        // trip counter of the current loop
    bool conv;
    bool numer_ok;  // hvl and the table products of the stack are inside eval_f_fp's fast-path guard
    bool skip0;     // solvebst's first evaluation, f(x1 = 1d-10) (:354), is provably positive: only its sign is used
};

// The source's layer-dependent constants (InsertLayer :38-69 as table lookups).  il = nl - 1 >= 1.
template <typename T>
__device__ __forceinline__ void lane_setup(Lane<T> &s, const LayerTable<T> &t, int il, T R, T z)
{
    const T vl = t.V[il];
    const T hl = z - t.D[il - 1];  // depths_new(nl) - depths_new(nl-1)
    s.R = R;
    s.hvl = hl * vl;
    s.vvl = t.VV[il];
    s.vmi = t.VMI[il];
    s.safe = s.vmi - T(1e-10);
    s.clampv = s.vmi - T(1e-9);
    s.nfull = il;
    s.numer_ok = mag_ok(s.hvl) && (t.HVOK[il - 1] != T(0.0));
    // solvebst starts with f1 = costFunc(x1 = 1d-10) and uses NOTHING of it but `f1 < 0` (:354-365).  The sum in f1 is
    // at most x1 * SUM |H_i V_i| / min s_i, and with every velocity below 1e7 m/s  s_i = sqrt(1 - x1^2 V_i^2) > 0.9999995;
    // the roundings of the products, quotients and of either sum are ~1e-13 relative.  So R > 1.01e-10 * SUM |H_i V_i|
    // (a 1 % margin over all of that) proves f1 > 0 whatever the last bits are, and the evaluation -- one whole trip
    // of the source, for two sources in three -- need not be made.  NaN or Inf anywhere makes the test false.
    s.skip0 = s.R > T(1.01e-10) * (t.PSA[il - 1] + fabs(s.hvl)) && t.VMAX[il] < T(1e7);
}

// ---------------------------------------------------------------------------------------------
// The evaluation that need not be made: solve()'s LAST one (:281-291), which only confirms abs(fx) < tol.
//
// Model of the arithmetic: F(p) = R - sum_i hv_i p / sqrt(1 - p^2 vv_i) with the table's (rounded) hv_i >= 0, vv_i as exact
// coefficients; the solver's fx, fxprime at x are F(x) + e1 and F'(x) (1 + e2).  F is decreasing and CONCAVE on
// (0, 1/vmax):  F'' = -sum 3 hv_i vv_i p / s_i^5.  A Newton step taken where fx < 0 moves left, xnew = x - fx/fxprime < x,
// and Taylor at x gives
//     F(xnew) = [F(x) + F'(x) (xnew - x)] + F''(xi) (xnew - x)^2 / 2,      xi in [xnew, x].
// |F''| grows with p, and vv_i / w_i(x) <= vmax^2 / w_min(x) with w_min = 1 - x^2 vmax^2, so on that interval
//     |F''(xi)| <= 3 x (vmax^2 / w_min) sum hv_i / s_i(x)^3 = 3 x (vmax^2 / w_min) |F'(x)|,
// and with |F'(x)| q^2 = |fx q| (q = fx/fxprime) the quadratic term is at most  B = 1.5 x vmax^2 |fx q| / w_min.
// The bracket is what the Newton step cancels; what is left of it are the roundings:
//     |bracket| <= |F(x)| (|e2| + 2^-53) + |e1| + |F'(x)| ulp(x)/2.
// Under the guards below (R < 1e5, every w_i >= w_min > 1e-8, |fx| < 1e4, |q| > 1e-9 x, <= 1025 layers) a term of the sums
// carries a relative error below 2.3e-16 / (2 w) + 5 ulp < 1.2e-8 (3.5e-8 in f'), hence |e1| < 1.4e-3, |e2| < 4e-8, the
// bracket < 2.9e-3, and the value the solver would COMPUTE at xnew differs from F(xnew) by < 1.2e-3.  B <= 0.08 therefore
// proves abs(costFunc(xnew)) < 0.0842 < tol: the next trip of solve's loop would set conv = .TRUE. and leave with x = xnew
// (:287-291) -- provided it is a trip of the loop (this one is not the 15th) and xnew was not clamped (:299-304).
// The bound decides only WHETHER that evaluation is made, never a value.  NaN anywhere makes it false.  Double precision
// only: in single precision the noise of the sums is of the order of tol.  On the C2 stream this is the last evaluation
// of 92 % of the sources: 3.85 -> 2.94 evaluations per source in the solve loop (profiles/sim_predict.py).
// ---------------------------------------------------------------------------------------------
// the model's part of newton_lands_converged's guards (wave-uniform; the flag stage_model leaves in HVOK[nlayers])
template <typename T>
__device__ __forceinline__ bool model_terms_nonneg(const LayerTable<T> &t)
{
    return std::is_same<T, double>::value && RTMI_PREDICT && t.HVOK[t.nul - 1] == T(2.0);
}

template <typename T>
__device__ __forceinline__ bool newton_lands_converged(const LayerTable<T> &, const Lane<T> &s, T x, T f, T q, T xnew)
{
    if constexpr (!std::is_same<T, double>::value || !RTMI_PREDICT) {
        return false;
    } else {
        // in terms of vmi = RN(1/vmax), which the lane holds:  w_min / vmax^2 = (vmi - x)(vmi + x) (1 + O(2.2e-16 / w_min)),
        // so  B <= 0.08  <=>  x |fx q| <= (0.08 / 1.5) (vmi - x)(vmi + x);  w_min >= (vmi - x) / vmi.  q > 0 (both sums are
        // negative) makes xnew < x < vmi: no clamp.
        const T gap = s.vmi - x;
        const T u = gap * (s.vmi + x);
        // the lane's part of the guards: no term of f, f' is negative (with the model's part, model_terms_nonneg: sorted
        // interfaces from a non-negative depth, positive velocities -- the caller's `use_pred` -- the source not above its
        // layer's top), everything of moderate magnitude (R < 1e5, vmax < 1e7)
        const bool lane_ok = s.numer_ok && s.hvl >= T(0.0) && s.R < T(1e5) && s.vmi > T(1e-7);
        return lane_ok && f < T(0.0) && f > T(-1e4) && q > T(1e-9) * x && xnew > T(0.0) && gap > T(1e-8) * s.vmi &&
               x * fabs(f * q) <= T(0.0533) * u;
    }
}

// The state after solvebst's entry (:354-365) and its first `dx = dx*0.5; xmid = x + dx` (:368-369), given the sign of
// f(x1): x1 = 1d-10, x2 = 1/maxval(vp) - 1d-12.
template <typename T>
__device__ __forceinline__ void lane_enter_bisection(Lane<T> &s, bool neg)
{
    const T e2 = s.vmi - T(1e-12);
    const T e21 = e2 - T(1e-10);
    const T b_x = neg ? T(1e-10) : e2;
    const T b_dx = (neg ? e21 : -e21) * T(0.5);
    s.xb = b_x;
    s.dx = b_dx;
    s.xe = b_x + b_dx;
    s.k = 1;
    s.st = ST_BIS;
}

// GetPTime head (:112-133): the starting ray parameter.  Returns NaN when the reference would loop
// forever (NaN/Inf source data): such a source is reported as -999.
// FAST: the three divides and the square root in their guarded forms (same bits inside the guards); `ok`
// returns whether every operand was inside them.
template <bool FAST, typename T>
__device__ __forceinline__ T initial_p0_impl(const LayerTable<T> &t, int il, T R, T z, bool &ok)
{
    const T vl = t.V[il];
    const T hl = z - t.D[il - 1];
    const T rr = R * R + z * z;
    T p0;
    if constexpr (FAST && std::is_same<T, double>::value) {
        bool dok = true;  // no divisor with an all-ones low mantissa word
        const T tv = t.PS[il - 1] + div_guarded(hl, vl, dok);  // vertical time above the source
        const T c_harmonic = div_guarded(z, tv, dok);
        const T cos_t = div_guarded(z, sqrt_guarded(rr), dok);
        p0 = div_guarded(cos_t, c_harmonic, dok);
        // every numerator 0 or of moderate magnitude, every denominator of moderate magnitude
        ok = dok && mag_ok(hl) && mag_nz(vl) && mag_nz(tv) && mag_nz(z) && rr > 1e-60 && rr < 1e60 && mag_nz(c_harmonic) &&
             mag_ok(cos_t);
    } else {
        const T c_harmonic = div_ieee(z, t.PS[il - 1] + div_ieee(hl, vl));
        const T cos_t = div_ieee(z, T(sqrt(rr)));
        p0 = div_ieee(cos_t, c_harmonic);
        ok = true;
    }
    const T vmax = t.VMAX[il];
    const T vm1sq = (vmax + T(1.0)) * (vmax + T(1.0));
    // sum(sqrt(1-(p0**2)*(vp+1)**2)) is NaN  <=>  (p0*p0)*(vmax+1)^2 > 1  (monotone in V; the
    // subtraction from 1 has the exact sign).
    int guard = 0;
#pragma unroll 1
    while (!((p0 * p0) * vm1sq <= T(1.0))) {
        p0 = p0 / T(2.0);
        if (++guard > 2200 || p0 != p0)
            return T(__builtin_nan(""));
    }
    return p0;
}

// `valid`: the lane carries a source (other lanes never veto the fast forms).  Every lane of the wave calls it.
template <typename T>
__device__ __forceinline__ T initial_p0(const LayerTable<T> &t, int il, T R, T z, bool valid = true)
{
    bool ok;
    if constexpr (std::is_same<T, double>::value) {
        T p0 = initial_p0_impl<true>(t, il, R, z, ok);
        if (__any(valid && !ok))
            p0 = initial_p0_impl<false>(t, il, R, z, ok);
        return p0;
    } else {
        return initial_p0_impl<false>(t, il, R, z, ok);
    }
}

// Source in the top layer (:94-97): t = sqrt(z^2 + R^2) / v_1.  Called from divergent code (the lanes of a chunk
// that are straight rays): the guarded forms when every such lane is inside the guards, the plain operators otherwise.
template <typename T>
__device__ __forceinline__ T straight_time(T z, T R, T v0)
{
    const T rr = z * z + R * R;
    if constexpr (std::is_same<T, double>::value) {
        bool ok = rr > 1e-60 && rr < 1e60 && mag_nz(v0);
        const T tt = div_guarded(sqrt_guarded(rr), v0, ok);
        if (__all(ok))
            return tt;
    }
    return div_ieee(T(sqrt(rr)), v0);
}

template <typename T>
__device__ __forceinline__ void lane_start(Lane<T> &s, T p0)
{
    s.x = p0;
    s.xe = p0;
    s.xb = T(0.0);
    s.dx = T(0.0);
    s.k = 0;
    s.conv = false;
    s.st = ST_INIT;
}

// One transition of the per-lane state machine after (f, f') was evaluated at s.xe.
// GetPTime's three-way branch (:136-153), solvebst (:339-405) and solve (:226-332).
//
// Written as predicates + selects rather than one branch per state: the lanes of a wave are in
// different states almost always, so a branchy dispatch executes every state's body on every trip;
// the select form shares the common pieces (one quotient, one `check_x`, one Newton update).
// Every value is computed by the same operation on the same operands as in the reference.
// MODE tells what the caller knows about the states present, so that the other states' selects fold away:
// the fused kernel evaluates ST_INIT in a pass of its own and never meets it in the solve loop.
enum : int { ADV_ANY = 0, ADV_INIT_ONLY = 1, ADV_NO_INIT = 2 };
template <int MODE = ADV_ANY, typename T>
__device__ __forceinline__ void lane_advance(const LayerTable<T> &t, Lane<T> &s, T f, T fp, bool use_pred)
{
    const int st = s.st;
    const bool isI = MODE == ADV_INIT_ONLY ? true : (MODE == ADV_NO_INIT ? false : st == ST_INIT);
    // ST_BIS0 comes up once per bisection source: in the fused kernel's solve loop (ADV_NO_INIT) it is handled by a
    // block of its own, entered only on trips where some lane is in that state, and is no operand of the selects.
    const bool isB0 = MODE == ADV_ANY && (st == ST_BIS0), isB = MODE != ADV_INIT_ONLY && (st == ST_BIS),
               isN = MODE != ADV_INIT_ONLY && (st == ST_NEWTON), isF = MODE != ADV_INIT_ONLY && (st == ST_FINAL);
    // cf/cfp of GetPTime (:138), deltax of solvebst (:384) and of solve (:294): same operands, one divide
    const T q = solver_quotient(f, fp, st != ST_DONE);
    const bool neg = f < T(0.0);
    const bool small = fabs(f) < T(0.1);   // abs(fx) < tol (:287), abs(fmid) < tol (:394)
    const T safe = s.safe;  // 1/maxval(V)-(1d-10)      (:142, :388)

    // solvebst trip, first half: `if (fmid < 0) x = xmid` (:371-373)
    const T xb1 = (isB && neg) ? s.xe : s.xb;
    // check_x = p0 - cf/cfp (:138)  |  check_x = x - deltax (:386)
    const T check_x = (isI ? s.xe : xb1) - q;
    const bool jump = check_x < safe;

    const bool initN = isI && (neg || jump);  // (:139-144) Newton straight from p0
    const bool initB = isI && !initN;         // (:145-148) bisection first
    const bool zero = (f == T(0.0));
    const bool bis_take = isB && !zero && jump;                                  // (:388-391) x = xmid; exit
    const bool bis_done = isB && (zero || jump || small || s.k == 20);           // (:375, :388, :394, maxiter)
    const bool bis_cont = isB && !bis_done;
    const T xb2 = bis_take ? s.xe : xb1;

    // solve trip (:269-311); for initN this evaluation is solve's first (x = p0)
    const bool newt = isN || initN;
    const T xN = initN ? s.xe : s.x;
    const int kN = initN ? 1 : s.k;
    const bool n_conv = newt && small;
    const bool n_go = newt && !small;
    T xnew = xN - q;                // x = x - deltax (:298)
    // the next trip of solve's loop would only confirm convergence at xnew (newton_lands_converged): it is not made
    const bool n_pred = use_pred && n_go && kN != 15 && newton_lands_converged(t, s, xN, f, q, xnew);
    if (xnew > s.vmi)
        xnew = s.clampv;                    // (:299-304)

    // solvebst entry (:354-365): x1 = 1d-10, x2 = 1/maxval(vp)-1d-12
    const T x2 = s.vmi - T(1e-12);
    const T d21 = x2 - T(1e-10);
    const T xb0 = neg ? T(1e-10) : x2;
    const T dx0 = neg ? d21 : -d21;  // x2-x1 | x1-x2 (exact negation)
    // dx = dx*0.5; xmid = x + dx (:368-369) for the first and for every further trip
    const bool halve = isB0 || bis_cont;
    const T dxh = (isB0 ? dx0 : s.dx) * T(0.5);
    const T xbh = isB0 ? xb0 : xb2;
    const T xmid = xbh + dxh;

    // ---- commit -----------------------------------------------------------------------------
    s.xb = isB0 ? xb0 : (isB ? xb2 : s.xb);
    s.dx = halve ? dxh : s.dx;
    s.x = n_go ? xnew : (n_conv ? xN : (bis_done ? xb2 : s.x));
    s.xe = initB ? T(1e-10) : (halve ? xmid : (bis_done ? xb2 : (n_go ? xnew : s.xe)));
    s.k = isB0 ? 1 : (bis_cont ? s.k + 1 : (bis_done ? 1 : ((n_go && kN != 15) ? kN + 1 : (initN ? 1 : s.k))));
    s.conv = (n_conv || n_pred) ? true : (isF ? (fabs(f) > T(0.1)) : s.conv);  // (:288) | the inverted test (:327-330)
    s.st = initB ? ST_BIS0
                 : (isB0 || bis_cont) ? ST_BIS
                                      : bis_done ? ST_NEWTON
                                                 : n_go ? (n_pred ? ST_DONE : (kN == 15 ? ST_FINAL : ST_NEWTON))
                                                        : (n_conv || isF) ? ST_DONE : st;
    if constexpr (MODE == ADV_ANY) {
        if (initB && s.skip0)  // f(x1) is provably positive: straight to the first midpoint (see lane_setup)
            lane_enter_bisection(s, false);
        // solvebst left through `fmid < 0` or the jump test: solve() starts at the point just evaluated, so its first
        // iteration is taken now (see lane_advance_solve)
        if (bis_done && (neg || (!zero && jump))) {
            if (small) {
                s.conv = true;
                s.st = ST_DONE;
            } else {
                T xn = xb2 - q;
                const bool pred = use_pred && newton_lands_converged(t, s, xb2, f, q, xn);
                if (xn > s.vmi)
                    xn = s.clampv;
                s.x = xn;
                s.xe = xn;
                s.k = 2;
                s.st = pred ? ST_DONE : ST_NEWTON;
                s.conv = pred ? true : s.conv;
            }
        }
    }
    if constexpr (MODE == ADV_NO_INIT) {
        const bool b0 = (st == ST_BIS0);  // untouched by everything above
        if (__any(b0)) {
            // solvebst entry (:354-365), then dx = dx*0.5; xmid = x + dx (:368-369)
            const T e2 = s.vmi - T(1e-12);
            const T e21 = e2 - T(1e-10);
            const T b_x = neg ? T(1e-10) : e2;
            const T b_dx = (neg ? e21 : -e21) * T(0.5);
            if (b0) {
                s.xb = b_x;
                s.dx = b_dx;
                s.xe = b_x + b_dx;
                s.k = 1;
                s.st = ST_BIS;
            }
        }
    }
}

// Bisection trips that provably change nothing.  solvebst started with f(x1) > 0 walks xmid_k = x2 - (x2 - x1) 2^-k towards
// the critical ray parameter; while f(xmid_k) >= tol (and f' < 0) a trip only halves dx: x stays x2, check = x2 - f/f' >
// x2 > 1/vmax - 1d-10, no exit (:367-397).  f is decreasing in x when no thickness is negative, so ONE rigorous lower
// bound f(xmid_m) > tol (costFunc's sum bounded from above through the table of stage_model, 1e-6 relative and 1e-4
// absolute margins over the ~1e-10 noise of the computed sums) proves trips 1..m idle, and the source enters the
// bisection at trip m + 1 with the state those trips would have left: dx = -(x2 - x1) 2^-(m+1) (halvings are exact),
// xmid = x2 + dx, trip counter m + 1.  m is first estimated in closed form, then VERIFIED at the solver's own xmid_m
// (up to three candidates, downwards); a failed verification just means no skip.  Returns m (0 = none).  Double precision only: in
// single precision the noise of the sums is of the order of tol itself.
template <typename T>
__device__ __forceinline__ int bisection_skip(const LayerTable<T> &t, const Lane<T> &s, int il, T hl)
{
    // margins: relative on the bound's own (approximate) arithmetic, absolute over the noise of the solver's computed
    // sum -- ~(n + 4) ulp of R + sum, i.e. 1e-10 in double but up to 1e-5 R in single precision for n <= 64 terms
    constexpr bool DBL = std::is_same<T, double>::value;
    const T relm = DBL ? T(1e-6) : T(1e-4);
    if (!t.SA)
        return 0;
    const T vm = t.VMAX[il];
    const bool own_crit = t.V[il] >= vm;
    const T A = t.SA[il] + (own_crit ? T(0.0) : hl * t.SC[il]);
    const T hc = t.HC[il] + (own_crit ? s.hvl : T(0.0));
    const T target = s.R - A * (T(1.0) + relm) - (T(0.1001) + (DBL ? T(1e-7) : T(3e-5)) * s.R);
    if (!(A < T(1e30)) || !(hc >= T(0.0)) || !(hl >= T(0.0)) || !(target > T(0.0)) || !(vm < T(1e7)) || !(s.R < T(1e12)))
        return 0;
    const T e2 = s.vmi - T(1e-12), e21 = e2 - T(1e-10);   // as lane_enter_bisection
    // approximate reciprocal / reciprocal square root, one Newton step each (~1e-9 relative in double, 1e-6 in single
    // precision: far inside the margins above; the bound decides which evaluations are skipped, never a value)
    auto rcp = [](T a) {
        T r;
        if constexpr (DBL)
            r = __builtin_amdgcn_rcp(a);
        else
            r = __builtin_amdgcn_rcpf(a);
        return r * (T(2.0) - a * r);
    };
    auto rsq = [](T a) {
        T r;
        if constexpr (DBL)
            r = __builtin_amdgcn_rsq(a);
        else
            r = __builtin_amdgcn_rsqf(a);
        return r * (T(1.5) - T(0.5) * a * r * r);
    };
    int m = 19;
    if (hc > T(0.0)) {
        // hc x / sqrt(1 - x^2 vm^2) <= target  <=>  1 - x vm >= g,  g = 1 / (s1 (s1 + tq)),  tq = target vm / hc
        const T tq = target * vm * rcp(hc);
        const T q2 = T(1.0) + tq * tq;
        const T s1 = q2 * rsq(q2);
        const T gap = rcp(s1 * (s1 + tq) * vm) - T(1e-12);             // x2 - xmid_m may be this small
        if (gap > T(0.0)) {
            const T ratio = e21 * rcp(gap);                             // 2^m <= ratio (estimate: verified below)
            int ex;
            (void)frexp(ratio, &ex);                                    // ratio = f 2^ex, f in [0.5, 1): floor(log2) = ex - 1
            m = ratio >= T(2.0) ? ex - 1 : 0;
            m = m > 19 ? 19 : m;
        }
    }
#pragma unroll 1
    for (int attempt = 0; attempt < 3 && m >= 1; ++attempt, --m) {
        const T x = e2 + ldexp(-e21, -m);                               // the solver's xmid of trip m
        const T y = x * vm, w = T(1.0) - y * y;
        if (!(w > (DBL ? T(1e-6) : T(1e-3))))
            continue;
        const T term = hc * x * rsq(w);
        if (term * (T(1.0) + relm) <= target)
            return m;
    }
    return 0;
}

// For a lane that has just entered solvebst with f(x1) > 0 (lane_enter_bisection(s, false)): skip the provably idle trips.
template <typename T>
__device__ __forceinline__ void lane_skip_idle_trips(const LayerTable<T> &t, Lane<T> &s, int il, T hl)
{
    const int m = bisection_skip(t, s, il, hl);
    if (m >= 1) {
        const T e2 = s.vmi - T(1e-12), e21 = e2 - T(1e-10);
        s.dx = ldexp(-e21, -(m + 1));
        s.xe = e2 + s.dx;
        s.k = m + 1;
    }
}

// lane_advance for a lane in ST_INIT (the fused kernel's first-evaluation pass): GetPTime's branch (:136-153) after
// cf = f(p0), cfp = f'(p0).  Same operations on the same operands as lane_advance.
template <typename T>
__device__ __forceinline__ void lane_advance_init(const LayerTable<T> &, Lane<T> &s, T f, T fp, bool live)
{
    const T q = solver_quotient(f, fp, live);  // cf/cfp (:138)
    const bool neg = f < T(0.0);
    const bool small = fabs(f) < T(0.1);       // abs(fx) < tol (:287) of solve's first trip
    const T check_x = s.xe - q;                // check_x = p0 - cf/cfp (:138)
    const bool newton = neg || check_x < s.safe;  // (:139-144) Newton straight from p0, else bisection first (:145-148)
    T xnew = s.xe - q;                         // solve's first trip: x = p0 - deltax (:298)
    const bool go = newton && !small;
    // (newton_lands_converged is not consulted here: one source in 700 is that close at p0, and the first-evaluation pass
    // of a small launch is on its critical path)
    if (xnew > s.vmi)
        xnew = s.clampv;                       // (:299-304)
    const bool conv = newton && small;
    s.x = go ? xnew : s.x;                     // converged at p0: x stays p0
    s.xe = go ? xnew : (newton ? s.xe : T(1e-10));
    s.k = go ? 2 : (newton ? 1 : s.k);
    s.conv = conv ? true : s.conv;
    s.st = conv ? ST_DONE : (go ? ST_NEWTON : ST_BIS0);
}

// lane_advance for the fused kernel's solve loop (states BIS0, BIS, NEWTON, FINAL; never INIT), one block per state,
// each skipped when no lane of the wave is in that state.  With the longest-first solve order the lanes of a wave
// are mostly in the same phase of their solves, so most trips execute one block; inside a block the updates are
// plain masked writes instead of chains of selects.  Same operations on the same operands as lane_advance.
// `use_pred` (wave-uniform): consult newton_lands_converged.  It pays where a round holds several records per lane (the
// evaluations it saves are throughput: config 4's share -2 %); a round of about one record per lane lasts as long as
// its longest source, and there the test only lengthens every trip (tempering slot +4 %), so the launcher leaves it out.
template <typename T>
__device__ __forceinline__ void lane_advance_solve(const LayerTable<T> &t, Lane<T> &s, T f, T fp, bool use_pred)
{
    const int st = s.st;
    const T q = solver_quotient(f, fp, st != ST_DONE);  // deltax of solvebst (:384) and of solve (:294)
    const bool neg = f < T(0.0);
    const bool small = fabs(f) < T(0.1);     // abs(fx) < tol (:287), abs(fmid) < tol (:394)
    // The Newton update of solve (:298-304), once for both states that take one: a lane in ST_NEWTON has x == xe (both are
    // set together, everywhere), and the trip in which solvebst leaves is solve's first trip at xmid = xe (below).
    T xnew = s.xe - q;                       // x = x - deltax (:298)
    // ... and the trip after it is not made when it would only confirm convergence at xnew (newton_lands_converged)
    bool lands = false;
    if (use_pred)
        lands = !small && newton_lands_converged(t, s, s.xe, f, q, xnew);
    if (xnew > s.vmi)
        xnew = s.clampv;                     // (:299-304)
    if (st == ST_BIS) {
        const T xb1 = neg ? s.xe : s.xb;     // `if (fmid < 0) x = xmid` (:371-373)
        const T check_x = xb1 - q;           // (:386)
        const bool jump = check_x < s.safe;
        const bool zero = (f == T(0.0));
        const bool done = zero || jump || small || s.k == 20;  // (:375, :388, :394, maxiter)
        const T xb2 = (!zero && jump) ? s.xe : xb1;            // (:388-391) x = xmid; exit
        const T dxh = s.dx * T(0.5);         // dx = dx*0.5; xmid = x + dx (:368-369)
        // solvebst returns x = xmid whenever it leaves through `fmid < 0` or the jump test; solve(x) then STARTS by
        // evaluating f and f' at that very point (:269-294) -- the values this trip has just computed, from the same
        // operands.  Its first iteration is therefore taken here (as lane_advance_init does for solve(p0)): one
        // evaluation less for nearly every bisection source.
        const bool fuse = done && (neg || (!zero && jump));
        const bool pred = fuse && lands;
        const bool f_conv = (fuse && small) || pred, f_go = fuse && !small;
        s.xb = xb2;
        s.dx = done ? s.dx : dxh;
        s.xe = f_go ? xnew : (done ? xb2 : xb2 + dxh);
        s.x = f_go ? xnew : (done ? xb2 : s.x);
        s.k = f_go ? 2 : (done ? 1 : s.k + 1);
        s.conv = f_conv ? true : s.conv;     // (:288)
        s.st = f_conv ? ST_DONE : (done ? ST_NEWTON : ST_BIS);
    } else if (st == ST_NEWTON) {
        const bool last = (s.k == 15);
        const bool pred = lands && !last;    // (the 15th trip is followed by the inverted test, not by a trip of the loop)
        s.conv = (small || pred) ? true : s.conv;  // (:288)
        s.x = small ? s.x : xnew;
        s.xe = small ? s.xe : xnew;
        s.k = (small || last) ? s.k : s.k + 1;
        s.st = (small || pred) ? ST_DONE : (last ? ST_FINAL : ST_NEWTON);
    } else if (st == ST_FINAL) {
        s.conv = fabs(f) > T(0.1);           // the inverted test (:327-330)
        s.st = ST_DONE;
    } else if (st == ST_BIS0) {
        // solvebst entry (:354-365): x1 = 1d-10, x2 = 1/maxval(vp)-1d-12, then dx = dx*0.5; xmid = x + dx
        lane_enter_bisection(s, neg);
    }
}

// Travel time for a known ray parameter (:156, :165-166): tt = sum H_i / (V_i * sqrt(1 - p^2 V_i^2)) over the
// fully crossed layers, then the partial layer.  Wave-uniform loop, two layers side by side, no lane masked: a
// lane whose stack has ended reads the "no layer" entry (H = 0, V = 1, VV = 0), whose term is 0 / 1.
// FAST: guarded sqrt and divide (same bits as the plain operators inside the guards); returns the guard.
template <bool FAST, typename T>
__device__ __forceinline__ bool travel_time_terms(const LayerTable<T> &t, T p, int nfull, int nfull_wave, int il,
                                                  T hl, T &tt_out)
{
    const T pp = p * p;
    T tt = T(0.0);
    bool ok = true;
    if constexpr (!(FAST && std::is_same<T, double>::value)) {
        // the plain operators (rare path in double precision): one layer per trip, small code
#pragma unroll 1
        for (int i = 0; i < nfull_wave; ++i) {
            const int j = i < nfull ? i : t.nul;
            tt = tt + div_ieee(t.H[j], T(t.V[j] * sqrt(T(1.0) - pp * t.VV[j])));
        }
        tt = tt + div_ieee(hl, T(t.V[il] * sqrt(T(1.0) - pp * t.VV[il])));
        tt_out = tt;
        return true;
    }
    auto pair = [&](T h0, T v0, T vv0, T h1, T v1, T vv1) {
        const T w0 = T(1.0) - pp * vv0, w1 = T(1.0) - pp * vv1;
        T t0, t1;
        if constexpr (FAST && std::is_same<T, double>::value) {
            // sqrt_guarded and div_guarded of both terms, statement by statement side by side
            const double y0 = __builtin_amdgcn_rsq(w0), y1 = __builtin_amdgcn_rsq(w1);
            double g0 = w0 * y0, g1 = w1 * y1;
            double k0 = y0 * 0.5, k1 = y1 * 0.5;
            const double e0 = __builtin_fma(-k0, g0, 0.5), e1 = __builtin_fma(-k1, g1, 0.5);
            g0 = __builtin_fma(g0, e0, g0);
            g1 = __builtin_fma(g1, e1, g1);
            k0 = __builtin_fma(k0, e0, k0);
            k1 = __builtin_fma(k1, e1, k1);
            double d0 = __builtin_fma(-g0, g0, w0), d1 = __builtin_fma(-g1, g1, w1);
            g0 = __builtin_fma(d0, k0, g0);
            g1 = __builtin_fma(d1, k1, g1);
            d0 = __builtin_fma(-g0, g0, w0);
            d1 = __builtin_fma(-g1, g1, w1);
            g0 = __builtin_fma(d0, k0, g0);  // cos_0
            g1 = __builtin_fma(d1, k1, g1);  // cos_1
            const double n0 = v0 * g0, n1 = v1 * g1;  // denominators V*cos
            double r0 = __builtin_amdgcn_rcp(n0), r1 = __builtin_amdgcn_rcp(n1);
            double c0 = __builtin_fma(-n0, r0, 1.0), c1 = __builtin_fma(-n1, r1, 1.0);
            r0 = __builtin_fma(r0, c0, r0);
            r1 = __builtin_fma(r1, c1, r1);
            c0 = __builtin_fma(-n0, r0, 1.0);
            c1 = __builtin_fma(-n1, r1, 1.0);
            r0 = __builtin_fma(r0, c0, r0);
            r1 = __builtin_fma(r1, c1, r1);
            c0 = __builtin_fma(-n0, r0, 1.0);  // third step: correctly rounded reciprocals (see div_guarded)
            c1 = __builtin_fma(-n1, r1, 1.0);
            r0 = __builtin_fma(r0, c0, r0);
            r1 = __builtin_fma(r1, c1, r1);
            const double q0 = h0 * r0, q1 = h1 * r1;
            t0 = __builtin_fma(__builtin_fma(-n0, q0, h0), r0, q0);
            t1 = __builtin_fma(__builtin_fma(-n1, q1, h1), r1, q1);
            ok = ok && w0 > 1e-60 && w1 > 1e-60 &&
                 max((uint32_t)__double2loint(n0), (uint32_t)__double2loint(n1)) != 0xFFFFFFFFu;
        } else {
            t0 = div_ieee(h0, T(v0 * sqrt(w0)));
            t1 = div_ieee(h1, T(v1 * sqrt(w1)));
        }
        tt = tt + t0;
        tt = tt + t1;
    };
    int i = 0;
    for (; i + 1 < nfull_wave; i += 2) {
        const int j0 = i < nfull ? i : t.nul, j1 = i + 1 < nfull ? i + 1 : t.nul;
        pair(t.H[j0], t.V[j0], t.VV[j0], t.H[j1], t.V[j1], t.VV[j1]);
    }
    // the odd full layer (or "no layer") together with the source's own partial layer
    const int j0 = (i < nfull_wave && i < nfull) ? i : t.nul;
    pair(t.H[j0], t.V[j0], t.VV[j0], hl, t.V[il], t.VV[il]);
    tt_out = tt;
    return ok;
}

// `tab_ok`: the stack's H and V are 0 / of moderate magnitude (HVOK of the deepest full layer and the source's
// own layer); `valid`: the lane carries a source (other lanes never veto the fast path).
template <typename T>
__device__ __forceinline__ T travel_time(const LayerTable<T> &t, T p, int nfull, int nfull_wave, int il, T z,
                                         bool valid = true)
{
    const T hl = z - t.D[il - 1];
    T tt;
    if constexpr (std::is_same<T, double>::value) {
        const bool tab_ok = (t.HVOK[il - 1] != T(0.0)) && mag_ok(hl) && mag_nz(t.V[il]);
        const bool ok = travel_time_terms<true>(t, p, nfull, nfull_wave, il, hl, tt) && tab_ok;
        if (__any(valid && !ok))
            travel_time_terms<false>(t, p, nfull, nfull_wave, il, hl, tt);
    } else {
        travel_time_terms<false>(t, p, nfull, nfull_wave, il, hl, tt);
    }
    return tt;
}

}  // namespace rtmi
#endif
