// rtmi_kernels.h -- internal interface between the C ABI (rtmi_capi.cpp) and the kernels.
#ifndef RTMI_KERNELS_H
#define RTMI_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RTMI_BLOCK 256          // threads per workgroup (4 waves: one per SIMD of a CU)
#define RTMI_BLOCKS_PER_CU 16   // grid cap = CUs x this; further tiles are grid-strided
#define RTMI_STAMPS_PER_WAVE 12
#define RTMI_STAMPS_PER_WG (8 + 16 * RTMI_STAMPS_PER_WAVE)  // diagnostic words per workgroup: 8 phase clocks + 12 per wave
#define RTMI_POOL 256         // slots of the hand-over ring (power of two)
#define RTMI_PROMOTE_MIN 4    // old sources a wave collects before one hand-over
#define RTMI_FUSED_MAX_BLOCK 1024  // fused kernel: one workgroup of up to 16 waves per CU
// Launches the group-of-lanes kernel takes (trace_small_kernel: double precision, one model, travel times without a residual
// sum): nsrc x G lanes, G = 8 / 16 / 32 / 64 >= nlayers + 1, fill the chip's 256 CUs x 1024 resident lanes at most once.  Measured
// with 8 interfaces (G = 16; profiles/r04ag/small_kernel_crossover.txt): 20 sources 19.7 us against 27.1 for a lane per source,
// 2 048: 22 / 30, 8 000 ... 16 384: 32 against 42 (a lane per source) and 44 (the fused kernel); 32 768: 54 against 44.5.
// RTMI_SMALL_KERNEL_LANES overrides (0: off).
#ifndef RTMI_SMALL_KERNEL_LANES
#define RTMI_SMALL_KERNEL_LANES 262144
#endif
#include <stdlib.h>
static inline int rtmi_small_group(int nlayers)
{
    const int n1 = nlayers + 1;
    return n1 <= 8 ? 8 : (n1 <= 16 ? 16 : (n1 <= 32 ? 32 : (n1 <= 64 ? 64 : 0)));
}
static inline bool rtmi_small_kernel_takes(int64_t nsrc, int nlayers)
{
    static const long long lanes = getenv("RTMI_SMALL_KERNEL_LANES") ? atoll(getenv("RTMI_SMALL_KERNEL_LANES")) : RTMI_SMALL_KERNEL_LANES;
    const int G = rtmi_small_group(nlayers);
    return G != 0 && nsrc >= 1 && (long long)nsrc * G <= lanes;
}

namespace rtmi {

// Kernel arguments of one trace launch.  Every pointer is a device pointer.  T = double (the reference's
// precision) or float (BASELINE config 5).
template <typename T>
struct TraceArgsT {
    const T *d_vels;      // [nmodels][ld_model]
    const T *d_depths;    // [nmodels][ld_model]
    const int32_t *d_nlayers;  // [nmodels] or NULL -> `nlayers` for every model
    int nlayers;
    int ld_model;
    const T *d_src_offset;  // [nsrc]
    const T *d_src_depth;   // [nsrc]
    int64_t nsrc;
    T *d_timeP;    // [nmodels][nsrc] or NULL
    T *d_p_final;  // [nsrc] or NULL (single model only)
    const T *d_obs;  // [nsrc] or NULL
    T *d_partial;    // [nmodels][grid_x] per-workgroup sums of squared residuals
    int max_nlayers; // with d_nlayers: every d_nlayers[m] must lie in [1, max_nlayers] (the LDS table is sized for it)
    int allow_skip;  // with d_nlayers: d_nlayers[m] == 0 marks an inactive row (no output, partial sums 0)
    int *d_status;   // sticky fault word (host-mapped) or NULL: bit 8 = a d_nlayers[m] out of range
    int skiptab;     // set by the launcher: the layer tables carry the bound of bisection_skip (models of <= 64 interfaces)
    int lane_per_source;  // 1: never the group-of-lanes kernel (rtmi_set_option("path", 1): the plain kernel, strictly)
};
typedef TraceArgsT<double> TraceArgs;

// Incremental likelihood of the tempering sampler (fused kernel, double precision, with d_obs): a proposal changes the
// model only from some depth `zthr` downwards, and a source above it has the stack, hence the travel time and the
// squared residual, it has under the chain's current model -- bit for bit, so that value is taken from r2cur instead of
// being traced again.  One record per model ROW, written by the sampler's step (a chain has one row per speculated
// proposal, rtmi_pt_dev.hpp):
//   zthr, dl1, dl2   the proposal being evaluated: sources with z < zthr, z != dl1, z != dl2 are unaffected (dl1 / dl2 =
//                    the last interface of the current / the proposed model: whichLayer's special case for a source exactly
//                    on the LAST interface, subroutineR-quiet.f90:22-30, makes such a source depend on which interface is last)
//   pending, p*      an accepted proposal of the row's CHAIN whose residuals still sit in a proposal row (pend_row): a source
//                    THAT proposal affected has its current residual there, not in r2cur; the chain's `mover` row copies them
//                    to r2cur (each source by the thread that loads it), the other rows of the chain only read
//   cur_row, pend_row, prop_row   rows (of nsrc doubles) of the residual slab d_r2: the chain's current residuals, the
//                    residuals of the pending proposal, where this row's own residuals go.  A chain's rows write a buffer
//                    that none of them reads in the same launch (the step alternates two sets of proposal rows)
//   n0, pn0          the handle's sources are SORTED BY DEPTH: sources [0, n0) lie above everything this proposal changes
//                    (z < min(zthr, dl1, dl2): never loaded, their residuals are added from the cache as a prefix), sources
//                    [0, pn0) above everything the pending proposal changed.  Conservative bounds from a depth histogram
//                    (multiples of 64); the per-source tests above decide for the rest
struct PtInc {
    double zthr, dl1, dl2;
    double pzthr, pdl1, pdl2;
    int pending;
    int mover;
    int cur_row, pend_row, prop_row;
    int n0, pn0;
    int pend_idx;   // > 0: the pending proposal's record is d_pinfo[pend_idx - 1] (its zthr, dl1, dl2, n0, prop_row stand for
                    // pzthr, pdl1, pdl2, pn0, pend_row): the speculative step names the accepted entry instead of copying it
};
// What a later launch needs to know about an ACCEPTED proposal whose residuals are still in its own row (PtInc::pend_idx):
// written when the proposal is made, one record per model row and per set of proposal rows.
struct PtPend {
    double zthr, dl1, dl2;
    int n0, prop_row;
};

// Arguments of the fused large-NSrc kernel.  Model m = blockIdx.y uses row m of d_vels / d_depths.
#define RTMI_SLOT_BYTES(T) (3 * sizeof(T) + 10)  // LDS per queued record: offset, depth, p + key, nl scratch (u32) + order (u16)
template <typename T>
struct FusedArgsT {
    const T *d_vels;        // [nmodels][ld_model]
    const T *d_depths;      // [nmodels][ld_model]
    const int32_t *d_nlayers;    // [nmodels] or NULL -> `nlayers` for every model
    int nlayers;
    int ld_model;
    int refill_min;              // idle lanes that trigger a refill of the wave (1..64)
    int slots;                   // records a workgroup holds in LDS per round (multiple of 64)
    int tail_group;              // end-game: wave group g stops refilling below g * tail_group queued records (0 = off)
    int rows;                    // set by the launcher: table form (rtmi_device.hpp): 0, 1 = row tables + constant stride, 2 = constant stride
    int skiptab;                 // set by the launcher: the layer tables carry the bound of bisection_skip (<= 64 interfaces)
    int promote_age;             // local waves hand a source to the express waves after this many trips (0 = off)
    int retire_max;              // drain: waves 4.. hand over their last <= retire_max live lanes and leave (0 = off)
    const T *d_src_offset;  // [nsrc]
    const T *d_src_depth;   // [nsrc]
    int64_t nsrc;
    T *d_timeP;             // [nmodels][nsrc] or NULL
    T *d_p_final;           // [nsrc] or NULL (single model only)
    const T *d_obs;         // [nsrc] or NULL
    T *d_partial;           // [nmodels][grid.x] per-workgroup sums of squared residuals
    int *d_status;                 // sticky fault word (host-mapped) or NULL: a bounded wait of the hand-over ring ran
                                   // out (bits 1, 2, 4) or a d_nlayers[m] was out of range (bit 8)
    int max_nlayers;               // with d_nlayers: every d_nlayers[m] must lie in [1, max_nlayers]
    int allow_skip;                // with d_nlayers: d_nlayers[m] == 0 marks an inactive row (no output, partial sums 0)
    unsigned long long *d_partial_x;  // [nmodels][part_stride][4] or NULL: the SAMPLER's form of the sums -- per workgroup an exact
                                      // order-independent fixed-point sum of squared residuals (rtmi_device.hpp::Acc192: words 0, 1; 2, 3 unused)
                                      // instead of d_partial; needs d_obs, no d_timeP, T = double, the fused kernel.  In this form
                                      // THE SOURCE SET MUST BE SORTED BY DEPTH (ascending, NaN last): the kernel does not sort
    const PtInc *d_inc;            // [nmodels] or NULL: incremental likelihood (see PtInc); needs d_partial_x
    T *d_r2;                       // residual slab of the incremental likelihood: rows of nsrc squared residuals (PtInc names the rows)
    unsigned long long *d_inc_counts;  // [nmodels][part_stride][2] or NULL: sources traced / taken from the cache, added up over the launches
    int part_stride;               // words of d_partial_x / d_inc_counts per model row are indexed [m][part_stride]; 0 = the launch's grid.x
    int acc_atomic;                // the sampler's form: a row's workgroups ADD their exact sums into d_partial_x[m][0] (three limbs of 48
                                   // bits + the flags, 64-bit atomics: integer addition commutes) instead of writing a slot each
    const PtPend *d_pinfo;         // records of accepted proposals (PtInc::pend_idx) or NULL
    int count_words;               // with acc_atomic: pairs of words d_inc_counts holds (one pair per workgroup of the launch, modulo this)
    unsigned long long *d_stamps;  // diagnostic clocks [gridDim.x][RTMI_STAMPS_PER_WG] or NULL
    int diag_fine;                 // with d_stamps: also the per-trip section clocks and counters
};
typedef FusedArgsT<double> FusedArgs;

// Arguments of the persistent sweep kernel of the tempering sampler (pt_sweep_kernel): every chain has gx workgroups,
// which run the chain's proposal slots one after the other and meet only among themselves.
struct PtSweepArgs {
    int nslots_k;          // node slots of this sweep: 1 birth/death slot + 2 nslots_k - 1 perturbation slots + the sigma move
    int gx;                // workgroups per chain
    int nchains;
    int xcd_map;           // 1: the workgroups of a chain are blockIdx values that land on one XCD (shared L2)
    unsigned int epoch;    // slots the handle has run through this kernel before this launch (the words below only grow)
    unsigned int *flag;    // [nchains] slots whose model row the chain's leading workgroup has published
    unsigned int *count;   // [nchains] workgroup arrivals (gx per slot)
    unsigned int *abort_;  // [1] set when a bounded wait ran out: every workgroup leaves at its next wait
    int lds_ctl;           // set by the launcher: byte offset of the kernel's own control word in dynamic LDS
    int nspec;             // > 0: the speculative step (rtmi_pt_dev.hpp::pt_step_spec) with this many model rows per chain; a
                           // chain runs passes until its sweep is complete (nslots_k only bounds them), `epoch` is a tag of
                           // the sweep (the passes of a sweep count from it, at most 4096)
    unsigned long long *stats;  // NULL, or [8] diagnostic sums (RTMI_PT_QSTATS): ticks in the likelihood, in steps, waiting; passes
    int *walloc;           // speculative form: [2][nchains] weights (slots of the chain's sweep) by which the launch's workgroups
    int parity;            // are dealt to the chains: read from walloc[parity], rewritten for the next sweep in walloc[1 - parity]
    int adapt;             // multi-sweep launches: a chain's weight for the next launch = its workgroups x the share of the launch it was not waiting
    int poison_lds;        // diagnostic (RTMI_PT_POISON_LDS): pt_sweep_prop_kernel fills its dynamic LDS with NaN patterns before it starts
    // ---- launches of SEVERAL sweeps with the temperature exchange inside the kernel (every chain of the ladder local, one chain
    // per level): a chain that has completed its sweep meets only its PARTNER of the round -- through a mailbox per level --
    // and goes on, so that a sweep costs what the chains take on average, not what the unluckiest takes (pt_sweep_prop_kernel)
    int nsweeps;                 // sweeps of this launch; <= 1: one sweep, exchange and sample record by the host's kernels
    int swap_every, record_every;
    int s0;                      // index of the launch's first sweep within the host's rtmi_pt_run call (record_every counts from there)
    long long sweeps_done;       // sweeps the handle had completed before this launch
    unsigned long long round;    // exchange rounds before this launch
    unsigned long long seed;     // of the exchange's draws (rtmi_pt_rules.h::swap_pair)
    int nct;                     // chains of the ladder = levels
    int *level;                  // [nct]
    const double *beta_of_level; // [nct]
    long long *swstats;          // [2] accepted, proposed
    double *mb_logl;             // [4][nct] mailbox of a level: logL of the chain that sits on it, after its sweep of the round
    int *mb_chain;               // [4][nct] ... and which chain that is
    unsigned long long *mb_seq;  // [4][nct] round + 1 once the entry is there (set round mod 4: an entry is overwritten four rounds later,
                                 //          by when its reader is provably through with it -- see prop_exchange_post)
    int s_cap;                   // sample ring (rtmi_pt_sample_buffer) or 0
    int *s_count;
    long long *s_meta;
    double *s_val;
};

// Arguments of the single-precision fast forward model (rtmi_f32fast.hip).  Sorted interfaces, positive velocities.
#define RTMI_F32FAST_MAX_NL 63
struct F32FastArgs {
    const float *d_vels;        // [nlayers + 1]
    const float *d_depths;      // [nlayers]
    int nlayers;
    const float *d_src_offset;  // [nsrc]
    const float *d_src_depth;   // [nsrc]
    int64_t nsrc;
    float *d_timeP;             // [nsrc]
    float *d_p_final;           // [nsrc] or NULL
    int *d_status;              // sticky fault word or NULL: bit 8 = the model is not usable (unsorted interfaces, a
                                // velocity <= 0, NaN): every output is NaN
};

}  // namespace rtmi

extern "C" {
hipError_t rtmi_launch_f32fast(const rtmi::F32FastArgs *args, int num_cus, hipStream_t stream);
size_t rtmi_f32fast_lds_bytes(int nlayers);
hipError_t rtmi_prepare_kernels();
int rtmi_fused_geometry(int64_t nsrc, int max_nlayers, int nmodels, int num_cus, int waves_per_simd, int elem_bytes,
                        int *threads, int *slots, size_t *lds_bytes);
hipError_t rtmi_launch_fused_f32(const rtmi::FusedArgsT<float> *args, int nmodels, int max_nlayers, int num_cus,
                                 int waves_per_simd, int *grid_x_out, hipStream_t stream);
hipError_t rtmi_launch_trace_f32(const rtmi::TraceArgsT<float> *args, int nmodels, int max_nlayers, int grid_x,
                                 hipStream_t stream);
hipError_t rtmi_launch_fused(const rtmi::FusedArgs *args, int nmodels, int max_nlayers, int num_cus,
                             int waves_per_simd, int *grid_x_out, hipStream_t stream);
// ptdev: DEVICE copy of the sampler's device view (rtmi_pt_dev::PtDev, ptdev_bytes = its size: checked).  Cooperative launch of
// nchains x w->gx workgroups; hipErrorCooperativeLaunchTooLarge (or any other error) = use the per-slot launches.
hipError_t rtmi_launch_pt_sweep(const rtmi::FusedArgs *args, const void *ptdev, size_t ptdev_bytes,
                                const rtmi::PtSweepArgs *w, int nchains, int max_nlayers, int num_cus,
                                int waves_per_simd, hipStream_t stream);
// The speculative sweep with the launch's workgroups dealt to the chains in proportion to their slots (pt_sweep_prop_kernel):
// cooperative launch of one workgroup per CU.  d_w: DEVICE copy of the constant words (flag, count, abort_, walloc, nchains,
// nspec); epoch (tag of the sweep), parity (which half of walloc to read) and nslots_k (bound of the passes) per launch.
hipError_t rtmi_launch_pt_prop(const rtmi::FusedArgs *args, const void *ptdev, size_t ptdev_bytes, const rtmi::PtSweepArgs *d_w,
                               int nspec, unsigned int epoch, int parity, int nslots_k, int nchains, int max_nlayers, int num_cus,
                               int waves_per_simd, hipStream_t stream);
hipError_t rtmi_launch_fastmath_check_f32(const float *d_w, const float *d_num, const float *d_den, int64_t n,
                                          unsigned long long *d_out, hipStream_t stream);
hipError_t rtmi_launch_fastmath_check(const double *d_w, const double *d_num, const double *d_den, int64_t n,
                                      unsigned long long *d_out, int extended, hipStream_t stream);
hipError_t rtmi_launch_fastmath_values(const double *d_w, const double *d_num, const double *d_den, int64_t n,
                                       double *o_div, double *o_divg, double *o_sqrt, double *o_tf, double *o_tp,
                                       hipStream_t stream);
hipError_t rtmi_launch_fastmath_values_f32(const float *d_w, const float *d_num, const float *d_den, int64_t n,
                                           float *o_div, float *o_tf, float *o_tp, hipStream_t stream);
int rtmi_launch_grid_x(int64_t nsrc, int nmodels, int num_cus);
hipError_t rtmi_launch_trace(const rtmi::TraceArgs *args, int nmodels, int max_nlayers, int grid_x,
                             hipStream_t stream);
hipError_t rtmi_launch_loglike(const double *d_partial, int nparts, const double *d_sd, int64_t ndat, int nmodels,
                               double *d_logL, double *d_ss, hipStream_t stream);
}

#endif
