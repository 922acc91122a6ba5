// rtmi_host.h -- host-side internals shared by the translation units of librtmi.so (rtmi_capi.cpp, rtmi_pt.hip,
// rtmi_comm.cpp).  Not part of the C ABI (that is include/rtmi.h).
#ifndef RTMI_HOST_H
#define RTMI_HOST_H

#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

#include "../../include/rtmi.h"
#include "rtmi_kernels.h"

namespace rtmi_host {

// Records the text rtmi_last_error() returns for the calling thread; returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return rtmi_host::fail(RTMI_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));           \
    } while (0)

// Per-process context of the host-pointer entry points (one bound device, one stream, grow-only buffers).
struct Context {
    std::mutex mu;
    bool ready = false;
    int device = 0;
    int cus = 0;
    hipStream_t stream = nullptr;
    // resident source set of the likelihood entry points (rtmi_set_sources)
    double *d_set = nullptr;    // offset | depth | obs
    size_t set_cap = 0;
    int64_t set_n = 0;
    bool set_has_obs = false;
    uint64_t set_generation = 0;  // bumped by every rtmi_set_sources
    double *d_lk = nullptr;     // models | sd | logL | ss | partial sums
    size_t lk_cap = 0;
    double *h_pin = nullptr;    // pinned staging of one batch (models in, logL/ss out)
    size_t pin_cap = 0;
    double *h_small = nullptr;  // pinned, device-mapped buffer of the small-call path of dff_ / TraceRays
    size_t small_cap = 0;
    // completion word of the small-call path: the stream writes a sequence number into it behind the kernel
    // (hipStreamWriteValue32) and the host spins on it instead of sleeping in hipStreamSynchronize
    volatile unsigned int *h_done = nullptr;
    unsigned int done_seq = 0;
    int done_ok = -1;           // -1 untried, 0 the stream memory operation is not available: hipStreamSynchronize, 1 in use
};
Context &ctx();
int ensure_ready_locked(Context &c);

// Scheduling knobs (never change results).  Read through tuning_snapshot(): rtmi_set_option may run concurrently.
struct Tuning {
    int force_path;      // 0 auto, 1 plain kernel, 2 fused kernel
    int solve_waves;     // resident waves per SIMD of the fused kernel
    int refill_min;      // idle lanes that trigger a refill
    int tail_group;      // end-game concentration (records per wave group), 0 = off
    int retire_max;      // drain: live lanes a retiring wave hands over (0 = off)
    int promote_age;     // trips after which a local wave hands a source to the express waves (0 = off)
    unsigned long long *d_stamps;  // diagnostic phase clocks of the fused kernel (rtmi_debug_stamps)
    int diag_fine;                 // ... including the per-trip clocks and counters
};
Tuning tuning_snapshot();
void set_path_override(int v);  // (thread-local) 2: route as if the fused path were forced, unless the caller forced one; 0: off
int path_override();

// Sticky fault word of the device code (pinned host memory mapped into the device).  `dev` is NULL until the first
// successful allocation; status_word_for(stream) refuses (returns NULL and records an error) when the word would have
// to be allocated while `stream` is capturing -- call rtmi_init (or any un-captured entry point) first.
struct StatusWord {
    int *host = nullptr;
    int *dev = nullptr;
};
const StatusWord &status_word();
int status_word_for(hipStream_t stream, int **dev_out);
int device_fault(int st);
int num_cus_for_current_device();

// One batched likelihood launch (travel times optional, per-workgroup residual sums into d_partial[nmodels][*gx]):
// picks the fused or the plain kernel like rtmi_trace_batch_device does.  max_nlayers bounds every d_nlayers[m];
// allow_skip: rows with d_nlayers[m] == 0 are inactive (their workgroups exit at once, partial sums 0).
int launch_batch(const double *d_vels, const double *d_depths, const int32_t *d_nlayers, int ld_model, int nmodels,
                 int max_nlayers, const double *d_src_offset, const double *d_src_depth, int64_t nsrc,
                 double *d_timeP, const double *d_obs, double *d_partial, bool allow_skip, int cus, int *gx_out,
                 hipStream_t stream, const rtmi::PtInc *d_inc = nullptr, double *d_r2 = nullptr,
                 unsigned long long *d_inc_counts = nullptr, unsigned long long *d_partial_x = nullptr, int part_stride = 0,
                 int acc_atomic = 0, const rtmi::PtPend *d_pinfo = nullptr);
// whether launch_batch takes the fused kernel for this many sources (current knobs)
bool batch_is_fused(int64_t nsrc);
// workgroups per model (= partial sums per model) launch_batch will use for this shape
int batch_grid_x(int nmodels, int max_nlayers, int64_t nsrc, int cus);
// The tempering sampler's sweep as one persistent cooperative launch (rtmi_kernels.hip::pt_sweep_kernel): the batched
// likelihood's arguments plus the sampler's device view (rtmi_pt_dev::PtDev, opaque here) and the sweep's
// rtmi::PtSweepArgs.  Non-zero = not available for this shape / device state: use the per-slot launches.
int launch_pt_sweep(const double *d_vels, const double *d_depths, const int32_t *d_nlayers, int ld_model, int nchains,
                    int max_nlayers, const double *d_src_offset, const double *d_src_depth, int64_t nsrc,
                    const double *d_obs, double *d_partial, int cus, const void *ptdev, size_t ptdev_bytes,
                    const void *sweep_args, hipStream_t stream, const rtmi::PtInc *d_inc = nullptr, double *d_r2 = nullptr,
                    unsigned long long *d_inc_counts = nullptr, unsigned long long *d_partial_x = nullptr, int part_stride = 0);

// The speculative sweep with the workgroups dealt to the chains in proportion to their slots (pt_sweep_prop_kernel): cooperative
// launch; d_sweep_args = DEVICE copy of rtmi::PtSweepArgs (flag / count / abort_, walloc, nchains, nspec).
int launch_pt_prop(const double *d_vels, const double *d_depths, const int32_t *d_nlayers, int ld_model, int nchains, int max_nlayers,
                   const double *d_src_offset, const double *d_src_depth, int64_t nsrc, const double *d_obs, int cus, const void *ptdev,
                   size_t ptdev_bytes, const void *d_sweep_args, int nspec, unsigned int epoch, int parity, int nslots_k, hipStream_t stream,
                   const rtmi::PtInc *d_inc, double *d_r2, unsigned long long *d_inc_counts, unsigned long long *d_partial_x,
                   int part_stride, const rtmi::PtPend *d_pinfo, int count_words);
}  // namespace rtmi_host

#endif
