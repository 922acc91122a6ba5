// rtmi_capi.cpp -- the C ABI of librtmi.so (include/rtmi.h).
//
// Host side of the drop-in boundary: owns the per-process device context, moves
// the caller's host buffers to HBM and back for the reference-shaped entry points
// (dff_, rtmi_tracerays, rtmi_dff7), and validates + forwards the device-resident
// entry points to the launchers in rtmi_kernels.hip.  There is NO CPU fallback:
// if the GPU path cannot run, the call fails loudly.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rtmi.h"
#include "rtmi_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(RTMI_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));              \
    } while (0)

#define RTMI_SMALL_CALL_NSRC 4096  // dff_ / TraceRays calls up to this size run on a mapped pinned buffer

// Per-process context of the host-pointer entry points.
struct Context {
    std::mutex mu;
    bool ready = false;
    int device = 0;
    int cus = 0;
    hipStream_t stream = nullptr;
    // grow-only device buffers of the drop-in path
    double *d_model = nullptr;  // vels | depths
    size_t model_cap = 0;
    double *d_src = nullptr;    // offset | depth | timeP | p_final
    size_t src_cap = 0;
    // resident source set of the likelihood entry points (rtmi_set_sources)
    double *d_set = nullptr;    // offset | depth | obs
    size_t set_cap = 0;
    int64_t set_n = 0;
    bool set_has_obs = false;
    double *d_lk = nullptr;     // models | sd | logL | ss | partial sums
    size_t lk_cap = 0;
    double *h_pin = nullptr;    // pinned staging of one batch (models in, logL/ss out)
    size_t pin_cap = 0;
    double *h_small = nullptr;  // pinned, device-mapped buffer of the small-call path of dff_ / TraceRays
    size_t small_cap = 0;
};

// Sticky fault word of the fused kernel: pinned host memory mapped into the device (written by a kernel only if one
// of its bounded waits runs out; read by the host for free).  Allocated once per process, by the first caller.
struct StatusWord {
    int *host = nullptr;
    int *dev = nullptr;
};
StatusWord &status_word()
{
    static StatusWord w = [] {
        StatusWord v;
        int *h = nullptr;
        if (hipHostMalloc((void **)&h, 64, hipHostMallocMapped) == hipSuccess && h) {
            *h = 0;
            void *d = nullptr;
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                v.host = h;
                v.dev = (int *)d;
            }
        }
        return v;
    }();
    return w;
}

int device_fault(int st)
{
    return fail(RTMI_ERR_HIP, "the fused kernel reported an internal scheduling fault (status %d): results are not valid", st);
}

// Tuning knobs of the fused kernel, overridable from the environment for A/B runs.
struct Tuning {
    int force_path;      // 0 auto, 1 plain kernel, 2 fused kernel
    int solve_waves;     // resident waves per SIMD of the fused kernel
    int refill_min;      // idle lanes that trigger a refill
    int tail_group;      // end-game concentration (records per wave group), 0 = off
    int retire_max;      // drain: live lanes a retiring wave hands over (0 = off)
    int promote_age;     // trips after which a local wave hands a source to the express waves (0 = off)
    unsigned long long *d_stamps = nullptr;  // diagnostic phase clocks of the fused kernel (rtmi_debug_stamps)
    int diag_fine = 1;                       // ... including the per-trip clocks and counters
};

Tuning &tuning()
{
    static Tuning t = [] {
        Tuning v{0, 4, 16, 256, 8, 10, nullptr};
        if (const char *e = getenv("RTMI_FORCE_PATH")) {
            if (!strcmp(e, "plain"))
                v.force_path = 1;
            else if (!strcmp(e, "fused"))
                v.force_path = 2;
        }
        if (const char *e = getenv("RTMI_SOLVE_WAVES"))
            v.solve_waves = atoi(e) > 0 ? atoi(e) : v.solve_waves;
        if (const char *e = getenv("RTMI_REFILL_MIN"))
            v.refill_min = atoi(e) > 0 ? atoi(e) : v.refill_min;
        if (const char *e = getenv("RTMI_TAIL_GROUP"))
            v.tail_group = atoi(e) >= 0 ? atoi(e) : v.tail_group;
        if (const char *e = getenv("RTMI_RETIRE_MAX"))
            v.retire_max = (atoi(e) >= 0 && atoi(e) <= 64) ? atoi(e) : v.retire_max;
        if (const char *e = getenv("RTMI_PROMOTE_AGE"))
            v.promote_age = (atoi(e) >= 0 && atoi(e) <= 64) ? atoi(e) : v.promote_age;
        return v;
    }();
    return t;
}

Context &ctx()
{
    static Context c;
    return c;
}

int ensure_ready_locked(Context &c)
{
    if (c.ready)
        return RTMI_OK;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(RTMI_ERR_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    if (c.device >= ndev)
        return fail(RTMI_ERR_NO_DEVICE, "device %d requested, %d visible", c.device, ndev);
    HIP_TRY(hipSetDevice(c.device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c.device));
    c.cus = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    HIP_TRY(rtmi_prepare_kernels());
    (void)status_word();  // one-time pinned allocation: here, not inside a caller's stream capture
    c.ready = true;
    return RTMI_OK;
}

int grow(double **p, size_t *cap, size_t need)
{
    if (need <= *cap)
        return RTMI_OK;
    if (*p)
        HIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    size_t n = need + need / 4 + 1024;
    HIP_TRY(hipMalloc((void **)p, n * sizeof(double)));
    *cap = n;
    return RTMI_OK;
}

int check_model(int nlayers)
{
    if (nlayers < 1 || nlayers > RTMI_MAX_LAYERS)
        return fail(RTMI_ERR_BAD_ARG, "nlayers = %d outside [1, %d]", nlayers, RTMI_MAX_LAYERS);
    return RTMI_OK;
}

int num_cus_for_current_device()
{
    Context &c = ctx();
    if (c.cus > 0)
        return c.cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return 256;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return 256;
    return cus;
}

// The reference-shaped call: host buffers in, host buffers out, optional rays.dat.
int trace_host(const double *vels, const double *depths, int nlayers, const double *src_offset,
               const double *src_depth, int nsrc, double *timeP, int keep_delta)
{
    if (!vels || !depths || (nsrc > 0 && (!src_offset || !src_depth || !timeP)))
        return fail(RTMI_ERR_BAD_ARG, "NULL argument");
    if (nsrc < 0)
        return fail(RTMI_ERR_BAD_ARG, "NSrc = %d < 0", nsrc);
    int rc = check_model(nlayers);
    if (rc)
        return rc;
    if (nsrc == 0)
        return RTMI_OK;
    Context &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    rc = ensure_ready_locked(c);
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(c.device));
    const bool want_p = keep_delta > 0;
    const size_t n = (size_t)nsrc;
    if (nsrc <= RTMI_SMALL_CALL_NSRC) {
        // Small calls (the reference's own example has NSrc = 20): the kernel reads the model and the sources from,
        // and writes the times to, ONE pinned host buffer mapped into the GPU's address space -- no copy commands,
        // the call costs one launch and one stream synchronise.
        const size_t nm = (size_t)2 * nlayers + 1;
        const size_t need = nm + 4 * n;
        if (c.small_cap < need) {
            if (c.h_small)
                HIP_TRY(hipHostFree(c.h_small));
            c.h_small = nullptr;
            c.small_cap = 0;
            HIP_TRY(hipHostMalloc((void **)&c.h_small, (need + 1024) * sizeof(double), hipHostMallocMapped));
            c.small_cap = need + 1024;
        }
        double *h = c.h_small, *dv = nullptr;
        HIP_TRY(hipHostGetDevicePointer((void **)&dv, h, 0));
        memcpy(h, vels, (size_t)(nlayers + 1) * sizeof(double));
        memcpy(h + nlayers + 1, depths, (size_t)nlayers * sizeof(double));
        memcpy(h + nm, src_offset, n * sizeof(double));
        memcpy(h + nm + n, src_depth, n * sizeof(double));
        rc = rtmi_trace_device(dv, dv + nlayers + 1, nlayers, dv + nm, dv + nm + n, nsrc, dv + nm + 2 * n,
                               want_p ? dv + nm + 3 * n : nullptr, c.stream);
        if (rc)
            return rc;
        HIP_TRY(hipStreamSynchronize(c.stream));
        if (int st = rtmi_device_status())
            return device_fault(st);
        memcpy(timeP, h + nm + 2 * n, n * sizeof(double));
        if (want_p)
            return rtmi_write_rays_dat("rays.dat", vels, depths, nlayers, src_offset, src_depth, h + nm + 3 * n, nsrc);
        return RTMI_OK;
    }
    rc = grow(&c.d_model, &c.model_cap, (size_t)2 * nlayers + 1);
    if (rc)
        return rc;
    rc = grow(&c.d_src, &c.src_cap, 4 * n);
    if (rc)
        return rc;
    double *d_vels = c.d_model, *d_depths = c.d_model + nlayers + 1;
    double *d_off = c.d_src, *d_dep = c.d_src + n, *d_t = c.d_src + 2 * n, *d_p = c.d_src + 3 * n;
    HIP_TRY(hipMemcpyAsync(d_vels, vels, (size_t)(nlayers + 1) * sizeof(double), hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipMemcpyAsync(d_depths, depths, (size_t)nlayers * sizeof(double), hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipMemcpyAsync(d_off, src_offset, n * sizeof(double), hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipMemcpyAsync(d_dep, src_depth, n * sizeof(double), hipMemcpyHostToDevice, c.stream));
    rc = rtmi_trace_device(d_vels, d_depths, nlayers, d_off, d_dep, nsrc, d_t, want_p ? d_p : nullptr, c.stream);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(timeP, d_t, n * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    std::vector<double> p;
    if (want_p) {
        p.resize(n);
        HIP_TRY(hipMemcpyAsync(p.data(), d_p, n * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    }
    HIP_TRY(hipStreamSynchronize(c.stream));
    if (int st = rtmi_device_status())
        return device_fault(st);
    if (want_p)
        return rtmi_write_rays_dat("rays.dat", vels, depths, nlayers, src_offset, src_depth, p.data(), nsrc);
    return RTMI_OK;
}

// The reference API has no status channel: fail loudly.
void dropin_failed(const char *who, double *timeP, int nsrc)
{
    fprintf(stderr, "librtmi: %s failed: %s\n", who, g_err.c_str());
    if (timeP)
        for (int i = 0; i < nsrc; ++i)
            timeP[i] = NAN;
    const char *e = getenv("RTMI_ABORT_ON_ERROR");
    if (!e || strcmp(e, "0") != 0)
        abort();
}

// gfortran list-directed REAL(8): 17 significant digits, F form for 0.1 <= |x| < 1e17, else E form.
void format_real(char *out, size_t cap, double x)
{
    const double ax = fabs(x);
    if (x == 0.0) {
        snprintf(out, cap, "%21s     ", "0.0000000000000000");
        return;
    }
    if (std::isfinite(x) && ax >= 0.1 && ax < 1e17) {
        int int_digits = (ax < 1.0) ? 0 : (int)floor(log10(ax)) + 1;
        int dec = 17 - int_digits;
        if (dec < 0)
            dec = 0;
        char num[64];
        snprintf(num, sizeof num, "%.*f", dec, x);
        // rounding may have added a digit (9.99.. -> 10.0..): recompute once
        int int_digits2 = 0;
        for (const char *q = (num[0] == '-') ? num + 1 : num; *q && *q != '.'; ++q)
            ++int_digits2;
        if (ax >= 1.0 && int_digits2 != int_digits) {
            dec = 17 - int_digits2;
            if (dec < 0)
                dec = 0;
            snprintf(num, sizeof num, "%.*f", dec, x);
        }
        snprintf(out, cap, "%21s     ", num);
    } else {
        char num[64];
        snprintf(num, sizeof num, "%.16E", x);
        snprintf(out, cap, "%26s", num);
    }
}

}  // namespace

extern "C" {

int rtmi_version(void) { return RTMI_VERSION; }

const char *rtmi_last_error(void) { return g_err.c_str(); }

int rtmi_init(int device)
{
    Context &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.ready && c.device != device) {
        // rebind: drop the old context's buffers
        (void)hipSetDevice(c.device);
        if (c.d_model)
            (void)hipFree(c.d_model);
        if (c.d_src)
            (void)hipFree(c.d_src);
        if (c.d_set)
            (void)hipFree(c.d_set);
        if (c.d_lk)
            (void)hipFree(c.d_lk);
        if (c.h_pin)
            (void)hipHostFree(c.h_pin);
        if (c.h_small)
            (void)hipHostFree(c.h_small);
        c.d_set = c.d_lk = c.h_pin = c.h_small = nullptr;
        c.set_cap = c.lk_cap = c.pin_cap = c.small_cap = 0;
        c.set_n = 0;
        c.set_has_obs = false;
        if (c.stream)
            (void)hipStreamDestroy(c.stream);
        c.d_model = c.d_src = nullptr;
        c.model_cap = c.src_cap = 0;
        c.stream = nullptr;
        c.cus = 0;
        c.ready = false;
    }
    c.device = device;
    return ensure_ready_locked(c);
}

int rtmi_set_option(const char *name, int value)
{
    if (!name)
        return fail(RTMI_ERR_BAD_ARG, "NULL option name");
    Tuning &t = tuning();
    if (!strcmp(name, "path")) {
        if (value < 0 || value > 2)
            return fail(RTMI_ERR_BAD_ARG, "path must be 0 (auto), 1 (plain) or 2 (fused)");
        t.force_path = value;
    } else if (!strcmp(name, "waves_per_simd")) {
        if (value < 1 || value > 8)
            return fail(RTMI_ERR_BAD_ARG, "waves_per_simd must be in [1, 8]");
        t.solve_waves = value;
    } else if (!strcmp(name, "refill_min")) {
        if (value < 1 || value > 64)
            return fail(RTMI_ERR_BAD_ARG, "refill_min must be in [1, 64]");
        t.refill_min = value;
    } else if (!strcmp(name, "diag_fine")) {
        t.diag_fine = value != 0;
    } else if (!strcmp(name, "promote_age")) {
        if (value < 0 || value > 64)
            return fail(RTMI_ERR_BAD_ARG, "promote_age must be in [0, 64]");
        t.promote_age = value;
    } else if (!strcmp(name, "retire_max")) {
        if (value < 0 || value > 64)
            return fail(RTMI_ERR_BAD_ARG, "retire_max must be in [0, 64]");
        t.retire_max = value;
    } else if (!strcmp(name, "tail_group")) {
        if (value < 0 || value > 4096)
            return fail(RTMI_ERR_BAD_ARG, "tail_group must be in [0, 4096]");
        t.tail_group = value;
    } else {
        return fail(RTMI_ERR_BAD_ARG, "unknown option %s", name);
    }
    return RTMI_OK;
}

int rtmi_device_status(void)
{
    const StatusWord &w = status_word();
    return w.host ? *(volatile int *)w.host : 0;
}

int rtmi_debug_stamps(unsigned long long *d_stamps)
{
    tuning().d_stamps = d_stamps;
    return RTMI_OK;
}

int rtmi_device_cus(void)
{
    Context &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (ensure_ready_locked(c) != RTMI_OK)
        return 0;
    return c.cus;
}

int rtmi_trace_device(const double *d_vels, const double *d_depths, int nlayers, const double *d_src_offset,
                      const double *d_src_depth, int64_t nsrc, double *d_timeP, double *d_p_final, void *stream)
{
    int rc = check_model(nlayers);
    if (rc)
        return rc;
    if (nsrc < 0)
        return fail(RTMI_ERR_BAD_ARG, "nsrc < 0");
    if (nsrc == 0)
        return RTMI_OK;
    if (!d_vels || !d_depths || !d_src_offset || !d_src_depth || !d_timeP)
        return fail(RTMI_ERR_BAD_ARG, "NULL device pointer");
    const Tuning tn = tuning();
    const bool fused = tn.force_path != 1 && (nsrc >= RTMI_FUSED_MIN_NSRC || tn.force_path == 2);
    if (fused) {
        rtmi::FusedArgs fa;
        memset(&fa, 0, sizeof fa);
        fa.d_vels = d_vels;
        fa.d_depths = d_depths;
        fa.nlayers = nlayers;
        fa.refill_min = tn.refill_min;
        fa.tail_group = tn.tail_group;
        fa.retire_max = tn.retire_max;
        fa.d_status = status_word().dev;
        fa.promote_age = tn.promote_age;
        fa.d_src_offset = d_src_offset;
        fa.d_src_depth = d_src_depth;
        fa.nsrc = nsrc;
        fa.d_timeP = d_timeP;
        fa.d_p_final = d_p_final;
        fa.d_stamps = tn.d_stamps;
        fa.diag_fine = tn.diag_fine;
        HIP_TRY(rtmi_launch_fused(&fa, 1, nlayers, num_cus_for_current_device(), tn.solve_waves, nullptr,
                                  (hipStream_t)stream));
        return RTMI_OK;
    }
    rtmi::TraceArgs a;
    memset(&a, 0, sizeof a);
    a.d_vels = d_vels;
    a.d_depths = d_depths;
    a.d_nlayers = nullptr;
    a.nlayers = nlayers;
    a.ld_model = 0;
    a.d_src_offset = d_src_offset;
    a.d_src_depth = d_src_depth;
    a.nsrc = nsrc;
    a.d_timeP = d_timeP;
    a.d_p_final = d_p_final;
    const int gx = rtmi_launch_grid_x(nsrc, 1, num_cus_for_current_device());
    HIP_TRY(rtmi_launch_trace(&a, 1, nlayers, gx, (hipStream_t)stream));
    return RTMI_OK;
}

int rtmi_selftest_fastmath(const double *d_w, const double *d_num, const double *d_den, int64_t n,
                           unsigned long long *d_mismatch, void *stream)
{
    if (n < 0 || !d_w || !d_num || !d_den || !d_mismatch)
        return fail(RTMI_ERR_BAD_ARG, "bad argument");
    if (n == 0)
        return RTMI_OK;
    HIP_TRY(rtmi_launch_fastmath_check(d_w, d_num, d_den, n, d_mismatch, (hipStream_t)stream));
    return RTMI_OK;
}

int rtmi_selftest_fastmath_f32(const float *d_w, const float *d_num, const float *d_den, int64_t n,
                               unsigned long long *d_mismatch, void *stream)
{
    if (n < 0 || !d_w || !d_num || !d_den || !d_mismatch)
        return fail(RTMI_ERR_BAD_ARG, "bad argument");
    if (n == 0)
        return RTMI_OK;
    HIP_TRY(rtmi_launch_fastmath_check_f32(d_w, d_num, d_den, n, d_mismatch, (hipStream_t)stream));
    return RTMI_OK;
}

int rtmi_set_sources(const double *src_offset, const double *src_depth, const double *obs, int64_t nsrc)
{
    if (nsrc < 1 || !src_offset || !src_depth)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_set_sources: nsrc < 1 or NULL pointer");
    Context &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    int rc = ensure_ready_locked(c);
    if (rc)
        return rc;
    HIP_TRY(hipSetDevice(c.device));
    const size_t n = (size_t)nsrc;
    rc = grow(&c.d_set, &c.set_cap, 3 * n);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(c.d_set, src_offset, n * sizeof(double), hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipMemcpyAsync(c.d_set + n, src_depth, n * sizeof(double), hipMemcpyHostToDevice, c.stream));
    if (obs)
        HIP_TRY(hipMemcpyAsync(c.d_set + 2 * n, obs, n * sizeof(double), hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    if (int st = rtmi_device_status())
        return device_fault(st);
    c.set_n = nsrc;
    c.set_has_obs = obs != nullptr;
    return RTMI_OK;
}

int rtmi_loglhood_batch(const double *vels, const double *depths, const int32_t *nlayers, int ld_model, int nmodels,
                        const double *sd, double *logL, double *ss)
{
    if (nmodels < 0)
        return fail(RTMI_ERR_BAD_ARG, "nmodels < 0");
    if (nmodels == 0)
        return RTMI_OK;
    if (!vels || !depths || !nlayers || !sd || !logL)
        return fail(RTMI_ERR_BAD_ARG, "NULL argument");
    if (ld_model < 2 || ld_model - 1 > RTMI_MAX_LAYERS)
        return fail(RTMI_ERR_BAD_ARG, "ld_model = %d outside [2, %d]", ld_model, RTMI_MAX_LAYERS + 1);
    for (int m = 0; m < nmodels; ++m)
        if (nlayers[m] < 1 || nlayers[m] > ld_model - 1)
            return fail(RTMI_ERR_BAD_ARG, "nlayers[%d] = %d outside [1, ld_model - 1]", m, nlayers[m]);
    Context &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (!c.ready || c.set_n < 1 || !c.set_has_obs)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_loglhood_batch: call rtmi_set_sources with observations first");
    HIP_TRY(hipSetDevice(c.device));
    const size_t nm = (size_t)nmodels, ld = (size_t)ld_model;
    const size_t n_in = 2 * nm * ld + nm + (nm + 1) / 2;  // vels | depths | sd | nlayers (i32 packed)
    const size_t n_out = 2 * nm;                           // logL | ss
    const int64_t wbytes = rtmi_trace_batch_work_bytes(nmodels, c.set_n);
    int rc = grow(&c.d_lk, &c.lk_cap, n_in + n_out + (size_t)wbytes / 8 + 8);
    if (rc)
        return rc;
    if (c.pin_cap < n_in + n_out) {
        if (c.h_pin)
            HIP_TRY(hipHostFree(c.h_pin));
        c.h_pin = nullptr;
        c.pin_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&c.h_pin, (n_in + n_out + 1024) * sizeof(double), hipHostMallocDefault));
        c.pin_cap = n_in + n_out + 1024;
    }
    double *h = c.h_pin;
    memcpy(h, vels, nm * ld * sizeof(double));
    memcpy(h + nm * ld, depths, nm * ld * sizeof(double));
    memcpy(h + 2 * nm * ld, sd, nm * sizeof(double));
    memcpy(h + 2 * nm * ld + nm, nlayers, nm * sizeof(int32_t));
    double *d_v = c.d_lk, *d_d = d_v + nm * ld, *d_sd = d_d + nm * ld;
    int32_t *d_nl = reinterpret_cast<int32_t *>(d_sd + nm);
    double *d_out = c.d_lk + n_in;  // logL | ss
    double *d_work = d_out + n_out;
    HIP_TRY(hipMemcpyAsync(c.d_lk, h, n_in * sizeof(double), hipMemcpyHostToDevice, c.stream));
    const size_t n = (size_t)c.set_n;
    // travel times are not materialised: only the residual sums leave the kernel
    rtmi::FusedArgs fa;
    memset(&fa, 0, sizeof fa);
    const Tuning tn = tuning();
    int gx = 0;
    if (tn.force_path != 1 && (c.set_n >= RTMI_FUSED_MIN_NSRC / 4 || tn.force_path == 2)) {
        fa.d_vels = d_v;
        fa.d_depths = d_d;
        fa.d_nlayers = d_nl;
        fa.ld_model = ld_model;
        fa.refill_min = tn.refill_min;
        fa.tail_group = tn.tail_group;
        fa.retire_max = tn.retire_max;
        fa.d_status = status_word().dev;
        fa.promote_age = tn.promote_age;
        fa.d_src_offset = c.d_set;
        fa.d_src_depth = c.d_set + n;
        fa.nsrc = c.set_n;
        fa.d_obs = c.d_set + 2 * n;
        fa.d_partial = d_work;
        HIP_TRY(rtmi_launch_fused(&fa, nmodels, ld_model - 1, c.cus, tn.solve_waves, &gx, c.stream));
    } else {
        rtmi::TraceArgs a;
        memset(&a, 0, sizeof a);
        a.d_vels = d_v;
        a.d_depths = d_d;
        a.d_nlayers = d_nl;
        a.ld_model = ld_model;
        a.d_src_offset = c.d_set;
        a.d_src_depth = c.d_set + n;
        a.nsrc = c.set_n;
        a.d_obs = c.d_set + 2 * n;
        a.d_partial = d_work;
        gx = rtmi_launch_grid_x(c.set_n, nmodels, c.cus);
        HIP_TRY(rtmi_launch_trace(&a, nmodels, ld_model - 1, gx, c.stream));
    }
    HIP_TRY(rtmi_launch_loglike(d_work, gx, d_sd, c.set_n, nmodels, d_out, d_out + nm, c.stream));
    HIP_TRY(hipMemcpyAsync(h + n_in, d_out, n_out * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    if (int st = rtmi_device_status())
        return device_fault(st);
    memcpy(logL, h + n_in, nm * sizeof(double));
    if (ss)
        memcpy(ss, h + n_in + nm, nm * sizeof(double));
    return RTMI_OK;
}

int rtmi_trace_device_f32(const float *d_vels, const float *d_depths, int nlayers, const float *d_src_offset,
                          const float *d_src_depth, int64_t nsrc, float *d_timeP, float *d_p_final, void *stream)
{
    int rc = check_model(nlayers);
    if (rc)
        return rc;
    if (nsrc < 0)
        return fail(RTMI_ERR_BAD_ARG, "nsrc < 0");
    if (nsrc == 0)
        return RTMI_OK;
    if (!d_vels || !d_depths || !d_src_offset || !d_src_depth || !d_timeP)
        return fail(RTMI_ERR_BAD_ARG, "NULL device pointer");
    const Tuning tn = tuning();
    const bool fused = tn.force_path != 1 && (nsrc >= RTMI_FUSED_MIN_NSRC || tn.force_path == 2);
    if (fused) {
        rtmi::FusedArgsT<float> fa;
        memset(&fa, 0, sizeof fa);
        fa.d_vels = d_vels;
        fa.d_depths = d_depths;
        fa.nlayers = nlayers;
        fa.refill_min = tn.refill_min;
        fa.tail_group = tn.tail_group;
        fa.retire_max = tn.retire_max;
        fa.d_status = status_word().dev;
        fa.promote_age = tn.promote_age;
        fa.d_src_offset = d_src_offset;
        fa.d_src_depth = d_src_depth;
        fa.nsrc = nsrc;
        fa.d_timeP = d_timeP;
        fa.d_p_final = d_p_final;
        HIP_TRY(rtmi_launch_fused_f32(&fa, 1, nlayers, num_cus_for_current_device(), tn.solve_waves, nullptr,
                                      (hipStream_t)stream));
        return RTMI_OK;
    }
    rtmi::TraceArgsT<float> a;
    memset(&a, 0, sizeof a);
    a.d_vels = d_vels;
    a.d_depths = d_depths;
    a.nlayers = nlayers;
    a.d_src_offset = d_src_offset;
    a.d_src_depth = d_src_depth;
    a.nsrc = nsrc;
    a.d_timeP = d_timeP;
    a.d_p_final = d_p_final;
    const int gx = rtmi_launch_grid_x(nsrc, 1, num_cus_for_current_device());
    HIP_TRY(rtmi_launch_trace_f32(&a, 1, nlayers, gx, (hipStream_t)stream));
    return RTMI_OK;
}

int64_t rtmi_trace_batch_work_bytes(int nmodels, int64_t nsrc)
{
    if (nmodels < 1 || nsrc < 0)
        return 0;
    // enough for either kernel's per-workgroup partial sums
    const int cus = num_cus_for_current_device();
    int64_t gx = rtmi_launch_grid_x(nsrc, nmodels, cus);
    if (gx < 2 * cus)
        gx = 2 * cus;
    return (int64_t)nmodels * gx * (int64_t)sizeof(double);
}

int rtmi_trace_batch_device(const double *d_vels, const double *d_depths, const int32_t *d_nlayers, int ld_model,
                            int nmodels, const double *d_src_offset, const double *d_src_depth, int64_t nsrc,
                            double *d_timeP, const double *d_obs, const double *d_sd, double *d_logL, void *d_work,
                            void *stream)
{
    if (nmodels < 0 || nsrc < 0)
        return fail(RTMI_ERR_BAD_ARG, "negative count");
    if (nmodels == 0 || nsrc == 0)
        return RTMI_OK;
    if (ld_model < 2 || ld_model - 1 > RTMI_MAX_LAYERS)
        return fail(RTMI_ERR_BAD_ARG, "ld_model = %d outside [2, %d]", ld_model, RTMI_MAX_LAYERS + 1);
    if (nmodels > 65535)
        return fail(RTMI_ERR_BAD_ARG, "nmodels = %d > 65535 (grid.y)", nmodels);
    if (!d_vels || !d_depths || !d_nlayers || !d_src_offset || !d_src_depth)
        return fail(RTMI_ERR_BAD_ARG, "NULL device pointer");
    const bool want_ll = d_obs || d_sd || d_logL;
    if (want_ll && (!d_obs || !d_sd || !d_logL || !d_work))
        return fail(RTMI_ERR_BAD_ARG, "logL needs d_obs, d_sd, d_logL and d_work together");
    if (!want_ll && !d_timeP)
        return fail(RTMI_ERR_BAD_ARG, "nothing to compute: d_timeP and d_logL both NULL");
    rtmi::TraceArgs a;
    memset(&a, 0, sizeof a);
    a.d_vels = d_vels;
    a.d_depths = d_depths;
    a.d_nlayers = d_nlayers;
    a.nlayers = 0;
    a.ld_model = ld_model;
    a.d_src_offset = d_src_offset;
    a.d_src_depth = d_src_depth;
    a.nsrc = nsrc;
    a.d_timeP = d_timeP;
    a.d_p_final = nullptr;
    a.d_obs = want_ll ? d_obs : nullptr;
    a.d_partial = (double *)d_work;
    const Tuning tn = tuning();
    int gx = 0;
    const bool fused = tn.force_path != 1 && (nsrc >= RTMI_FUSED_MIN_NSRC / 4 || tn.force_path == 2);
    if (fused) {
        rtmi::FusedArgs fa;
        memset(&fa, 0, sizeof fa);
        fa.d_vels = d_vels;
        fa.d_depths = d_depths;
        fa.d_nlayers = d_nlayers;
        fa.ld_model = ld_model;
        fa.refill_min = tn.refill_min;
        fa.tail_group = tn.tail_group;
        fa.retire_max = tn.retire_max;
        fa.d_status = status_word().dev;
        fa.promote_age = tn.promote_age;
        fa.d_src_offset = d_src_offset;
        fa.d_src_depth = d_src_depth;
        fa.nsrc = nsrc;
        fa.d_timeP = d_timeP;
        fa.d_obs = want_ll ? d_obs : nullptr;
        fa.d_partial = (double *)d_work;
        HIP_TRY(rtmi_launch_fused(&fa, nmodels, ld_model - 1, num_cus_for_current_device(), tn.solve_waves, &gx,
                                  (hipStream_t)stream));
    } else {
        gx = rtmi_launch_grid_x(nsrc, nmodels, num_cus_for_current_device());
        HIP_TRY(rtmi_launch_trace(&a, nmodels, ld_model - 1, gx, (hipStream_t)stream));
    }
    if (want_ll)
        HIP_TRY(rtmi_launch_loglike((const double *)d_work, gx, d_sd, nsrc, nmodels, d_logL, nullptr,
                                    (hipStream_t)stream));
    return RTMI_OK;
}

void dff_(const double *vels, const double *depths, const int *NLayers, double *src_offset, double *src_depth,
          const int *NSrc, double *timeP, const int *keep_delta)
{
    if (!NLayers || !NSrc) {
        fail(RTMI_ERR_BAD_ARG, "NULL NLayers/NSrc");
        dropin_failed("dff_", nullptr, 0);
        return;
    }
    const int kd = keep_delta ? *keep_delta : -1;
    if (trace_host(vels, depths, *NLayers, src_offset, src_depth, *NSrc, timeP, kd) != RTMI_OK)
        dropin_failed("dff_", timeP, *NSrc);
}

void rtmi_tracerays(const double *vels, const double *depths, const int *NLayers, double *src_offset,
                    double *src_depth, const int *NSrc, double *timeP, const int *keep_delta)
{
    if (!NLayers || !NSrc) {
        fail(RTMI_ERR_BAD_ARG, "NULL NLayers/NSrc");
        dropin_failed("TraceRays", nullptr, 0);
        return;
    }
    const int kd = keep_delta ? *keep_delta : -1;
    if (trace_host(vels, depths, *NLayers, src_offset, src_depth, *NSrc, timeP, kd) != RTMI_OK)
        dropin_failed("TraceRays", timeP, *NSrc);
}

void rtmi_dff7(const double *vels, const double *depths, const int *NLayers, double *src_offset, double *src_depth,
               const int *NSrc, double *timeP)
{
    if (!NLayers || !NSrc) {
        fail(RTMI_ERR_BAD_ARG, "NULL NLayers/NSrc");
        dropin_failed("dff_ (7-argument)", nullptr, 0);
        return;
    }
    if (trace_host(vels, depths, *NLayers, src_offset, src_depth, *NSrc, timeP, -1) != RTMI_OK)
        dropin_failed("dff_ (7-argument)", timeP, *NSrc);
}

int rtmi_write_rays_dat(const char *path, const double *vels, const double *depths, int nlayers,
                        const double *src_offset, const double *src_depth, const double *p_final, int64_t nsrc)
{
    int rc = check_model(nlayers);
    if (rc)
        return rc;
    FILE *fp = fopen(path, "w");  // status='REPLACE'  (subroutineR-quiet.f90:432)
    if (!fp)
        return fail(RTMI_ERR_IO, "cannot open %s for writing", path);
    std::vector<double> d((size_t)nlayers + 1), h((size_t)nlayers + 1);
    char cell[64];
    for (int64_t k = 0; k < nsrc; ++k) {
        const double z = src_depth[k];
        // whichLayer (:9-32)
        int nl = 0;
        double diff = 0.0;
        for (int i = 1; i <= nlayers; ++i) {
            nl = i;
            diff = depths[i - 1] - z;
            if (diff > 0.0)
                break;
        }
        if (diff < 0.0)
            nl = nlayers + 1;
        int nseg;
        if (nl == 1) {
            // :98-105 -- `d` has nl_src = 1 element = the offset; then the source depth
            d[0] = src_offset[k];
            h[0] = z;
            nseg = 1;
        } else {
            const double p = p_final[k];
            for (int i = 0; i < nl; ++i) {
                double Hi;
                if (i == 0)
                    Hi = depths[0];
                else if (i < nl - 1)
                    Hi = depths[i] - depths[i - 1];
                else
                    Hi = z - depths[nl - 2];
                const double cosv = sqrt(1.0 - (p * p) * (vels[i] * vels[i]));
                const double q = Hi / cosv;
                d[i] = sqrt(q * q - Hi * Hi);  // :158
                h[i] = Hi;
            }
            nseg = nl;
        }
        for (int i = 0; i < nseg; ++i) {
            format_real(cell, sizeof cell, d[i]);
            fputs(cell, fp);
        }
        fputc('\n', fp);
        for (int i = 0; i < nseg; ++i) {
            format_real(cell, sizeof cell, h[i]);
            fputs(cell, fp);
        }
        fputc('\n', fp);
    }
    if (fclose(fp) != 0)
        return fail(RTMI_ERR_IO, "error closing %s", path);
    return RTMI_OK;
}

}  // extern "C"
