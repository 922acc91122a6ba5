// rtmi_capi.cpp -- the C ABI of librtmi.so (include/rtmi.h).
//
// Host side of the drop-in boundary: owns the per-process device context, moves
// the caller's host buffers to HBM and back for the reference-shaped entry points
// (dff_, rtmi_tracerays, rtmi_dff7), and validates + forwards the device-resident
// entry points to the launchers in rtmi_kernels.hip.  There is NO CPU fallback:
// if the GPU path cannot run, the call fails loudly.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "rtmi_host.h"

namespace rtmi_host {

static thread_local std::string g_err;

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

const StatusWord &status_word()
{
    // Allocated once per process, by the first caller: portable (every device of the process sees it) and mapped.
    static StatusWord w = [] {
        StatusWord v;
        int *h = nullptr;
        if (hipHostMalloc((void **)&h, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && h) {
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

// The word is a pinned allocation, which must not happen inside a caller's stream capture: a launch that arrives
// capturing before the word exists is refused (rtmi_init, or any earlier un-captured call, creates it).
static std::mutex g_status_mu;
static bool g_status_made = false;
int status_word_for(hipStream_t stream, int **dev_out)
{
    {
        std::lock_guard<std::mutex> lk(g_status_mu);
        if (!g_status_made) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
                return fail(RTMI_ERR_BAD_ARG, "first librtmi launch of the process inside a stream capture: call "
                                               "rtmi_init (or one un-captured launch) before capturing");
            (void)status_word();
            g_status_made = true;
        }
    }
    *dev_out = status_word().dev;
    return RTMI_OK;
}

int device_fault(int st)
{
    if (st & 8)
        return fail(RTMI_ERR_BAD_ARG, "a model row of a batch launch had nlayers outside [1, ld_model - 1] (device status "
                                       "%d): its outputs were filled with NaN", st);
    return fail(RTMI_ERR_HIP, "the fused kernel reported an internal scheduling fault (status %d): results are not valid", st);
}

// Tuning knobs of the fused kernel, overridable from the environment for A/B runs.
static std::mutex g_tune_mu;
static Tuning &tuning_locked()
{
    static Tuning t = [] {
        Tuning v{0, 4, 32, 128, 12, 3, nullptr, 1};
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

// A sampler handle whose source set is below the batch path's own threshold still wants the fused kernel's sampler form (exact
// sums -> the speculative persistent sweep): its entry points set this for the calling thread (rtmi_pt.hip::PathScope), and
// every routing decision below sees force_path = 2 unless the caller forced a path itself.
static thread_local int g_path_override = 0;
void set_path_override(int v) { g_path_override = v; }
int path_override() { return g_path_override; }

Tuning tuning_snapshot()
{
    std::lock_guard<std::mutex> lk(g_tune_mu);
    Tuning t = tuning_locked();
    if (g_path_override && t.force_path == 0)
        t.force_path = g_path_override;
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
    int *unused = nullptr;
    int rc = status_word_for(nullptr, &unused);  // one-time pinned allocation: here, not inside a caller's capture
    if (rc)
        return rc;
    c.ready = true;
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

static bool batch_takes_fused(const Tuning &tn, int64_t nsrc)
{
    return tn.force_path != 1 && (nsrc >= RTMI_FUSED_MIN_NSRC / 4 || tn.force_path == 2);
}

bool batch_is_fused(int64_t nsrc)
{
    return batch_takes_fused(tuning_snapshot(), nsrc);
}

int batch_grid_x(int nmodels, int max_nlayers, int64_t nsrc, int cus)
{
    const Tuning tn = tuning_snapshot();
    if (batch_takes_fused(tn, nsrc)) {
        int threads = 0, slots = 0;
        size_t lds = 0;
        return rtmi_fused_geometry(nsrc, max_nlayers, nmodels, cus, tn.solve_waves, (int)sizeof(double), &threads, &slots,
                                   &lds);
    }
    return rtmi_launch_grid_x(nsrc, nmodels, cus);
}

int launch_batch(const double *d_vels, const double *d_depths, const int32_t *d_nlayers, int ld_model, int nmodels,
                 int max_nlayers, const double *d_src_offset, const double *d_src_depth, int64_t nsrc, double *d_timeP,
                 const double *d_obs, double *d_partial, bool allow_skip, int cus, int *gx_out, hipStream_t stream,
                 const rtmi::PtInc *d_inc, double *d_r2, unsigned long long *d_inc_counts,
                 unsigned long long *d_partial_x, int part_stride, int acc_atomic, const rtmi::PtPend *d_pinfo)
{
    const Tuning tn = tuning_snapshot();
    int *d_status = nullptr;
    int rc = status_word_for(stream, &d_status);
    if (rc)
        return rc;
    int gx = 0;
    if (batch_takes_fused(tn, nsrc)) {
        rtmi::FusedArgs fa;
        memset(&fa, 0, sizeof fa);
        fa.d_vels = d_vels;
        fa.d_depths = d_depths;
        fa.d_nlayers = d_nlayers;
        fa.ld_model = ld_model;
        fa.refill_min = tn.refill_min;
        fa.tail_group = tn.tail_group;
        fa.retire_max = tn.retire_max;
        fa.d_status = d_status;
        fa.promote_age = tn.promote_age;
        fa.allow_skip = allow_skip ? 1 : 0;
        fa.d_src_offset = d_src_offset;
        fa.d_src_depth = d_src_depth;
        fa.nsrc = nsrc;
        fa.d_timeP = d_timeP;
        fa.d_obs = d_obs;
        fa.d_partial = d_partial;
        if (d_partial_x && d_obs && !d_timeP) {  // the sampler's form of the sums (exact), optionally incremental
            fa.d_partial_x = d_partial_x;
            fa.part_stride = part_stride;
            fa.acc_atomic = acc_atomic;
            fa.d_pinfo = d_pinfo;
            fa.count_words = nmodels * (part_stride > 0 ? part_stride : 1);
            if (d_inc && d_r2) {
                fa.d_inc = d_inc;
                fa.d_r2 = d_r2;
                fa.d_inc_counts = d_inc_counts;
            }
        }
        HIP_TRY(rtmi_launch_fused(&fa, nmodels, max_nlayers, cus, tn.solve_waves, &gx, stream));
    } else {
        rtmi::TraceArgs a;
        memset(&a, 0, sizeof a);
        a.d_vels = d_vels;
        a.d_depths = d_depths;
        a.d_nlayers = d_nlayers;
        a.ld_model = ld_model;
        a.d_src_offset = d_src_offset;
        a.d_src_depth = d_src_depth;
        a.nsrc = nsrc;
        a.d_timeP = d_timeP;
        a.d_obs = d_obs;
        a.d_partial = d_partial;
        a.allow_skip = allow_skip ? 1 : 0;
        a.d_status = d_status;
        gx = rtmi_launch_grid_x(nsrc, nmodels, cus);
        HIP_TRY(rtmi_launch_trace(&a, nmodels, max_nlayers, gx, stream));
    }
    if (gx_out)
        *gx_out = gx;
    return RTMI_OK;
}

// The tempering sampler's sweep as one persistent launch (rtmi_kernels.hip::pt_sweep_kernel): same arguments as the
// batched likelihood launch of launch_batch plus the sampler's device view.  RTMI_ERR_UNSUPPORTED-like outcome (any
// non-zero return WITHOUT a sticky error message) tells the caller to use the per-slot launches instead.
int launch_pt_sweep(const double *d_vels, const double *d_depths, const int32_t *d_nlayers, int ld_model, int nchains,
                    int max_nlayers, const double *d_src_offset, const double *d_src_depth, int64_t nsrc,
                    const double *d_obs, double *d_partial, int cus, const void *ptdev, size_t ptdev_bytes,
                    const void *sweep_args, hipStream_t stream, const rtmi::PtInc *d_inc, double *d_r2,
                    unsigned long long *d_inc_counts, unsigned long long *d_partial_x, int part_stride)
{
    const rtmi::PtSweepArgs *w = static_cast<const rtmi::PtSweepArgs *>(sweep_args);
    const Tuning tn = tuning_snapshot();
    if (!batch_takes_fused(tn, nsrc)) {
        if (getenv("RTMI_PT_VERBOSE"))
            fprintf(stderr, "rtmi: persistent sweep needs the fused kernel path\n");
        return -1;
    }
    int *d_status = nullptr;
    int rc = status_word_for(stream, &d_status);
    if (rc)
        return rc;
    rtmi::FusedArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.d_vels = d_vels;
    fa.d_depths = d_depths;
    fa.d_nlayers = d_nlayers;
    fa.ld_model = ld_model;
    fa.refill_min = tn.refill_min;
    fa.tail_group = tn.tail_group;
    fa.retire_max = tn.retire_max;
    fa.d_status = d_status;
    fa.promote_age = tn.promote_age;
    fa.allow_skip = 1;
    fa.d_src_offset = d_src_offset;
    fa.d_src_depth = d_src_depth;
    fa.nsrc = nsrc;
    fa.d_obs = d_obs;
    fa.d_partial = d_partial;
    fa.d_partial_x = d_partial_x;
    fa.part_stride = part_stride;
    if (d_partial_x && d_inc && d_r2) {
        fa.d_inc = d_inc;
        fa.d_r2 = d_r2;
        fa.d_inc_counts = d_inc_counts;
    }
    const hipError_t e = rtmi_launch_pt_sweep(&fa, ptdev, ptdev_bytes, w, nchains, max_nlayers, cus, tn.solve_waves, stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // not sticky: the caller falls back
        if (getenv("RTMI_PT_VERBOSE"))
            fprintf(stderr, "rtmi: persistent sweep launch refused (%s): using the per-slot launches\n", hipGetErrorString(e));
        return -1;
    }
    return RTMI_OK;
}

int launch_pt_prop(const double *d_vels, const double *d_depths, const int32_t *d_nlayers, int ld_model, int nchains, int max_nlayers,
                   const double *d_src_offset, const double *d_src_depth, int64_t nsrc, const double *d_obs, int cus, const void *ptdev,
                   size_t ptdev_bytes, const void *d_sweep_args, int nspec, unsigned int epoch, int parity, int nslots_k, hipStream_t stream,
                   const rtmi::PtInc *d_inc, double *d_r2, unsigned long long *d_inc_counts, unsigned long long *d_partial_x,
                   int part_stride, const rtmi::PtPend *d_pinfo, int count_words)
{
    const Tuning tn = tuning_snapshot();
    if (!batch_takes_fused(tn, nsrc) || !d_partial_x)
        return -1;
    int *d_status = nullptr;
    int rc = status_word_for(stream, &d_status);
    if (rc)
        return rc;
    rtmi::FusedArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.d_vels = d_vels;
    fa.d_depths = d_depths;
    fa.d_nlayers = d_nlayers;
    fa.ld_model = ld_model;
    fa.refill_min = tn.refill_min;
    fa.tail_group = tn.tail_group;
    fa.retire_max = tn.retire_max;
    fa.d_status = d_status;
    fa.promote_age = tn.promote_age;
    fa.allow_skip = 1;
    fa.d_src_offset = d_src_offset;
    fa.d_src_depth = d_src_depth;
    fa.nsrc = nsrc;
    fa.d_obs = d_obs;
    fa.d_partial_x = d_partial_x;
    fa.part_stride = part_stride;
    fa.acc_atomic = 1;
    fa.d_pinfo = d_pinfo;
    fa.count_words = count_words > 0 ? count_words : 1;
    if (d_inc && d_r2) {
        fa.d_inc = d_inc;
        fa.d_r2 = d_r2;
        fa.d_inc_counts = d_inc_counts;
    }
    const hipError_t e = rtmi_launch_pt_prop(&fa, ptdev, ptdev_bytes, static_cast<const rtmi::PtSweepArgs *>(d_sweep_args), nspec, epoch,
                                             parity, nslots_k, nchains, max_nlayers, cus, tn.solve_waves, stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // not sticky: the caller falls back
        if (getenv("RTMI_PT_VERBOSE"))
            fprintf(stderr, "rtmi: speculative sweep launch refused (%s): using the per-slot launches\n", hipGetErrorString(e));
        return -1;
    }
    return RTMI_OK;
}

}  // namespace rtmi_host

using namespace rtmi_host;

namespace {

// dff_ / TraceRays calls up to this size run on a mapped pinned buffer (no copy commands).  Measured per call
// (profiles/r02m/small_threshold.txt): 1e4 sources 61 us against 122 us through the copy pipeline, 1e5 sources 145 against
// 179 us; at 2e5 the pipeline wins (218 against 268 us).  RTMI_SMALL_CALL_NSRC overrides.
#define RTMI_SMALL_CALL_NSRC 131072

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

// ---------------------------------------------------------------------------------------------------------------
// The drop-in path for large calls: a chunked three-queue pipeline per device.
//
// The caller's buffers are ordinary pageable host memory and every byte crosses PCIe once (16 B in, 8 B out per
// source), so the call is transfer bound: the source arrays are cut into chunks and chunk i+1's upload, chunk i's
// kernel and chunk i-1's download run on three streams of the device (uploads and downloads use opposite directions
// of the link).  With RTMI_DEVICES the sources are additionally sharded in contiguous slices over several GPUs
// ("single forward": no collective, the model is replicated, the host gathers timeP): each GPU has
// its own link to the host, and each slice is driven by its own host thread.
// ---------------------------------------------------------------------------------------------------------------
struct DropinLane {
    int device = 0;
    int cus = 0;
    hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
    double *d_model = nullptr;
    size_t model_cap = 0;
    double *d_src = nullptr;  // offset | depth | timeP | p_final of this lane's slice
    size_t src_cap = 0;
    std::vector<hipEvent_t> ev_in, ev_k;
};

struct DropinState {
    std::vector<std::unique_ptr<DropinLane>> lanes;  // lanes[0] = the bound device
    bool parsed = false;
    std::vector<int> extra_devices;                  // RTMI_DEVICES beyond the bound device (may repeat a device)
    int64_t chunk = 0;                               // sources per chunk
    int64_t shard_min = 0;                           // slices smaller than this are not worth another device
    int reg = 0;                                     // RTMI_DROPIN_REGISTER=1: pin the caller's buffers for the call
};
DropinState g_dropin;

int lane_ready(DropinLane &L, int device, size_t nevents)
{
    HIP_TRY(hipSetDevice(device));
    if (!L.s_in) {
        L.device = device;
        HIP_TRY(hipDeviceGetAttribute(&L.cus, hipDeviceAttributeMultiprocessorCount, device));
        HIP_TRY(hipStreamCreateWithFlags(&L.s_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&L.s_k, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&L.s_out, hipStreamNonBlocking));
        HIP_TRY(rtmi_prepare_kernels());
    }
    while (L.ev_in.size() < nevents) {
        hipEvent_t a = nullptr, b = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        L.ev_in.push_back(a);
        L.ev_k.push_back(b);
    }
    return RTMI_OK;
}

// Sources per chunk of the pipeline for a slice of n sources: an eighth of the slice, between 262 144 (below that the
// fused kernel no longer fills the chip and the per-copy cost of ~10 us shows: 1e6 sources 0.55 ms with 4 chunks, 0.65 ms
// with 8) and 1 048 576 (1e7 sources: 3.4 ms with chunks of 1 M, 3.7 ms with 4 M -- the first upload and the last
// download are not overlapped -- and 5.5 ms with 262 144; profiles/r02m/dropin_modes_1e7.txt).
int64_t dropin_chunk_for(int64_t n)
{
    if (g_dropin.chunk > 0)
        return g_dropin.chunk;
    int64_t c = (n / 8 + 65535) / 65536 * 65536;
    if (c < 262144)
        c = 262144;
    if (c > 1048576)
        c = 1048576;
    return c;
}

void parse_dropin_env_locked(Context &c)
{
    if (g_dropin.parsed)
        return;
    g_dropin.parsed = true;
    g_dropin.chunk = 0;  // 0: chosen per call from the slice length (dropin_chunk_for)
    if (const char *e = getenv("RTMI_DROPIN_CHUNK"))
        if (atoll(e) >= 4096)
            g_dropin.chunk = atoll(e) / 64 * 64;
    g_dropin.shard_min = 131072;
    if (const char *e = getenv("RTMI_SHARD_MIN"))
        if (atoll(e) >= 1)
            g_dropin.shard_min = atoll(e);
    // The caller's arrays are pinned for the duration of the call (hipHostRegister / hipHostUnregister around the
    // pipeline): the chunk copies are then true asynchronous DMA and overlap; measured on MI355X, 1e6 sources:
    // 0.56 ms against 0.66 ms for one un-chunked pass over pageable memory and 0.98 ms for chunks of pageable memory
    // (profiles/r02a/dropin_modes.txt).  A registration never outlives the call (no stale pins on freed memory); if a
    // range cannot be registered (already pinned by the caller, exotic mapping) its copies simply stay pageable.
    g_dropin.reg = 1;
    if (const char *e = getenv("RTMI_DROPIN_REGISTER"))
        g_dropin.reg = atoi(e);
    g_dropin.lanes.emplace_back(new DropinLane);
    const char *e = getenv("RTMI_DEVICES");
    if (!e || !*e)
        return;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess)
        return;
    if (!strcmp(e, "all")) {
        for (int d = 0; d < ndev; ++d)
            if (d != c.device)
                g_dropin.extra_devices.push_back(d);
    } else {
        // comma-separated list; the bound device takes the first slice, every further entry one more slice (an
        // entry may name the bound device again: two pipelines on one GPU, used by the tests on a 1-GPU box)
        bool first = true;
        for (const char *q = e; *q;) {
            const int d = atoi(q);
            if (d >= 0 && d < ndev && !first)
                g_dropin.extra_devices.push_back(d);
            first = false;
            while (*q && *q != ',')
                ++q;
            if (*q == ',')
                ++q;
        }
    }
    for (size_t i = 0; i < g_dropin.extra_devices.size(); ++i)
        g_dropin.lanes.emplace_back(new DropinLane);
}

// One slice [k0, k0 + n) of the call on one device.
// timeP_dev: the caller's (pinned) timeP as the device sees it, or NULL.  With it the kernels write the travel times STRAIGHT
// into the caller's array over PCIe (posted writes, behind the chunk's arithmetic) and no download is queued at all: round 4's
// timeline of a 1e6-source call (profiles/r04ao/dropin_timeline.txt) showed the four downloads as blit kernels of 37 us each
// that no upload overlapped.
int run_slice(DropinLane &L, int device, const double *vels, const double *depths, int nlayers, const double *src_offset,
              const double *src_depth, int64_t n, double *timeP, double *p_out, int64_t chunk, const Tuning &tn, double *timeP_dev = nullptr)
{
    const int64_t nch = (n + chunk - 1) / chunk;
    int rc = lane_ready(L, device, (size_t)nch);
    if (rc)
        return rc;
    rc = grow(&L.d_model, &L.model_cap, (size_t)2 * nlayers + 1);
    if (rc)
        return rc;
    rc = grow(&L.d_src, &L.src_cap, (size_t)4 * n);
    if (rc)
        return rc;
    int *d_status = nullptr;
    rc = status_word_for(nullptr, &d_status);
    if (rc)
        return rc;
    double *d_vels = L.d_model, *d_depths = L.d_model + nlayers + 1;
    double *d_off = L.d_src, *d_dep = L.d_src + n, *d_t = timeP_dev ? timeP_dev : L.d_src + 2 * n, *d_p = L.d_src + 3 * n;
    HIP_TRY(hipMemcpyAsync(d_vels, vels, (size_t)(nlayers + 1) * sizeof(double), hipMemcpyHostToDevice, L.s_in));
    HIP_TRY(hipMemcpyAsync(d_depths, depths, (size_t)nlayers * sizeof(double), hipMemcpyHostToDevice, L.s_in));
    for (int64_t c = 0; c < nch; ++c) {
        const int64_t k0 = c * chunk, m = std::min(chunk, n - k0);
        HIP_TRY(hipMemcpyAsync(d_off + k0, src_offset + k0, (size_t)m * sizeof(double), hipMemcpyHostToDevice, L.s_in));
        HIP_TRY(hipMemcpyAsync(d_dep + k0, src_depth + k0, (size_t)m * sizeof(double), hipMemcpyHostToDevice, L.s_in));
        HIP_TRY(hipEventRecord(L.ev_in[c], L.s_in));
        HIP_TRY(hipStreamWaitEvent(L.s_k, L.ev_in[c], 0));
        const bool fused = tn.force_path != 1 && (m >= RTMI_FUSED_MIN_NSRC || tn.force_path == 2);
        if (fused) {
            rtmi::FusedArgs fa;
            memset(&fa, 0, sizeof fa);
            fa.d_vels = d_vels;
            fa.d_depths = d_depths;
            fa.nlayers = nlayers;
            fa.refill_min = tn.refill_min;
            fa.tail_group = tn.tail_group;
            fa.retire_max = tn.retire_max;
            fa.d_status = d_status;
            fa.promote_age = tn.promote_age;
            fa.d_src_offset = d_off + k0;
            fa.d_src_depth = d_dep + k0;
            fa.nsrc = m;
            fa.d_timeP = d_t + k0;
            fa.d_p_final = p_out ? d_p + k0 : nullptr;
            HIP_TRY(rtmi_launch_fused(&fa, 1, nlayers, L.cus, tn.solve_waves, nullptr, L.s_k));
        } else {
            rtmi::TraceArgs a;
            memset(&a, 0, sizeof a);
            a.d_vels = d_vels;
            a.d_depths = d_depths;
            a.nlayers = nlayers;
            a.d_src_offset = d_off + k0;
            a.d_src_depth = d_dep + k0;
            a.nsrc = m;
            a.d_timeP = d_t + k0;
            a.d_p_final = p_out ? d_p + k0 : nullptr;
            HIP_TRY(rtmi_launch_trace(&a, 1, nlayers, rtmi_launch_grid_x(m, 1, L.cus), L.s_k));
        }
        if (timeP_dev && !p_out)
            continue;  // (nothing to download: the kernel wrote the caller's array)
        HIP_TRY(hipEventRecord(L.ev_k[c], L.s_k));
        HIP_TRY(hipStreamWaitEvent(L.s_out, L.ev_k[c], 0));
        if (!timeP_dev)
            HIP_TRY(hipMemcpyAsync(timeP + k0, d_t + k0, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, L.s_out));
        if (p_out)
            HIP_TRY(hipMemcpyAsync(p_out + k0, d_p + k0, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, L.s_out));
    }
    if (timeP_dev)
        HIP_TRY(hipStreamSynchronize(L.s_k));
    if (!timeP_dev || p_out)
        HIP_TRY(hipStreamSynchronize(L.s_out));
    return RTMI_OK;
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
    static const int small_call_nsrc = getenv("RTMI_SMALL_CALL_NSRC") ? atoi(getenv("RTMI_SMALL_CALL_NSRC")) : RTMI_SMALL_CALL_NSRC;
    if (nsrc <= small_call_nsrc) {
        // Small and medium calls (the reference's own example has NSrc = 20, its sampler calls TraceRays with 1e4): the
        // kernel reads the model and the sources from, and writes the times to, ONE pinned host buffer mapped into the
        // GPU's address space -- no copy commands, the call costs two host memcpy, one launch and one stream synchronise.
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
        // Completion: a launch plus an interrupt-driven hipStreamSynchronize costs more than the reference's whole call at its
        // own example size (41 us against 28 us for NSrc = 20, round 3).  The stream writes a sequence number into a mapped
        // pinned word right behind the kernel (a command-processor write, ordered after the kernel's completion and its
        // release to system scope) and this thread spins on that word -- bounded: a call that takes longer than a few
        // hundred microseconds sleeps in hipStreamSynchronize like before.  RTMI_DROPIN_SPIN=0 switches it off.
        bool seen = false;
        if (c.done_ok != 0) {
            static const bool spin_on = !(getenv("RTMI_DROPIN_SPIN") && !strcmp(getenv("RTMI_DROPIN_SPIN"), "0"));
            if (c.done_ok < 0) {
                unsigned int *w = nullptr;
                c.done_ok = (spin_on && hipHostMalloc((void **)&w, 64, hipHostMallocMapped) == hipSuccess) ? 1 : 0;
                if (c.done_ok) {
                    *w = 0u;
                    c.h_done = w;
                } else {
                    (void)hipGetLastError();
                }
            }
            if (c.done_ok == 1) {
                const unsigned int seq = ++c.done_seq;
                if (hipStreamWriteValue32(c.stream, (void *)c.h_done, seq, 0) == hipSuccess) {
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int spins = 0;; ++spins) {
                        if (*c.h_done == seq) {
                            seen = true;
                            break;
                        }
                        if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400))
                            break;
                        __builtin_ia32_pause();
                    }
                    std::atomic_thread_fence(std::memory_order_acquire);
                } else {
                    (void)hipGetLastError();
                    c.done_ok = 0;  // not available on this stack: the synchronise below, from now on
                }
            }
        }
        if (!seen)
            HIP_TRY(hipStreamSynchronize(c.stream));
        if (int st = rtmi_device_status())
            return device_fault(st);
        memcpy(timeP, h + nm + 2 * n, n * sizeof(double));
        if (want_p)
            return rtmi_write_rays_dat("rays.dat", vels, depths, nlayers, src_offset, src_depth, h + nm + 3 * n, nsrc);
        return RTMI_OK;
    }
    parse_dropin_env_locked(c);
    const Tuning tn = tuning_snapshot();
    std::vector<double> p;
    if (want_p)
        p.resize(n);
    // contiguous slices over the lanes (the bound device first); a slice below shard_min is not worth a device
    int nl_use = (int)g_dropin.lanes.size();
    while (nl_use > 1 && (int64_t)n / nl_use < g_dropin.shard_min)
        --nl_use;
    bool registered[3] = {false, false, false};
    if (g_dropin.reg) {
        registered[0] = hipHostRegister((void *)src_offset, n * sizeof(double), hipHostRegisterDefault) == hipSuccess;
        registered[1] = hipHostRegister((void *)src_depth, n * sizeof(double), hipHostRegisterDefault) == hipSuccess;
        registered[2] = hipHostRegister((void *)timeP, n * sizeof(double), hipHostRegisterDefault) == hipSuccess;
        (void)hipGetLastError();
    }
    // RTMI_DROPIN_DIRECT_OUT=1: the bound device's slice writes its travel times straight into the caller's pinned array instead
    // of queueing downloads.  Measured at 1e6 sources (profiles/r04ao/direct_out.txt): 0.568 against 0.551 ms on arrays that are
    // reused (the kernels' 8-byte writes cross PCIe less efficiently than the copy engine's bursts), 1.04 against 1.18 ms on newly
    // allocated arrays (R's .Fortran copies: no second party faults the result's pages in).  Off by default.
    double *timeP_dev = nullptr;
    {
        static const bool direct_out = getenv("RTMI_DROPIN_DIRECT_OUT") && !strcmp(getenv("RTMI_DROPIN_DIRECT_OUT"), "1");
        if (direct_out && registered[2] && hipHostGetDevicePointer((void **)&timeP_dev, (void *)timeP, 0) != hipSuccess) {
            (void)hipGetLastError();
            timeP_dev = nullptr;
        }
        if (!direct_out)
            timeP_dev = nullptr;
    }
    std::vector<std::future<int>> jobs;
    std::vector<std::string> errs((size_t)nl_use);
    int rc_all = RTMI_OK;
    for (int i = nl_use - 1; i >= 0; --i) {
        const int64_t k0 = (int64_t)n * i / nl_use, k1 = (int64_t)n * (i + 1) / nl_use;
        const int dev = i == 0 ? c.device : g_dropin.extra_devices[(size_t)i - 1];
        DropinLane *L = g_dropin.lanes[(size_t)i].get();
        auto job = [=, &tn, &errs, &p]() {
            const int r = run_slice(*L, dev, vels, depths, nlayers, src_offset + k0, src_depth + k0, k1 - k0, timeP + k0,
                                    want_p ? p.data() + k0 : nullptr, dropin_chunk_for(k1 - k0), tn,
                                    (i == 0 && timeP_dev) ? timeP_dev + k0 : nullptr);
            if (r)
                errs[(size_t)i] = g_err;  // the worker thread's message
            return r;
        };
        if (i == 0) {
            rc_all = job();  // the bound device's slice on the calling thread
        } else {
            jobs.push_back(std::async(std::launch::async, job));
        }
    }
    for (auto &j : jobs) {
        const int r = j.get();
        if (r && !rc_all)
            rc_all = r;
    }
    if (registered[0])
        (void)hipHostUnregister((void *)src_offset);
    if (registered[1])
        (void)hipHostUnregister((void *)src_depth);
    if (registered[2])
        (void)hipHostUnregister((void *)timeP);
    HIP_TRY(hipSetDevice(c.device));
    if (rc_all) {
        for (const auto &e : errs)
            if (!e.empty()) {
                g_err = e;
                break;
            }
        return rc_all;
    }
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
    // an R session (or any host program) that loads the library survives a device failure: the reason is on stderr and
    // every output is NaN -- never a plausible number.  RTMI_ABORT_ON_ERROR=1 asks for the hard stop instead.
    const char *e = getenv("RTMI_ABORT_ON_ERROR");
    if (e && strcmp(e, "0") != 0)
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
        // rebind: drop the old context's buffers and the drop-in pipelines
        (void)hipSetDevice(c.device);
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
        ++c.set_generation;
        for (auto &L : g_dropin.lanes) {
            if (!L->s_in)
                continue;
            (void)hipSetDevice(L->device);
            if (L->d_model)
                (void)hipFree(L->d_model);
            if (L->d_src)
                (void)hipFree(L->d_src);
            for (hipEvent_t e : L->ev_in)
                (void)hipEventDestroy(e);
            for (hipEvent_t e : L->ev_k)
                (void)hipEventDestroy(e);
            (void)hipStreamDestroy(L->s_in);
            (void)hipStreamDestroy(L->s_k);
            (void)hipStreamDestroy(L->s_out);
        }
        g_dropin = DropinState();
        (void)hipSetDevice(c.device);
        if (c.stream)
            (void)hipStreamDestroy(c.stream);
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
    std::lock_guard<std::mutex> lk(g_tune_mu);
    Tuning &t = tuning_locked();
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

int rtmi_clear_device_status(void)
{
    const StatusWord &w = status_word();
    if (!w.host)
        return 0;
    return __atomic_exchange_n(w.host, 0, __ATOMIC_SEQ_CST);
}

int rtmi_debug_stamps(unsigned long long *d_stamps)
{
    std::lock_guard<std::mutex> lk(g_tune_mu);
    tuning_locked().d_stamps = d_stamps;
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
    const Tuning tn = tuning_snapshot();
    int *d_status = nullptr;
    rc = status_word_for((hipStream_t)stream, &d_status);
    if (rc)
        return rc;
    // (launches the group-of-lanes kernel takes go to it unless a path is forced: rtmi_kernels.h::rtmi_small_kernel_takes)
    const bool fused = tn.force_path != 1 && (tn.force_path == 2 || (nsrc >= RTMI_FUSED_MIN_NSRC && (tn.d_stamps || !rtmi_small_kernel_takes(nsrc, nlayers))));
    if (fused) {
        rtmi::FusedArgs fa;
        memset(&fa, 0, sizeof fa);
        fa.d_vels = d_vels;
        fa.d_depths = d_depths;
        fa.nlayers = nlayers;
        fa.refill_min = tn.refill_min;
        fa.tail_group = tn.tail_group;
        fa.retire_max = tn.retire_max;
        fa.d_status = d_status;
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
    a.lane_per_source = tn.force_path == 1 ? 1 : 0;
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
    HIP_TRY(rtmi_launch_fastmath_check(d_w, d_num, d_den, n, d_mismatch, 0, (hipStream_t)stream));
    return RTMI_OK;
}

int rtmi_selftest_fastmath_ex(const double *d_w, const double *d_num, const double *d_den, int64_t n,
                              unsigned long long *d_mismatch5, void *stream)
{
    if (n < 0 || !d_w || !d_num || !d_den || !d_mismatch5)
        return fail(RTMI_ERR_BAD_ARG, "bad argument");
    if (n == 0)
        return RTMI_OK;
    HIP_TRY(rtmi_launch_fastmath_check(d_w, d_num, d_den, n, d_mismatch5, 1, (hipStream_t)stream));
    return RTMI_OK;
}

int rtmi_selftest_values(const double *d_w, const double *d_num, const double *d_den, int64_t n, double *d_div,
                         double *d_div_guarded, double *d_sqrt, double *d_tf, double *d_tp, void *stream)
{
    if (n < 0 || !d_w || !d_num || !d_den)
        return fail(RTMI_ERR_BAD_ARG, "bad argument");
    if (n == 0)
        return RTMI_OK;
    HIP_TRY(rtmi_launch_fastmath_values(d_w, d_num, d_den, n, d_div, d_div_guarded, d_sqrt, d_tf, d_tp, (hipStream_t)stream));
    return RTMI_OK;
}

int rtmi_selftest_values_f32(const float *d_w, const float *d_num, const float *d_den, int64_t n, float *d_div,
                             float *d_tf, float *d_tp, void *stream)
{
    if (n < 0 || !d_w || !d_num || !d_den)
        return fail(RTMI_ERR_BAD_ARG, "bad argument");
    if (n == 0)
        return RTMI_OK;
    HIP_TRY(rtmi_launch_fastmath_values_f32(d_w, d_num, d_den, n, d_div, d_tf, d_tp, (hipStream_t)stream));
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
    ++c.set_generation;
    return RTMI_OK;
}

uint64_t rtmi_sources_generation(void)
{
    Context &c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    return c.set_n > 0 ? c.set_generation : 0;
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
    int max_nl = 1;  // the launch is sized for the largest model of THIS batch, not for the row stride
    for (int m = 0; m < nmodels; ++m) {
        if (nlayers[m] < 1 || nlayers[m] > ld_model - 1)
            return fail(RTMI_ERR_BAD_ARG, "nlayers[%d] = %d outside [1, ld_model - 1]", m, nlayers[m]);
        max_nl = nlayers[m] > max_nl ? nlayers[m] : max_nl;
    }
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
    int gx = 0;
    rc = launch_batch(d_v, d_d, d_nl, ld_model, nmodels, max_nl, c.d_set, c.d_set + n, c.set_n, nullptr, c.d_set + 2 * n,
                      d_work, false, c.cus, &gx, c.stream);
    if (rc)
        return rc;
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

int rtmi_trace_device_f32_fast(const float *d_vels, const float *d_depths, int nlayers, const float *d_src_offset,
                               const float *d_src_depth, int64_t nsrc, float *d_timeP, float *d_p_final, void *stream)
{
    int rc = check_model(nlayers);
    if (rc)
        return rc;
    if (nlayers > RTMI_F32FAST_MAX_NL)
        return fail(RTMI_ERR_BAD_ARG, "rtmi_trace_device_f32_fast: at most %d interfaces (got %d); use rtmi_trace_device_f32",
                    RTMI_F32FAST_MAX_NL, nlayers);
    if (nsrc < 0)
        return fail(RTMI_ERR_BAD_ARG, "nsrc < 0");
    if (nsrc == 0)
        return RTMI_OK;
    if (!d_vels || !d_depths || !d_src_offset || !d_src_depth || !d_timeP)
        return fail(RTMI_ERR_BAD_ARG, "NULL device pointer");
    int *d_status = nullptr;
    rc = status_word_for((hipStream_t)stream, &d_status);
    if (rc)
        return rc;
    rtmi::F32FastArgs a;
    memset(&a, 0, sizeof a);
    a.d_vels = d_vels;
    a.d_depths = d_depths;
    a.nlayers = nlayers;
    a.d_src_offset = d_src_offset;
    a.d_src_depth = d_src_depth;
    a.nsrc = nsrc;
    a.d_timeP = d_timeP;
    a.d_p_final = d_p_final;
    a.d_status = d_status;
    HIP_TRY(rtmi_launch_f32fast(&a, num_cus_for_current_device(), (hipStream_t)stream));
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
    const Tuning tn = tuning_snapshot();
    int *d_status = nullptr;
    rc = status_word_for((hipStream_t)stream, &d_status);
    if (rc)
        return rc;
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
        fa.d_status = d_status;
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
    // d_nlayers lives on the device: the launch is sized for the row stride, and the kernels check every
    // d_nlayers[m] against it (fault bit 8 + NaN outputs for a row outside [1, ld_model - 1])
    int gx = 0;
    const int rc = launch_batch(d_vels, d_depths, d_nlayers, ld_model, nmodels, ld_model - 1, d_src_offset, d_src_depth,
                                nsrc, d_timeP, want_ll ? d_obs : nullptr, (double *)d_work, false,
                                num_cus_for_current_device(), &gx, (hipStream_t)stream);
    if (rc)
        return rc;
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
