// profiles/ubench/launch_latency.hip -- what ONE small synchronous call costs on this stack, piece by piece: the floor under the
// drop-in dff_ at the reference's own example size (NSrc = 20).  Each variant: host writes its input into mapped pinned
// memory, launches one workgroup, learns of the completion, reads the result.  Medians of 2000 calls after 200 warm-ups.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_latency profiles/ubench/launch_latency.hip && /tmp/launch_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

// empty: nothing but the launch
__global__ void k_empty() {}
// flag: the kernel itself publishes its completion to the host (system-scope release store)
__global__ void k_flag(unsigned *flag, unsigned seq)
{
    if (threadIdx.x == 0)
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// work: a dependent chain of `hops` reads of mapped host memory (model -> sources -> ...), one write of a result, then the flag
__global__ void k_work(const double *in, double *out, int hops, unsigned *flag, unsigned seq)
{
    double acc = 0.0;
    int idx = threadIdx.x;
    for (int h = 0; h < hops; ++h) {
        const double v = in[idx];
        acc += v;
        idx = (idx + (int)v) & 63;  // the next address depends on the value
    }
    out[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && flag)
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double median(std::vector<double> &v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main()
{
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *flag = nullptr;
    double *buf = nullptr;
    CK(hipHostMalloc((void **)&flag, 64, hipHostMallocMapped));
    CK(hipHostMalloc((void **)&buf, 4096, hipHostMallocMapped));
    for (int i = 0; i < 512; ++i)
        buf[i] = 1.0;
    *flag = 0;
    unsigned seq = 0;
    const int warm = 200, reps = 2000;
    auto spin = [&](unsigned want) {
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want)
            __builtin_ia32_pause();
    };
    auto run = [&](const char *name, auto &&body) {
        std::vector<double> ts;
        for (int i = 0; i < warm + reps; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            body();
            const auto t1 = std::chrono::steady_clock::now();
            if (i >= warm)
                ts.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        std::vector<double> s = ts;
        std::sort(s.begin(), s.end());
        printf("%-78s median %6.1f us   min %6.1f us   p90 %6.1f us\n", name, median(ts), s.front(), s[s.size() * 9 / 10]);
    };
    run("empty kernel + hipStreamSynchronize", [&] {
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
        (void)hipStreamSynchronize(st);
    });
    run("empty kernel + hipStreamWriteValue32 + spin on the word", [&] {
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
        ++seq;
        (void)hipStreamWriteValue32(st, flag, seq, 0);
        spin(seq);
    });
    run("kernel stores the word itself (system-scope release) + spin", [&] {
        ++seq;
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, flag, seq);
        spin(seq);
    });
    for (int hops : {1, 3, 6}) {
        char name[128];
        snprintf(name, sizeof name, "kernel: %d dependent reads of mapped host memory, result, word + spin", hops);
        run(name, [&] {
            ++seq;
            hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, buf, buf + 256, hops, flag, seq);
            spin(seq);
        });
    }
    run("same, 3 reads, completion by hipStreamSynchronize", [&] {
        hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, buf, buf + 256, 3, (unsigned *)nullptr, 0u);
        (void)hipStreamSynchronize(st);
    });
    // the launch call alone (host side), the device idle afterwards
    {
        std::vector<double> ts;
        for (int i = 0; i < warm + reps; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            ++seq;
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, flag, seq);
            const auto t1 = std::chrono::steady_clock::now();
            spin(seq);
            if (i >= warm)
                ts.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        printf("%-78s median %6.1f us\n", "of which: the hipLaunchKernelGGL call returns after", median(ts));
    }
    return 0;
}
