// profiles/ubench/f32fast_phases.hip -- where a launch of the fp32 fast kernel spends its time at config 5's shape: the kernel
// source compiled with F32F_CLOCKS (wave 0 of every workgroup adds the cycles of each phase to a device array), driven with
// sources like profiles/f32fast_bench.py's.  Prints the launch time by events and the phase shares.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DF32F_CLOCKS -o f32fast_phases profiles/ubench/f32fast_phases.hip
// Without -DF32F_CLOCKS: the plain kernel (for rocprofv3 --pmc runs of this binary); -DF32F_NO_SOLVE: without its layer loops.
#include "../../raytracerfortran_amd/csrc/rtmi_f32fast.hip"
#include <vector>
#include <algorithm>
#include <random>

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1250000;
    const int NL = 20;
    std::mt19937_64 g(2025);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    std::vector<float> d(NL), v(NL + 1), off(n), dep(n);
    double z = 0.0;
    for (int i = 0; i < NL; ++i)
        d[i] = (float)(z += 200.0 + 400.0 * u(g));
    for (auto &x : v)
        x = (float)(1500.0 + 5000.0 * u(g));
    std::sort(v.begin(), v.end());
    for (int64_t i = 0; i < n; ++i) {
        off[i] = (float)(10.0 + 7990.0 * u(g));
        dep[i] = (float)(50.0 + (z + 450.0) * u(g));
    }
    float *dd, *dv, *doff, *ddep, *dt;
    hipMalloc(&dd, NL * 4); hipMalloc(&dv, (NL + 1) * 4); hipMalloc(&doff, n * 4); hipMalloc(&ddep, n * 4); hipMalloc(&dt, n * 4);
    hipMemcpy(dd, d.data(), NL * 4, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), (NL + 1) * 4, hipMemcpyHostToDevice);
    hipMemcpy(doff, off.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(ddep, dep.data(), n * 4, hipMemcpyHostToDevice);
    rtmi::F32FastArgs a{};
    a.d_vels = dv; a.d_depths = dd; a.nlayers = NL; a.d_src_offset = doff; a.d_src_depth = ddep; a.nsrc = n; a.d_timeP = dt;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    for (int i = 0; i < 5; ++i)
        rtmi_launch_f32fast(&a, cus, 0);
    hipDeviceSynchronize();
#ifdef F32F_CLOCKS
    static unsigned long long raw[8 * 1024];
    hipMemcpyToSymbol(HIP_SYMBOL(rtmi::f32f_clk), raw, sizeof raw);
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
        rtmi_launch_f32fast(&a, cus, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[8] = {0};
#ifdef F32F_CLOCKS
    hipMemcpyFromSymbol(raw, HIP_SYMBOL(rtmi::f32f_clk), sizeof raw);
    for (int b = 0; b < 1024; ++b)
        for (int i = 0; i < 7; ++i)
            clk[i] += raw[b * 8 + i];
#endif
    const int64_t ntiles = (n + F32F_TILE - 1) / F32F_TILE;
    const int64_t grid = std::min<int64_t>((int64_t)cus * 4, ntiles);
    const char *names[7] = {"model tables: pairs + sums", "load sources", "locate + count + sort", "solve (wave 0)", "wait for the other waves", "store", "model tables: load, check, per layer"};
    double tot = 0;
    for (int i = 0; i < 7; ++i) tot += (double)clk[i];
    printf("%lld sources x %d interfaces: %.4f ms per launch (%.3e travel times/s), %lld workgroups\n", (long long)n, NL, ms / reps,
           n / (ms / reps) * 1e3, (long long)grid);
#ifdef F32F_CLOCKS
    printf("clock64 ticks per workgroup and launch:\n");
    for (int i = 0; i < 7; ++i)
        printf("  %-38s %9.0f  %5.1f %%\n", names[i], (double)clk[i] / reps / grid, 100.0 * clk[i] / tot);
    printf("  %-38s %9.0f  (= the launch's %.1f us if a workgroup lives as long as the launch: %.2f GHz)\n", "sum", tot / reps / grid, ms / reps * 1e3, tot / reps / grid / (ms / reps * 1e6));
#else
    (void)names; (void)tot;
#endif
    return 0;
}
