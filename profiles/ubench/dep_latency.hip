// profiles/ubench/dep_latency.hip -- what ONE wave64 alone on its SIMD pays for DEPENDENT fp64 operations on gfx950: the
// regime of the solver's tails, of the tempering slot and of small calls, where a source's next evaluation cannot start
// before the previous one has ended.
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o /tmp/dep_latency profiles/ubench/dep_latency.hip && /tmp/dep_latency
// k_chain<N>: N independent chains of dependent v_fma_f64, interleaved instruction by instruction, ITER x 8 steps per chain;
// cycles per instruction = ticks / (ITER * 8 * N).  With one chain that is the dependent-issue latency; it falls to the
// 4-cycle issue rate once there are enough chains in flight.  k_rsq / k_rcp: the same for the quarter-rate ops; k_lds: a
// dependent chain of ds_read_b64 (pointer chasing through LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4000

template <int N>
__global__ void k_chain(double *out, unsigned long long *ticks, double seed)
{
    double a[8];
    for (int j = 0; j < 8; ++j)
        a[j] = seed + threadIdx.x * 1e-3 + j;
    const double b = 1.000001, c = 1e-9;
    unsigned long long t0 = clock64();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int j = 0; j < N; ++j)
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
        }
    }
    unsigned long long t1 = clock64();
    double s = 0;
    for (int j = 0; j < 8; ++j)
        s += a[j];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0)
        ticks[0] = t1 - t0;
}

template <int N, int OP>
__global__ void k_trans(double *out, unsigned long long *ticks, double seed)
{
    double a[8];
    for (int j = 0; j < 8; ++j)
        a[j] = seed + threadIdx.x * 1e-3 + j;
    unsigned long long t0 = clock64();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (OP == 0)
                    asm volatile("v_rsq_f64 %0, %0" : "+v"(a[j]));
                else
                    asm volatile("v_rcp_f64 %0, %0" : "+v"(a[j]));
            }
        }
    }
    unsigned long long t1 = clock64();
    double s = 0;
    for (int j = 0; j < 8; ++j)
        s += a[j];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0)
        ticks[0] = t1 - t0;
}

__global__ void k_lds(double *out, unsigned long long *ticks)
{
    __shared__ unsigned int next[1024];
    for (int i = threadIdx.x; i < 1024; i += 64)
        next[i] = (i * 17 + 64) & 1023;
    __syncthreads();
    unsigned int p = threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < ITER * 8; ++i)
        p = next[p];
    unsigned long long t1 = clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0)
        ticks[0] = t1 - t0;
}

int main()
{
    double *out;
    unsigned long long *ticks, h;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&ticks, 8);
#define RUN(NAME, KERNEL, PER, ...)                                                       \
    do {                                                                                  \
        for (int r = 0; r < 2; ++r) {                                                     \
            hipLaunchKernelGGL(KERNEL, dim3(1), dim3(64), 0, 0, out, ticks, ##__VA_ARGS__); \
            hipDeviceSynchronize();                                                       \
        }                                                                                 \
        hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);                                   \
        printf("%-44s %7.2f cycles per instruction\n", NAME, (double)h / ((double)ITER * 8 * (PER))); \
    } while (0)
    RUN("v_fma_f64, 1 dependent chain", (k_chain<1>), 1, 1.5);
    RUN("v_fma_f64, 2 chains", (k_chain<2>), 2, 1.5);
    RUN("v_fma_f64, 3 chains", (k_chain<3>), 3, 1.5);
    RUN("v_fma_f64, 4 chains", (k_chain<4>), 4, 1.5);
    RUN("v_fma_f64, 6 chains", (k_chain<6>), 6, 1.5);
    RUN("v_fma_f64, 8 chains", (k_chain<8>), 8, 1.5);
    RUN("v_rsq_f64, 1 dependent chain", (k_trans<1, 0>), 1, 1.5);
    RUN("v_rsq_f64, 3 chains", (k_trans<3, 0>), 3, 1.5);
    RUN("v_rcp_f64, 1 dependent chain", (k_trans<1, 1>), 1, 1.5);
    RUN("ds_read_b32, dependent (pointer chase)", k_lds, 1);
    return 0;
}
