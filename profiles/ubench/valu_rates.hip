// profiles/ubench/valu_rates.hip -- cycles per wave64 instruction of the fp64 VALU ops the solver uses, on gfx950.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates profiles/ubench/valu_rates.hip && /tmp/valu_rates
// Each kernel runs ITER x 8 independent instances of one instruction per wave (8 independent register sets, so the
// chain is throughput- not latency-bound) on every SIMD of the chip with `waves` waves per SIMD; the clock is
// s_memtime (shader clock).  Reported: cycles per instruction per SIMD = ticks * 1 / (ITER * 8 * waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 20000

#define DEF_KERNEL(NAME, ASM_BODY)                                                                         \
    __global__ void NAME(double *out, unsigned long long *ticks, double seed)                              \
    {                                                                                                      \
        double a0 = seed + threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,          \
               a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                                       \
        double b = 1.000001, c = 1e-9;                                                                     \
        float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = (float)a4, f5 = (float)a5, \
              f6 = (float)a6, f7 = (float)a7;                                                              \
        __syncthreads();                                                                                   \
        unsigned long long t0 = clock64();                                                                 \
        for (int i = 0; i < ITER; ++i) {                                                                   \
            ASM_BODY                                                                                       \
        }                                                                                                  \
        unsigned long long t1 = clock64();                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] =                                                       \
            a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;                  \
        if (threadIdx.x == 0)                                                                              \
            ticks[blockIdx.x] = t1 - t0;                                                                   \
    }

#define OP1_F64(op)                                                                                         \
    asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op \
                    " %6, %6\n" op " %7, %7"                                                                  \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define OP1_F32(op)                                                                                         \
    asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op \
                    " %6, %6\n" op " %7, %7"                                                                  \
                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
#define OP3_F64(op)                                                                                          \
    asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op \
                    " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9"       \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)               \
                 : "v"(b), "v"(c));
#define OP2_F64(op)                                                                                        \
    asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op             \
                    " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8"                   \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)           \
                 : "v"(b));
#define OP_DIVSCALE                                                                                        \
    asm volatile("v_div_scale_f64 %0, vcc, %0, %8, %0\nv_div_scale_f64 %1, vcc, %1, %8, %1\n"              \
                 "v_div_scale_f64 %2, vcc, %2, %8, %2\nv_div_scale_f64 %3, vcc, %3, %8, %3\n"              \
                 "v_div_scale_f64 %4, vcc, %4, %8, %4\nv_div_scale_f64 %5, vcc, %5, %8, %5\n"              \
                 "v_div_scale_f64 %6, vcc, %6, %8, %6\nv_div_scale_f64 %7, vcc, %7, %8, %7"                \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)           \
                 : "v"(b)                                                                                  \
                 : "vcc");

DEF_KERNEL(k_fma_f64, OP3_F64("v_fma_f64"))
DEF_KERNEL(k_mul_f64, OP2_F64("v_mul_f64"))
DEF_KERNEL(k_add_f64, OP2_F64("v_add_f64"))
DEF_KERNEL(k_rcp_f64, OP1_F64("v_rcp_f64"))
DEF_KERNEL(k_rsq_f64, OP1_F64("v_rsq_f64"))
DEF_KERNEL(k_sqrt_f64, OP1_F64("v_sqrt_f64"))
DEF_KERNEL(k_rcp_f32, OP1_F32("v_rcp_f32"))
DEF_KERNEL(k_rsq_f32, OP1_F32("v_rsq_f32"))
DEF_KERNEL(k_fixup_f64, OP3_F64("v_div_fixup_f64"))
DEF_KERNEL(k_fmas_f64, OP3_F64("v_div_fmas_f64"))
DEF_KERNEL(k_divscale_f64, OP_DIVSCALE)
DEF_KERNEL(k_ldexp_f64, asm volatile("v_ldexp_f64 %0, %0, 1\nv_ldexp_f64 %1, %1, 1\nv_ldexp_f64 %2, %2, 1\nv_ldexp_f64 %3, %3, 1\n"
                                     "v_ldexp_f64 %4, %4, 1\nv_ldexp_f64 %5, %5, 1\nv_ldexp_f64 %6, %6, 1\nv_ldexp_f64 %7, %7, 1"
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
DEF_KERNEL(k_cvt_f32_f64, asm volatile("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\n"
                                       "v_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15"
                                       : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
                                       : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
DEF_KERNEL(k_cvt_f64_f32, asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\n"
                                       "v_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15"
                                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                       : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));)
DEF_KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %1, %1, %2, vcc\nv_cndmask_b32 %2, %2, %3, vcc\nv_cndmask_b32 %3, %3, %4, vcc\n"
                                   "v_cndmask_b32 %4, %4, %5, vcc\nv_cndmask_b32 %5, %5, %6, vcc\nv_cndmask_b32 %6, %6, %7, vcc\nv_cndmask_b32 %7, %7, %0, vcc"
                                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)::"vcc");)
DEF_KERNEL(k_cmp_f64, asm volatile("v_cmp_gt_f64 vcc, %0, %1\nv_cmp_gt_f64 vcc, %1, %2\nv_cmp_gt_f64 vcc, %2, %3\nv_cmp_gt_f64 vcc, %3, %4\n"
                                   "v_cmp_gt_f64 vcc, %4, %5\nv_cmp_gt_f64 vcc, %5, %6\nv_cmp_gt_f64 vcc, %6, %7\nv_cmp_gt_f64 vcc, %7, %0"
                                   ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");)

#define OP3_F32(op)                                                                                          \
    asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op \
                    " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9"       \
                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)               \
                 : "v"((float)b), "v"((float)c));
// single precision, scalar and packed (two floats per lane in a 64-bit register pair: the double registers serve as containers)
DEF_KERNEL(k_fma_f32, OP3_F32("v_fma_f32"))
DEF_KERNEL(k_pk_fma_f32, OP3_F64("v_pk_fma_f32"))
DEF_KERNEL(k_pk_mul_f32, OP2_F64("v_pk_mul_f32"))
DEF_KERNEL(k_pk_add_f32, OP2_F64("v_pk_add_f32"))

typedef void (*kern_t)(double *, unsigned long long *, double);

int main()
{
    struct { const char *name; kern_t k; } ks[] = {
        {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_rcp_f64", k_rcp_f64},
        {"v_rsq_f64", k_rsq_f64}, {"v_sqrt_f64", k_sqrt_f64}, {"v_rcp_f32", k_rcp_f32}, {"v_rsq_f32", k_rsq_f32},
        {"v_div_scale_f64", k_divscale_f64}, {"v_div_fmas_f64", k_fmas_f64}, {"v_div_fixup_f64", k_fixup_f64},
        {"v_ldexp_f64", k_ldexp_f64}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32},
        {"v_cndmask_b32", k_cndmask}, {"v_cmp_gt_f64", k_cmp_f64}, {"v_fma_f32", k_fma_f32},
        {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32}};
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double *out;
    unsigned long long *ticks;
    hipMalloc(&out, sizeof(double) * cus * 8 * 1024);
    hipMalloc(&ticks, sizeof(unsigned long long) * cus * 8);
    // Throughput from the wall clock of the whole launch (HIP events): with several waves per SIMD the
    // in-kernel clock of wave 0 is not representative (the oldest wave wins issue arbitration).
    printf("%-18s %12s | wall-clock ns per wave-instruction per SIMD (x2.4 = cycles at 2.4 GHz)\n", "instruction",
           "1w ticks");
    printf("%-18s %12s %10s %10s %10s %10s\n", "", "(in-kernel)", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto &k : ks) {
        printf("%-18s", k.name);
        bool first = true;
        for (int waves : {1, 2, 4, 8}) {
            const int wg = waves > 4 ? 2 : 1;  // workgroups per CU
            const int threads = 256 * waves / wg;
            hipLaunchKernelGGL(k.k, dim3(cus * wg), dim3(threads), 0, 0, out, ticks, 1.5);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int r = 0; r < 5; ++r)
                hipLaunchKernelGGL(k.k, dim3(cus * wg), dim3(threads), 0, 0, out, ticks, 1.5);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (first) {
                std::vector<unsigned long long> h(cus);
                hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
                double sum = 0;
                for (auto t : h)
                    sum += (double)t;
                printf(" %12.2f", sum / cus / ((double)ITER * 8));
                first = false;
            }
            const double ns = ms * 1e6 / 5.0 / ((double)ITER * 8 * waves);
            printf(" %10.3f", ns);
        }
        printf("\n");
    }
    return 0;
}
