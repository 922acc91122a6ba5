// rtmi_kernels.h -- internal interface between the C ABI (rtmi_capi.cpp) and the kernels.
#ifndef RTMI_KERNELS_H
#define RTMI_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RTMI_BLOCK 256          // threads per workgroup (4 waves: one per SIMD of a CU)
#define RTMI_BLOCKS_PER_CU 16   // grid cap = CUs x this; further tiles are grid-strided

namespace rtmi {

// Kernel arguments of one trace launch.  Every pointer is a device pointer.
struct TraceArgs {
    const double *d_vels;      // [nmodels][ld_model]
    const double *d_depths;    // [nmodels][ld_model]
    const int32_t *d_nlayers;  // [nmodels] or NULL -> `nlayers` for every model
    int nlayers;
    int ld_model;
    const double *d_src_offset;  // [nsrc]
    const double *d_src_depth;   // [nsrc]
    int64_t nsrc;
    double *d_timeP;    // [nmodels][nsrc] or NULL
    double *d_p_final;  // [nsrc] or NULL (single model only)
    const double *d_obs;  // [nsrc] or NULL
    double *d_partial;    // [nmodels][grid_x] per-workgroup sums of squared residuals
};

}  // namespace rtmi

extern "C" {
int rtmi_launch_grid_x(int64_t nsrc, int nmodels, int num_cus);
hipError_t rtmi_launch_trace(const rtmi::TraceArgs *args, int nmodels, int max_nlayers, int grid_x,
                             hipStream_t stream);
hipError_t rtmi_launch_loglike(const double *d_partial, int nparts, const double *d_sd, int64_t ndat, int nmodels,
                               double *d_logL, hipStream_t stream);
}

#endif
