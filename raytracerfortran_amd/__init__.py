"""raytracerfortran_amd -- MI355X-native 1-D layered-earth travel-time forward model.

One hot path, behind the reference's own call boundary:
    raymod.dff / raymod.TraceRays        host buffers in/out  (C symbols dff_, rtmi_tracerays)
    raymod.trace_device / trace_batch_device   HBM-resident   (rtmi_trace_device, rtmi_trace_batch_device)
Kernels and the C ABI live in csrc/ and are built in-tree into librtmi.so by build.py.
"""
from . import raymod  # noqa: F401
from .raymod import (RtmiError, TraceRays, comm_rccl_library, comm_rccl_version, dff, dff7, dofullforwardproblem, init, last_kernel_name,  # noqa: F401
                     load_library, loglhood_batch, set_option, set_sources, trace_batch_device, trace_device, write_rays_dat)

__version__ = "0.4.0"
