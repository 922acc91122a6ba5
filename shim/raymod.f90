!  shim/raymod.f90 -- drop-in `module raymod` over librtmi.so (ISO_C_BINDING -> C ABI -> HIP).
!
!  The reference's callers do `USE RAYMOD` and `CALL TraceRays(vels, thickness, k-1, src_offset,
!  src_depth, NSRC, DpredRT, -1)` (ray_tracing_sampling/loglhood.f90:42,135,144).  TraceRays is a
!  MODULE procedure, so its link name and .mod file are compiler specific; a binary replacement is
!  therefore shipped as this source: compile it with the same compiler as the callers IN PLACE OF
!  subroutineR-quiet.f90 and link with -lrtmi.  loglhood.f90, replica.f90 and prjmh_temper_rf.f90
!  then build unchanged.
!
!     mpifort -c data_type.f90 rjmcmc_com.f90 ... shim/raymod.f90 loglhood.f90 ...
!     mpifort -o prjmh_temper_rf *.o -L<repo>/raytracerfortran_amd -lrtmi -Wl,-rpath,<repo>/raytracerfortran_amd
!
!  Exposed names and argument lists are those of the reference module
!  (subroutineR-quiet.f90:5, :408-425, :467-482).  The helper procedures of the reference's public
!  list (whichLayer, InsertLayer, GetPTime, costFunc, costFunc_Prime, solve, solvebst) are internal
!  steps of the solver, are not called from outside the module anywhere in the reference, and have
!  no counterpart here: the whole per-source computation is one GPU kernel.
module raymod
   use, intrinsic :: iso_c_binding, only : c_double, c_int
   implicit none
   private
   public :: TraceRays, dofullforwardproblem

   interface
      ! include/rtmi.h: void rtmi_tracerays(const double*, const double*, const int*, double*, double*,
      !                                     const int*, double*, const int*)
      subroutine rtmi_tracerays(vels, depths, NLayers, src_offset, src_depth, NSrc, timeP, keep_delta) &
            bind(C, name="rtmi_tracerays")
         import :: c_double, c_int
         integer(c_int), intent(in) :: NLayers, NSrc, keep_delta
         real(c_double), intent(in) :: vels(*), depths(*)
         real(c_double) :: src_offset(*), src_depth(*)
         real(c_double), intent(out) :: timeP(*)
      end subroutine rtmi_tracerays
      ! include/rtmi.h: void dff_(...)  -- the symbol R's .Fortran("dff", ...) resolves
      subroutine rtmi_dff(vels, depths, NLayers, src_offset, src_depth, NSrc, timeP, keep_delta) &
            bind(C, name="dff_")
         import :: c_double, c_int
         integer(c_int), intent(in) :: NLayers, NSrc, keep_delta
         real(c_double), intent(in) :: vels(*), depths(*)
         real(c_double) :: src_offset(*), src_depth(*)
         real(c_double), intent(out) :: timeP(*)
      end subroutine rtmi_dff
   end interface

contains

   ! subroutineR-quiet.f90:467-482
   subroutine TraceRays(vels, depths, NLayers, src_offset, src_depth, NSrc, timeP, keep_delta)
      integer, intent(in) :: NLayers, NSrc
      integer, intent(in) :: keep_delta
      double precision, dimension(NSrc) :: src_offset, src_depth
      double precision, dimension(NSrc), intent(out) :: timeP
      double precision, dimension(NLayers+1), intent(in) :: vels
      double precision, dimension(NLayers), intent(in) :: depths
      integer(c_int) :: nl_c, ns_c, kd_c
      nl_c = int(NLayers, c_int)
      ns_c = int(NSrc, c_int)
      kd_c = int(keep_delta, c_int)
      call rtmi_tracerays(vels, depths, nl_c, src_offset, src_depth, ns_c, timeP, kd_c)
   end subroutine TraceRays

   ! subroutineR-quiet.f90:408-425 (Fortran-side name of the C symbol dff_)
   subroutine dofullforwardproblem(vels, depths, NLayers, src_offset, src_depth, NSrc, timeP, keep_delta)
      integer(c_int), intent(in) :: NLayers, NSrc
      integer(c_int), intent(in) :: keep_delta
      real(c_double), dimension(NSrc) :: src_offset, src_depth
      real(c_double), dimension(NSrc), intent(out) :: timeP
      real(c_double), dimension(NLayers+1), intent(in) :: vels
      real(c_double), dimension(NLayers), intent(in) :: depths
      call rtmi_dff(vels, depths, NLayers, src_offset, src_depth, NSrc, timeP, keep_delta)
   end subroutine dofullforwardproblem

end module raymod


!  Optional extension for samplers that want the whole likelihood on the GPU: the source set is uploaded once
!  (what READDATA reads, read_input.f90:302-387) and LOGLHOOD is then evaluated for a batch of models in one launch
!  (loglhood.f90:35-207).  Interfaces to include/rtmi.h; arrays are column-major on the Fortran side, so a model
!  is a COLUMN of vels / depths: vels(ld_model, nmodels), depths(ld_model, nmodels).
module rtmi_likelihood
   use, intrinsic :: iso_c_binding, only : c_double, c_int, c_int32_t, c_int64_t
   implicit none
   interface
      integer(c_int) function rtmi_set_sources(src_offset, src_depth, obs, nsrc) bind(C, name="rtmi_set_sources")
         import :: c_double, c_int, c_int64_t
         real(c_double), intent(in) :: src_offset(*), src_depth(*), obs(*)
         integer(c_int64_t), value :: nsrc
      end function rtmi_set_sources
      integer(c_int) function rtmi_loglhood_batch(vels, depths, nlayers, ld_model, nmodels, sd, logL, ss) &
            bind(C, name="rtmi_loglhood_batch")
         import :: c_double, c_int, c_int32_t
         real(c_double), intent(in) :: vels(*), depths(*), sd(*)
         integer(c_int32_t), intent(in) :: nlayers(*)
         integer(c_int), value :: ld_model, nmodels
         real(c_double), intent(out) :: logL(*), ss(*)
      end function rtmi_loglhood_batch
   end interface
end module rtmi_likelihood
