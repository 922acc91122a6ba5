!  shim/raymod.f90 -- drop-in `module raymod` over librtmi.so (ISO_C_BINDING -> C ABI -> HIP).
!
!  The reference's callers do `USE RAYMOD` and `CALL TraceRays(vels, thickness, k-1, src_offset,
!  src_depth, NSRC, DpredRT, -1)` (ray_tracing_sampling/loglhood.f90:42,135,144).  TraceRays is a
!  MODULE procedure, so its link name and .mod file are compiler specific; a binary replacement is
!  therefore shipped as this source: compile it with the same compiler as the callers IN PLACE OF
!  subroutineR-quiet.f90 and link with -lrtmi.  loglhood.f90 -- the only file of the sampler that
!  names raymod -- then builds unchanged (tests/test_boundary.py compile-checks exactly that).
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


!  The sampler itself on the GPU (include/rtmi.h, "Metropolis-Hastings sweep ... RESIDENT ON THE GPU"): what
!  prjmh_temper_rf.f90 does per MCMC step -- EXPLORE_MH_NOVARPAR (:637-761), EXPLORE_MH (:521-634) and the temperature
!  exchange TEMPSWP_MH (:1329-1357) -- for the chains of this rank, plus the RCCL communicator that replaces the
!  MPI_SEND / MPI_RECV pairs through rank 0 (:331-349).  A host program keeps its MPI_INIT / file handling, broadcasts
!  the 128-byte id with MPI_BCAST and calls:
!     h = rtmi_pt_create(nchains, ld_model, ndata, prior, seed, rank)      ! after rtmi_set_sources
!     rc = rtmi_pt_set_chain(h, i, k, depths, vels, sd, beta) ...; rc = rtmi_pt_evaluate(h)
!     rc = rtmi_pt_set_ladder(h, nchains_total, first_chain, level, beta_of_level, nlevels)
!     rc = rtmi_pt_run(h, nsweeps, swap_every, comm, record_every)         ! no host synchronisation inside
module rtmi_sampler
   use, intrinsic :: iso_c_binding, only : c_double, c_int, c_int64_t, c_ptr, c_char, c_null_ptr
   implicit none
   integer, parameter :: RTMI_COMM_ID_BYTES = 128
   type, bind(C) :: rtmi_pt_prior
      integer(c_int) :: kmin, kmax                    ! NLMN, NLMX                read_input.f90:39-40
      real(c_double) :: lambda                        ! Poisson prior on k        read_input.f90:78-81
      real(c_double) :: hmx, hmin                     !                           read_input.f90:46-47
      real(c_double) :: vmin, vmax                    ! Vp bounds                 read_input.f90:207-208
      real(c_double) :: sdmin, sdmax                  ! sigma bounds              read_input.f90:234-236
      real(c_double) :: pertsdsc, pertsdsdsc          ! step scales               read_input.f90:169,237
   end type rtmi_pt_prior
   interface
      type(c_ptr) function rtmi_pt_create(nchains, ld_model, ndata, prior, seed, stream_id) bind(C, name="rtmi_pt_create")
         import :: c_ptr, c_int, c_int64_t, rtmi_pt_prior
         integer(c_int), value :: nchains, ld_model
         integer(c_int64_t), value :: ndata, seed, stream_id
         type(rtmi_pt_prior), intent(in) :: prior
      end function rtmi_pt_create
      subroutine rtmi_pt_destroy(h) bind(C, name="rtmi_pt_destroy")
         import :: c_ptr
         type(c_ptr), value :: h
      end subroutine rtmi_pt_destroy
      integer(c_int) function rtmi_pt_set_chain(h, chain, k, depths, vels, sd, beta) bind(C, name="rtmi_pt_set_chain")
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: h
         integer(c_int), value :: chain, k                 ! chain is 0-based
         real(c_double), intent(in) :: depths(*), vels(*)
         real(c_double), value :: sd, beta
      end function rtmi_pt_set_chain
      integer(c_int) function rtmi_pt_evaluate(h) bind(C, name="rtmi_pt_evaluate")
         import :: c_ptr, c_int
         type(c_ptr), value :: h
      end function rtmi_pt_evaluate
      integer(c_int) function rtmi_pt_set_ladder(h, nchains_total, first_chain, level, beta_of_level, nlevels) &
            bind(C, name="rtmi_pt_set_ladder")
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: h
         integer(c_int), value :: nchains_total, first_chain, nlevels
         integer(c_int), intent(in) :: level(*)
         real(c_double), intent(in) :: beta_of_level(*)
      end function rtmi_pt_set_ladder
      ! new betas for the ladder's levels in place (the burn-in re-spacing of prjmh_temper_rf.f90:363-383: after the master
      ! has multiplied dTlog, beta_of_level(l) = 1/dTlog**(l-1)); levels, exchange counters and random stream are kept
      integer(c_int) function rtmi_pt_set_ladder_betas(h, beta_of_level, nlevels) bind(C, name="rtmi_pt_set_ladder_betas")
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: h
         real(c_double), intent(in) :: beta_of_level(*)
         integer(c_int), value :: nlevels
      end function rtmi_pt_set_ladder_betas
      integer(c_int) function rtmi_pt_run(h, nsweeps, swap_every, comm, record_every) bind(C, name="rtmi_pt_run")
         import :: c_ptr, c_int
         type(c_ptr), value :: h, comm                     ! comm = c_null_ptr when every chain is local
         integer(c_int), value :: nsweeps, swap_every, record_every
      end function rtmi_pt_run
      integer(c_int) function rtmi_pt_swap_round(h, comm) bind(C, name="rtmi_pt_swap_round")
         import :: c_ptr, c_int
         type(c_ptr), value :: h, comm
      end function rtmi_pt_swap_round
      integer(c_int) function rtmi_pt_get_all(h, k, depths, vels, sd, logL, ss, beta) bind(C, name="rtmi_pt_get_all")
         import :: c_ptr, c_int, c_double
         type(c_ptr), value :: h
         integer(c_int), intent(out) :: k(*)
         real(c_double), intent(out) :: depths(*), vels(*), sd(*), logL(*), ss(*), beta(*)   ! depths/vels: (kmax, nchains)
      end function rtmi_pt_get_all
      ! the acceptance counters chain by chain (the reference's obj%iaccept_bd ..., prjmh_temper_rf.f90:1880-1883): stats(8, nchains) =
      ! accepted / proposed birth-death, accepted / proposed perturbations, accepted / proposed sigma, models decided, discarded
      integer(c_int) function rtmi_pt_chain_stats(h, stats) bind(C, name="rtmi_pt_chain_stats")
         import :: c_ptr, c_int, c_int64_t
         type(c_ptr), value :: h
         integer(c_int64_t), intent(out) :: stats(8, *)
      end function rtmi_pt_chain_stats
      integer(c_int) function rtmi_comm_unique_id(id) bind(C, name="rtmi_comm_unique_id")
         import :: c_int, c_char
         character(kind=c_char), intent(out) :: id(*)       ! RTMI_COMM_ID_BYTES bytes: MPI_BCAST them from rank 0
      end function rtmi_comm_unique_id
      type(c_ptr) function rtmi_comm_create(nranks, rank, id) bind(C, name="rtmi_comm_create")
         import :: c_ptr, c_int, c_char
         integer(c_int), value :: nranks, rank
         character(kind=c_char), intent(in) :: id(*)
      end function rtmi_comm_create
      subroutine rtmi_comm_destroy(comm) bind(C, name="rtmi_comm_destroy")
         import :: c_ptr
         type(c_ptr), value :: comm
      end subroutine rtmi_comm_destroy
   end interface
end module rtmi_sampler
