!  shim/test_shim.f90 -- Fortran caller in the style of LOGLHOOD_RT (loglhood.f90:128-146):
!  USE RAYMOD ; CALL TraceRays(vels, thickness, k-1, src_offset, src_depth, NSRC, DpredRT, -1)
!  Reads a model and sources from stdin, prints the travel times with 17 significant digits.
!     line 1: NLayers NSrc ; line 2: vels(1:NLayers+1) ; line 3: depths(1:NLayers)
!     then NSrc lines "offset depth" (the layout of test_1_src_data.txt, read_input.f90:376)
program test_shim
   use raymod
   implicit none
   integer :: NLayers, NSrc, i
   double precision, allocatable :: vels(:), depths(:), src_offset(:), src_depth(:), DpredRT(:)
   read(*,*) NLayers, NSrc
   allocate(vels(NLayers+1), depths(NLayers), src_offset(NSrc), src_depth(NSrc), DpredRT(NSrc))
   read(*,*) vels
   read(*,*) depths
   do i = 1, NSrc
      read(*,*) src_offset(i), src_depth(i)
   end do
   DpredRT = 0
   call TraceRays(vels, depths, NLayers, src_offset, src_depth, NSrc, DpredRT, -1)
   do i = 1, NSrc
      write(*,'(ES25.17)') DpredRT(i)
   end do
   ! the batched likelihood with the predictions as "observations": SS = 0, logL = -(N/2) ln 2pi - N ln sd
   block
      use rtmi_likelihood
      use, intrinsic :: iso_c_binding, only : c_int32_t, c_int64_t, c_int
      double precision :: v2(NLayers+2, 1), d2(NLayers+2, 1), sd(1), logL(1), ss(1)
      integer(c_int32_t) :: nl(1)
      integer(c_int) :: rc
      v2 = 0; d2 = 0
      v2(1:NLayers+1, 1) = vels
      d2(1:NLayers, 1) = depths
      nl(1) = NLayers
      sd(1) = 0.02d0
      rc = rtmi_set_sources(src_offset, src_depth, DpredRT, int(NSrc, c_int64_t))
      if (rc == 0) rc = rtmi_loglhood_batch(v2, d2, nl, int(NLayers+2, c_int), 1_c_int, sd, logL, ss)
      write(*,'(A,I3,2ES25.17)') 'LOGL ', rc, logL(1), ss(1)
   end block
   ! the sampler resident on the GPU, driven from Fortran (module rtmi_sampler): four chains start at the model, a few
   ! sweeps with temperature exchanges through a one-rank RCCL communicator, states read back
   block
      use rtmi_sampler
      use, intrinsic :: iso_c_binding, only : c_int, c_int64_t, c_ptr, c_char, c_associated, c_double
      type(rtmi_pt_prior) :: pr
      type(c_ptr) :: h, comm
      integer(c_int) :: rc, i, k(4), level(4)
      character(kind=c_char) :: id(RTMI_COMM_ID_BYTES)
      real(c_double) :: z(NLayers+1), beta_of_level(4), zz(10, 4), vv(10, 4), sd(4), logL(4), ss(4), beta(4), logL0
      pr = rtmi_pt_prior(1, 10, 3.01d0, 10000.1d0, 100.1d0, 1500.0d0, 10000.0d0, 0.001d0, 0.07d0, 30.0d0, 10.0d0)
      h = rtmi_pt_create(4_c_int, 11_c_int, int(NSrc, c_int64_t), pr, 7_c_int64_t, 0_c_int64_t)
      rc = merge(0, 1, c_associated(h))
      z(1) = 0.0d0
      z(2:NLayers+1) = depths
      do i = 0, 3
         if (rc == 0) rc = rtmi_pt_set_chain(h, i, int(NLayers+1, c_int), z, vels, 0.02d0, 1.0d0)
      end do
      if (rc == 0) rc = rtmi_pt_evaluate(h)
      if (rc == 0) rc = rtmi_pt_get_all(h, k, zz, vv, sd, logL, ss, beta)
      logL0 = logL(1)
      level = (/ 0, 1, 2, 3 /)
      beta_of_level = (/ 1.0d0, 0.7d0, 0.5d0, 0.35d0 /)
      if (rc == 0) rc = rtmi_pt_set_ladder(h, 4_c_int, 0_c_int, level, beta_of_level, 4_c_int)
      if (rc == 0) rc = rtmi_comm_unique_id(id)
      comm = rtmi_comm_create(1_c_int, 0_c_int, id)
      if (rc == 0) rc = merge(0, 1, c_associated(comm))
      if (rc == 0) rc = rtmi_pt_run(h, 20_c_int, 1_c_int, comm, 0_c_int)
      if (rc == 0) rc = rtmi_pt_get_all(h, k, zz, vv, sd, logL, ss, beta)
      write(*,'(A,I3,ES25.17,4I4,4F8.4)') 'PT ', rc, logL0, k, beta
      call rtmi_comm_destroy(comm)
      call rtmi_pt_destroy(h)
   end block
end program test_shim
