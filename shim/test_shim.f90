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
end program test_shim
