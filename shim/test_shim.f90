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
end program test_shim
