! Stand-in for `USE MPI` so that the reference's sampler sources can be COMPILE-checked against shim/raymod.f90 in an
! image without an MPI installation: constants only, nothing is linked or run.
module mpi
   implicit none
   integer, parameter :: MPI_COMM_WORLD = 0, MPI_DOUBLE_PRECISION = 1, MPI_INTEGER = 2, MPI_STATUS_SIZE = 4
   integer, parameter :: MPI_ANY_SOURCE = -1, MPI_ANY_TAG = -1, MPI_SOURCE = 1, MPI_TAG = 2, MPI_ADDRESS_KIND = 8
   integer, parameter :: MPI_REAL = 3, MPI_LOGICAL = 4, MPI_SUM = 5, MPI_MAX = 6, MPI_CHARACTER = 7
end module mpi
