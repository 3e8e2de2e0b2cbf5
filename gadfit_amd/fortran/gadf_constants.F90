! Kinds and constants (mirror of fortran/gadfit/gadf_constants.F90:20-33).
module gadf_constants
  use, intrinsic :: iso_fortran_env, only: real32, real64, real128
  implicit none
  public
  integer, parameter :: dp = real64, qp = real128
  integer, parameter :: kp = dp     ! QUAD_PRECISION is not a GPU type
  real(kp), parameter :: pi = 3.141592653589793238462643383279503_kp
  real(kp), parameter :: sqrtpi = 1.772453850905516027298167483341145_kp
end module gadf_constants
