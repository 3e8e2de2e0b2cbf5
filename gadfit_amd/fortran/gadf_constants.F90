! Kinds and constants (mirror of fortran/gadfit/gadf_constants.F90:20-40: the same names, so user modules that
! `use gadf_constants` compile unchanged).
module gadf_constants
  use, intrinsic :: iso_fortran_env, only: real32, real64, real128
  implicit none
  public
  integer, parameter :: dp = real64, qp = real128
  integer, parameter :: kp = dp     ! QUAD_PRECISION is not a GPU type
  real(kp), parameter :: pi = 3.141592653589793238462643383279503_kp
  real(kp), parameter :: pi_2 = 2*pi, pi2 = pi*pi       ! 2 pi and pi**2 (exact doublings / the correctly rounded square: checked)
  real(kp), parameter :: sqrtpi = 1.772453850905516027298167483341145_kp
  real(kp), parameter :: euler = 2.718281828459045235360287471352662_kp
end module gadf_constants
