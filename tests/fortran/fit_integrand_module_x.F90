! An integrand that takes the data point's abscissa from the ENCLOSING eval() without passing it through pars(:): eval() parks x in a
! module variable and the function handed to integrate() reads it -- once in affine form (1 + 0.1 x) and once through a real
! function (sin(0.3 x)).  The reference evaluates the integrand afresh, in that scope, at every point and every abscissa of the
! quadrature (numerical_integration.F90:195-201), so this is a valid program there.  Here the two reals reach the recorder as
! literals of the integrand's sub-tape; gadf_fit classifies them over the data like eval()'s own literals (affine in x -> rebuilt
! from the X node; neither constant nor affine -> an auxiliary per-point column, tabulated) and the device's integrand reads the
! point's abscissa and column back from its lane's stash (codegen.cpp, GFH_LANE_STASH).  Rounds 1-3 refused this loudly.
! Expected values: the oracle's fit (tests/golden/make_branching_goldens.py, case integrand_module_x; same data by the same formula).
! usage: fit_integrand_module_x [N]
module module_x_model
  use ad
  use fitfunction
  use gadf_constants
  use numerical_integration
  implicit none
  real(kp) :: x_now = 0.0_kp                 ! the abscissa of the point eval() is working on, for the integrand
  type, extends(fitfunc) :: mx_t
   contains
     procedure :: init => mx_init
     procedure :: eval => mx_eval
  end type mx_t
contains
  subroutine mx_init(this)
    class(mx_t), intent(out) :: this
    allocate(this%pars(3))
    call this%set(1, 'amp'); call this%set(2, 'rate'); call this%set(3, 'bgr')
  end subroutine mx_init

  type(advar) function mx_integrand(t, pars) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: pars(:)
    y = pars(1)*exp(-(pars(2)*t*t))*(1.0_kp + 0.1_kp*x_now) + sin(0.3_kp*x_now)*t
  end function mx_integrand

  type(advar) function mx_eval(this, x) result(y)
    class(mx_t), intent(in) :: this
    real(kp), intent(in) :: x
    type(advar) :: q(2)
    x_now = x
    q(1) = this%pars(1); q(2) = this%pars(2)
    y = integrate(mx_integrand, q, 0.0_kp, x) + this%pars(3)
  end function mx_eval
end module module_x_model

program fit_integrand_module_x
  use module_x_model
  use gadfit
  implicit none
  integer :: n
  type(mx_t) :: f
  real(kp), allocatable :: x(:), y(:)
  character(len=32) :: arg
  real(kp), parameter :: truth(3) = [1.3_kp, 0.7_kp, 0.2_kp]
  real(kp), parameter :: expected(3) = [1.3000258024206_kp, 0.70001281165894735_kp, 0.19998586379590264_kp]
  integer :: i
  logical :: ok
  ! (an argument: that many points instead of the 300 the expected values belong to -- the per-point column is then tabulated on
  ! threads, which the test compares with the serial tabulation bit for bit)
  n = 300
  if (command_argument_count() >= 1) then; call get_command_argument(1, arg); read(arg, *) n; end if
  allocate(x(n), y(n))
  do i = 1, n
     x(i) = 0.1_kp + 2.9_kp*real(i - 1, kp)/real(n - 1, kp)
     y(i) = truth(1)*(1.0_kp + 0.1_kp*x(i))*0.5_kp*sqrt(pi/truth(2))*erf(x(i)*sqrt(truth(2))) + 0.5_kp*sin(0.3_kp*x(i))*x(i)*x(i) + truth(3) &
          & + 1.0e-3_kp*sin(real(mod(37*(i - 1), 1000), kp))
  end do
  call gadf_init(f, rel_error=1e-10_kp)
  call gadf_add_dataset(x, y)
  call gadf_set('amp', 1.1_kp, .true.)
  call gadf_set('rate', 0.8_kp, .true.)
  call gadf_set('bgr', 0.0_kp, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output="/dev/null")
  call gadf_fit(1.0, accth=0.9, max_iter=6)
  ok = gadf_iterations == 6
  do i = 1, 3
     write(*, '(a, i0, a, es25.17, a, es10.2)') 'par ', i, ' = ', fitfuncs(1)%pars(i)%val, '   rel. dev. ', &
          & abs(fitfuncs(1)%pars(i)%val - expected(i))/abs(expected(i))
     ok = ok .and. (n /= 300 .or. abs(fitfuncs(1)%pars(i)%val - expected(i)) <= 1e-9_kp*abs(expected(i)))
  end do
  call gadf_close()
  if (ok) then
     print '(a)', 'PASS'
  else
     print '(a)', 'FAIL'
     error stop 1
  end if
end program fit_integrand_module_x
