! GPU parity test through the Fortran API: a fitting function that is an integral,
! pi * int_0^x t**a * exp(-b*t**2) dt, differentiated through the adaptive Gauss-Kronrod rule
! on the device.  Known answer of the reference (fortran/tests/2_integral_single.F90:74):
! a = 7.5549166396989014.  Data file = argument 1.
module integral_model
  use ad
  use fitfunction
  use gadf_constants
  use numerical_integration
  implicit none
  type, extends(fitfunc) :: integral_t
   contains
     procedure :: init => integral_init
     procedure :: eval => integral_eval
  end type integral_t
contains
  subroutine integral_init(this)
    class(integral_t), intent(out) :: this
    allocate(this%pars(2))
    call this%set(1, 'a'); call this%set(2, 'b')
  end subroutine integral_init

  type(advar) function integral_eval(this, x) result(y)
    class(integral_t), intent(in) :: this
    real(kp), intent(in) :: x
    type(advar) :: q(2)
    q(1) = this%pars(1); q(2) = this%pars(2)
    y = pi*integrate(kernel, q, 0.0_kp, x)
  end function integral_eval

  type(advar) function kernel(t, q) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: q(:)
    y = t**q(1)*exp(-q(2)*t**2)
  end function kernel
end module integral_model

program fit_integral_single
  use integral_model
  use gadfit
  implicit none
  type(integral_t) :: f
  character(len=512) :: path
  real(kp), parameter :: golden = 7.5549166396989014_kp
  call get_command_argument(1, path)
  call gadf_init(f, rel_error=1e-12_kp)
  call gadf_add_dataset(trim(path))
  call gadf_set('a', 10.0, .true.)
  call gadf_set('b', 1.0, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output='/dev/null')
  call gadf_fit(10.0, accth=0.9, max_iter=6, rel_error=1e-6)
  write(*, '(a, es25.17)') 'a = ', fitfuncs(1)%pars(1)%val
  if (abs(fitfuncs(1)%pars(1)%val - golden) > 1e-10_kp*golden) error stop 'a differs from the reference golden value'
  call gadf_close()
  print '(a)', 'PASS'
end program fit_integral_single
