! GPU parity test through the Fortran API: nested integrals with an infinite outer bound, an
! ACTIVE inner upper bound and erf, user-supplied errors (third data column).  Known answer of
! the reference (fortran/tests/3_integral_double.F90:96): a = 8.5799477799920343, tolerance 1e-9.
module nested_model
  use ad
  use fitfunction
  use gadf_constants
  use numerical_integration
  implicit none
  type, extends(fitfunc) :: nested_t
   contains
     procedure :: init => nested_init
     procedure :: eval => nested_eval
  end type nested_t
contains
  subroutine nested_init(this)
    class(nested_t), intent(out) :: this
    allocate(this%pars(2))
    call this%set(1, 'a'); call this%set(2, 'b')
  end subroutine nested_init

  type(advar) function nested_eval(this, x) result(y)
    class(nested_t), intent(in) :: this
    real(kp), intent(in) :: x
    type(advar) :: q(3)
    q(1) = this%pars(1); q(2) = this%pars(2); q(3) = x
    y = integrate(outer, q, 0.0_kp, INFINITY)/x
  end function nested_eval

  type(advar) function outer(t, q) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: q(:)
    type(advar) :: a, b, xx, q2(1)
    a = q(1); b = q(2); xx = q(3)
    q2(1) = 1 + b*a*erf(t)
    y = integrate(inner, q2, 0.0_kp, xx/b)
    y = exp(-t)*y
  end function outer

  type(advar) function inner(t, q) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: q(:)
    type(advar) :: c
    c = q(1)
    y = log((exp(t) - 1.0_kp)*c + 1.0_kp)/t
  end function inner
end module nested_model

program fit_integral_double
  use nested_model
  use gadfit
  implicit none
  type(nested_t) :: f
  character(len=512) :: path
  real(kp), parameter :: golden = 8.5799477799920343_kp
  call get_command_argument(1, path)
  call gadf_init(f, ad_memory='10 MB', rel_error_inner=1e-6_kp, rel_error=1e-5_kp)
  call gadf_set_errors(USER)
  call gadf_add_dataset(trim(path))
  call gadf_set('a', 1.0, .true.)
  call gadf_set('b', 1.0, .true.)
  call gadf_set_verbosity(output='/dev/null')
  call gadf_fit(0.1, accth=0.9, max_iter=3)
  write(*, '(a, es25.17)') 'a = ', fitfuncs(1)%pars(1)%val
  if (abs(fitfuncs(1)%pars(1)%val - golden) > 1e-9_kp) error stop 'a differs from the reference golden value'
  call gadf_close()
  print '(a)', 'PASS'
end program fit_integral_double
