! A real that an INTEGRAND forms from the %val of a FITTED parameter: the function handed to integrate() computes s = sin(pars(2)%val)
! in plain real arithmetic from its own pars(:).  The reference evaluates the integrand afresh at every abscissa of the quadrature of
! every point of every pass (numerical_integration.F90:195-201, 636-664), so s follows the parameter there, and its AD never sees it: no
! derivative flows through %val.  Here the recorder meets s as a literal of the integrand's sub-tape; gadf_fit finds that it moves with
! the parameters (probe_pars) and hands it to the integrand as ONE MORE ENTRY OF ITS pars(:), passive, bound at the call site to a
! pseudo-parameter that on_pars recomputes on the host before every pass (build_tape; the mechanism of fit_param_val.F90 one level
! down).  Rounds 1-4 refused this program.
! Expected values: the oracle's fit of the same model written with value() = GFH_VAL inside the integrand
! (tests/golden/make_branching_goldens.py, case integrand_param_val); same data by the same formula.
module integrand_pval_model
  use ad
  use fitfunction
  use gadf_constants
  use numerical_integration
  implicit none
  type, extends(fitfunc) :: ipv_t
   contains
     procedure :: init => ipv_init
     procedure :: eval => ipv_eval
  end type ipv_t
contains
  subroutine ipv_init(this)
    class(ipv_t), intent(out) :: this
    allocate(this%pars(3))
    call this%set(1, 'amp'); call this%set(2, 'rate'); call this%set(3, 'bgr')
  end subroutine ipv_init

  type(advar) function ipv_integrand(t, pars) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: pars(:)
    real(kp) :: s
    s = sin(pars(2)%val)
    y = pars(1)*exp(-(pars(2)*t*t))*(1.0_kp + 0.05_kp*s*s)
  end function ipv_integrand

  type(advar) function ipv_eval(this, x) result(y)
    class(ipv_t), intent(in) :: this
    real(kp), intent(in) :: x
    type(advar) :: q(2)
    q(1) = this%pars(1); q(2) = this%pars(2)
    y = integrate(ipv_integrand, q, 0.0_kp, x) + this%pars(3)
  end function ipv_eval
end module integrand_pval_model

program fit_integrand_param_val
  use integrand_pval_model
  use gadfit
  implicit none
  integer, parameter :: n = 300
  type(ipv_t) :: f
  real(kp) :: x(n), y(n), s
  real(kp), parameter :: truth(3) = [1.3_kp, 0.7_kp, 0.2_kp]
  real(kp), parameter :: expected(3) = [1.3000344081217667_kp, 0.70001694065755293_kp, 0.19998275715945979_kp]
  integer :: i
  logical :: ok
  s = sin(truth(2))
  do i = 1, n
     x(i) = 0.1_kp + 2.9_kp*real(i - 1, kp)/real(n - 1, kp)
     y(i) = truth(1)*(1.0_kp + 0.05_kp*s*s)*0.5_kp*sqrt(pi/truth(2))*erf(x(i)*sqrt(truth(2))) + truth(3) &
          & + 1.0e-3_kp*sin(real(mod(37*(i - 1), 1000), kp))
  end do
  call gadf_init(f, rel_error=1e-10_kp)
  call gadf_add_dataset(x, y)
  call gadf_set('amp', 1.1_kp, .true.)
  call gadf_set('rate', 0.8_kp, .true.)
  call gadf_set('bgr', 0.0_kp, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output="/dev/null")
  call gadf_fit(1.0, max_iter=6)
  ok = gadf_iterations == 6
  do i = 1, 3
     write(*, '(a, i0, a, es25.17, a, es10.2)') 'par ', i, ' = ', fitfuncs(1)%pars(i)%val, '   rel. dev. ', &
          & abs(fitfuncs(1)%pars(i)%val - expected(i))/abs(expected(i))
     ok = ok .and. abs(fitfuncs(1)%pars(i)%val - expected(i)) <= 1e-9_kp*abs(expected(i))
  end do
  call gadf_close()
  if (ok) then
     print '(a)', 'PASS'
  else
     print '(a)', 'FAIL'
     error stop 1
  end if
end program fit_integrand_param_val
