! A real that eval() forms from the %val of a FITTED parameter: s = sin(tau%val) enters the model through plain real arithmetic.
! The reference recomputes it whenever eval() runs (at every point of every pass, gadfit.F90:679-690) and its AD never sees it: no
! derivative flows through %val, so the Jacobian ignores that dependence while the residuals follow it.  The recorder cannot look
! inside the real arithmetic; gadf_fit finds the literal that moves with the parameters (probe_pars), declares a passive
! pseudo-parameter for it (GFH_VAL(GFH_PARAM(n)), include/gadfit_tape.h) and recomputes it on the host before every pass
! (gfh_set_pars_hook -> on_pars: one recording of eval() per dataset).  Rounds 1-3 refused this program.
! Expected values: the oracle's fit of the same model written with value() = GFH_VAL (tests/golden/make_branching_goldens.py, case
! param_val); same data by the same formula.
module param_val_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: pv_t
   contains
     procedure :: init => pv_init
     procedure :: eval => pv_eval
  end type pv_t
contains
  subroutine pv_init(this)
    class(pv_t), intent(out) :: this
    allocate(this%pars(3))
    call this%set(1, 'amp'); call this%set(2, 'tau'); call this%set(3, 'bgr')
  end subroutine pv_init

  type(advar) function pv_eval(this, x) result(y)
    class(pv_t), intent(in) :: this
    real(kp), intent(in) :: x
    real(kp) :: s
    s = sin(this%pars(2)%val)
    y = this%pars(1)*exp(-(x/this%pars(2)))*(1.0_kp + 0.05_kp*s*s) + this%pars(3)
  end function pv_eval
end module param_val_model

program fit_param_val
  use param_val_model
  use gadfit
  implicit none
  integer, parameter :: n = 400
  type(pv_t) :: f
  real(kp) :: x(n), y(n), s
  real(kp), parameter :: truth(3) = [5.0_kp, 20.0_kp, 1.0_kp]
  real(kp), parameter :: expected(3) = [4.9999048471680254_kp, 20.000394474341594_kp, 0.99998323488968721_kp]
  integer :: i
  logical :: ok
  s = sin(truth(2))
  do i = 1, n
     x(i) = 0.5_kp + 99.0_kp*real(i - 1, kp)/real(n - 1, kp)
     y(i) = truth(1)*exp(-(x(i)/truth(2)))*(1.0_kp + 0.05_kp*s*s) + truth(3) + 1.0e-3_kp*sin(real(mod(37*(i - 1), 1000), kp))
  end do
  call gadf_init(f)
  call gadf_add_dataset(x, y)
  call gadf_set('amp', 4.5_kp, .true.)
  call gadf_set('tau', 22.0_kp, .true.)
  call gadf_set('bgr', 1.2_kp, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output="/dev/null")
  call gadf_fit(1.0, max_iter=8)
  ok = gadf_iterations == 8
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
end program fit_param_val
