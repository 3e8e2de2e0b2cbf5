! An eval() whose plain real(kp) arithmetic on the abscissa looks like a CONSTANT at the three abscissas the recorder probes
! (first, second and last data point): a narrow bump exp(-((x-5)/0.05)**2) underflows to exactly 0 at x = 0, 0.005 and 10.
! The reference evaluates eval() at every point; here the classification is verified against every data point
! (gadfit.F90: verify_capture), the literal is promoted to an auxiliary per-point column and the fit sees the bump.
! y = 3 b(x) + 0.5 x + 1 without noise, so the linear fit must return (3, 0.5, 1).
module bump_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: bump_t
   contains
     procedure :: init => bump_init
     procedure :: eval => bump_eval
  end type bump_t
contains
  subroutine bump_init(this)
    class(bump_t), intent(out) :: this
    allocate(this%pars(3))
    call this%set(1, 'amp'); call this%set(2, 'slope'); call this%set(3, 'offset')
  end subroutine bump_init

  type(advar) function bump_eval(this, x) result(y)
    class(bump_t), intent(in) :: this
    real(kp), intent(in) :: x
    real(kp) :: b
    b = exp(-((x - 5.0_kp)/0.05_kp)**2)          ! real arithmetic: invisible to the recorder
    y = this%pars(1)*b + this%pars(2)*x + this%pars(3)
  end function bump_eval
end module bump_model

program fit_narrow_bump
  use bump_model
  use gadfit
  implicit none
  type(bump_t) :: f
  integer, parameter :: n = 2001
  real(kp), target, save :: xs(n), ys(n)
  integer :: i
  logical :: ok
  do i = 1, n
     xs(i) = 0.005_kp*(i - 1)
     ys(i) = 3.0_kp*exp(-((xs(i) - 5.0_kp)/0.05_kp)**2) + 0.5_kp*xs(i) + 1.0_kp
  end do
  call gadf_init(f)
  call gadf_add_dataset(xs, ys)
  call gadf_set('amp', 1.0, .true.)
  call gadf_set('slope', 1.0, .true.)
  call gadf_set('offset', 1.0, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output="/dev/null")
  call gadf_fit(1.0e-3, max_iter=6)
  do i = 1, 3
     write(*, '(a, i0, a, es25.17)') 'par ', i, ' = ', fitfuncs(1)%pars(i)%val
  end do
  ok = abs(fitfuncs(1)%pars(1)%val - 3.0_kp) < 1e-9_kp .and. abs(fitfuncs(1)%pars(2)%val - 0.5_kp) < 1e-10_kp .and. &
       & abs(fitfuncs(1)%pars(3)%val - 1.0_kp) < 1e-10_kp
  call gadf_close()
  if (ok) then
     print '(a)', 'PASS'
  else
     print '(a)', 'FAIL'
     error stop 1
  end if
end program fit_narrow_bump
