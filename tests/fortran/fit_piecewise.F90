! A piecewise fitting function: eval() branches on a comparison of the abscissa with a PARAMETER (real < advar,
! automatic_differentiation.F90:380-384), and the breakpoint is fitted.  The reference evaluates eval() afresh at every point
! (gadfit.F90:679-690), so each point takes its own branch at the current parameters; here every path through eval() is
! recorded as one variant tape and the device walks their decision tree per point (gfh_set_model_variants).  The second
! segment also carries a plain real(kp) function of x -- an auxiliary per-point column, tabulated for the points of BOTH
! segments, since points change segment while the breakpoint moves.
! Expected values: the CPU oracle's fit of the same data (tests/golden/make_branching_goldens.py, case piecewise_aux).
module piecewise_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: piecewise_t
   contains
     procedure :: init => pw_init
     procedure :: eval => pw_eval
  end type piecewise_t
contains
  subroutine pw_init(this)
    class(piecewise_t), intent(out) :: this
    allocate(this%pars(4))
    call this%set(1, 'top'); call this%set(2, 'break'); call this%set(3, 'slope'); call this%set(4, 'tau')
  end subroutine pw_init

  type(advar) function pw_eval(this, x) result(y)
    class(piecewise_t), intent(in) :: this
    real(kp), intent(in) :: x
    real(kp) :: g
    if (x < this%pars(2)) then
       y = this%pars(1) + this%pars(3)*(x - this%pars(2))
    else
       g = 1.0_kp/(1.0_kp + 1.0e-4_kp*x**2)            ! real arithmetic: invisible to the recorder
       y = this%pars(1)*exp(-((x - this%pars(2))/this%pars(4)))*(g/(1.0_kp/(1.0_kp + 1.0e-4_kp*this%pars(2)**2)))
    end if
  end function pw_eval
end module piecewise_model

program fit_piecewise
  use piecewise_model
  use gadfit
  implicit none
  type(piecewise_t) :: f
  character(len=512) :: path
  real(kp), parameter :: expected(4) = [3.9943098773016765_kp, 37.307873635126562_kp, 0.079713115860392633_kp, &
       & 10.998485263047687_kp]
  integer :: i
  logical :: ok
  call get_command_argument(1, path)
  call gadf_init(f)
  call gadf_add_dataset(trim(path))
  call gadf_set('top', 4.2_kp, .true.)
  call gadf_set('break', 34.0_kp, .true.)
  call gadf_set('slope', 0.088_kp, .true.)
  call gadf_set('tau', 10.0_kp, .true.)
  call gadf_set_errors(USER)
  call gadf_set_verbosity(output="/dev/null")
  call gadf_fit(1.0, accth=0.9, max_iter=6)
  ok = gadf_iterations == 6
  do i = 1, 4
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
end program fit_piecewise
