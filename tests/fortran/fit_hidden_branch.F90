! eval() branches on the PLAIN REAL abscissa (if (x < 37.3_kp)): no advar takes part in the comparison, so the recorder
! sees nothing of it -- only that the recorded operations differ from one data point to another.  The reference just runs
! the user's code at every point (gadfit.F90:679-690).  Here the recordings over the data yield two paths that part ways
! without a comparison; the host tabulates, per point, the path it takes (a per-point column the device follows).
! Expected values: the CPU oracle's fit with the breakpoint as a passive parameter at 37.3
! (tests/golden/make_branching_goldens.py, case hidden_branch).
module hidden_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: hidden_t
   contains
     procedure :: init => hd_init
     procedure :: eval => hd_eval
  end type hidden_t
contains
  subroutine hd_init(this)
    class(hidden_t), intent(out) :: this
    allocate(this%pars(3))
    call this%set(1, 'top'); call this%set(2, 'slope'); call this%set(3, 'tau')
  end subroutine hd_init

  type(advar) function hd_eval(this, x) result(y)
    class(hidden_t), intent(in) :: this
    real(kp), intent(in) :: x
    real(kp), parameter :: b = 37.3_kp
    if (x < b) then
       y = this%pars(1) + this%pars(2)*(x - b)
    else
       y = this%pars(1)*exp(-((x - b)/this%pars(3)))
    end if
  end function hd_eval
end module hidden_model

program fit_hidden_branch
  use hidden_model
  use gadfit
  implicit none
  type(hidden_t) :: f
  character(len=512) :: path
  real(kp), parameter :: expected(3) = [3.9976696615404195_kp, 0.079846352633799328_kp, 10.989217253920486_kp]
  integer :: i
  logical :: ok
  call get_command_argument(1, path)
  call gadf_init(f)
  call gadf_add_dataset(trim(path))
  call gadf_set('top', 4.2_kp, .true.)
  call gadf_set('slope', 0.088_kp, .true.)
  call gadf_set('tau', 10.0_kp, .true.)
  call gadf_set_errors(USER)
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
end program fit_hidden_branch
