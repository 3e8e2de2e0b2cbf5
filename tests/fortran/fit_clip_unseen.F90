! A clipped ramp: two comparisons of AD variables in a row (advar < real, advar > advar; automatic_differentiation.F90:
! 315-318, 361-365).  With the start values no data point reaches the upper clip, so the recordings over the data contain no
! such path; the fit steepens the ramp, the device meets the clip INSIDE gadf_fit, reports the points, the Fortran layer
! records eval() there (gadfit.F90: on_unseen), the model gains the path and the pass is repeated.  The clip level is passive.
! Also: the fourteen comparisons of the reference's own test (fortran/tests/ad_forward_mode.F90:9-25) on the host.
! Expected values: the CPU oracle's fit with all paths known (tests/golden/make_branching_goldens.py, case clip_unseen).
module clip_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: clip_t
   contains
     procedure :: init => cl_init
     procedure :: eval => cl_eval
  end type clip_t
contains
  subroutine cl_init(this)
    class(clip_t), intent(out) :: this
    allocate(this%pars(4))
    call this%set(1, 'slope'); call this%set(2, 'foot'); call this%set(3, 'level'); call this%set(4, 'offset')
  end subroutine cl_init

  type(advar) function cl_eval(this, x) result(y)
    class(clip_t), intent(in) :: this
    real(kp), intent(in) :: x
    type(advar) :: t
    t = this%pars(1)*(x - this%pars(2))
    if (t < 0.0_kp) t = 0.0_kp*t
    if (t > this%pars(3)) t = this%pars(3) + 0.0_kp*t
    y = t + this%pars(4)
  end function cl_eval
end module clip_model

program fit_clip_unseen
  use clip_model
  use gadfit
  implicit none
  type(clip_t) :: f
  character(len=512) :: path
  real(kp), parameter :: expected(4) = [0.19990154485721767_kp, 20.013866855118092_kp, 6.0_kp, 1.000541361632848_kp]
  ! fortran/tests/testing.F90:23-26 (fix_d(1:3))
  real(kp), parameter :: fix_d(3) = [6.1360420701563498d0, 2.9606444748278875d0, 9.9253736972586246d0]
  type(advar) :: a, b
  integer :: i
  logical :: ok
  ! ad_forward_mode.F90:9-25: comparisons look at val only, whatever the type of the other operand
  a%val = fix_d(1); b%val = fix_d(2)
  ok = (a > b) .and. (b < a) .and. (a > real(fix_d(2), real32)) .and. (real(fix_d(2), real32) < a) .and. &
       & (a < real(fix_d(3), real32)) .and. (real(fix_d(3), real32) > a) .and. (a > fix_d(2)) .and. (fix_d(2) < a) .and. &
       & (a < fix_d(3)) .and. (fix_d(3) > a) .and. (a > real(fix_d(2), qp)) .and. (real(fix_d(2), qp) < a) .and. &
       & (a < real(fix_d(3), qp)) .and. (real(fix_d(3), qp) > a)
  if (.not. ok) then
     print '(a)', 'FAIL (comparisons)'
     error stop 1
  end if
  call get_command_argument(1, path)
  call gadf_init(f)
  call gadf_add_dataset(trim(path))
  call gadf_set('slope', 0.07_kp, .true.)
  call gadf_set('foot', 17.0_kp, .true.)
  call gadf_set('level', 6.0_kp, .false.)
  call gadf_set('offset', 1.3_kp, .true.)
  call gadf_set_errors(USER)
  call gadf_set_verbosity(output="/dev/null")
  call gadf_fit(1.0, max_iter=6)
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
end program fit_clip_unseen
