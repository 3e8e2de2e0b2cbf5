! A branch the sampled recordings miss: N = 400001 points, so gadf_fit records eval() at every 4th abscissa only (gadfit.F90:
! discover); a window `pars(4) < x < pars(5)` holds exactly two points that fall between samples.  No recording contains that path;
! the device meets it in the first pass, reports the points, the Fortran layer records eval() there (on_unseen), the model gains the
! path and the pass is repeated.  The step inside the window is an active parameter: it can only come out right if those two points
! ran the path of their own.  Data generated here and, by the same formula, in tests/golden/make_branching_goldens.py (case
! rare_branch), whose oracle fit with all paths known gives the expected values.
module rare_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: rare_t
   contains
     procedure :: init => r_init
     procedure :: eval => r_eval
  end type rare_t
contains
  subroutine r_init(this)
    class(rare_t), intent(out) :: this
    allocate(this%pars(6))
    call this%set(1, 'amp'); call this%set(2, 'tau'); call this%set(3, 'bgr'); call this%set(4, 'from'); call this%set(5, 'to')
    call this%set(6, 'step')
  end subroutine r_init

  type(advar) function r_eval(this, x) result(y)
    class(rare_t), intent(in) :: this
    real(kp), intent(in) :: x
    y = this%pars(1)*exp(-(x/this%pars(2))) + this%pars(3)
    if (x > this%pars(4)) then
       if (x < this%pars(5)) y = y + this%pars(6)
    end if
  end function r_eval
end module rare_model

program fit_rare_branch
  use rare_model
  use gadfit
  implicit none
  integer, parameter :: n = 400001
  type(rare_t) :: f
  real(kp), allocatable :: x(:), y(:)
  real(kp), parameter :: expected(6) = [4.99999998897668_kp, 20.000000057136472_kp, 0.99999999061032718_kp, &
       & 50.000124999999997_kp, 50.000624999999999_kp, 0.49918566516634122_kp]
  real(kp) :: w0, w1
  integer :: i
  logical :: ok
  allocate(x(n), y(n))
  do i = 1, n
     x(i) = 100.0_kp*real(i - 1, kp)/real(n - 1, kp)
  end do
  w0 = 0.5_kp*(x(200001) + x(200002)); w1 = 0.5_kp*(x(200003) + x(200004))      ! points 200002 and 200003 (1-based) lie inside
  do i = 1, n
     y(i) = 5.0_kp*exp(-(x(i)/20.0_kp)) + 1.0_kp + 1.0e-3_kp*sin(real(mod(37*(i - 1), 1000), kp))
     if (x(i) > w0 .and. x(i) < w1) y(i) = y(i) + 0.5_kp
  end do
  call gadf_init(f)
  call gadf_add_dataset(x, y)
  call gadf_set('amp', 4.5_kp, .true.)
  call gadf_set('tau', 22.0_kp, .true.)
  call gadf_set('bgr', 1.2_kp, .true.)
  call gadf_set('from', w0, .false.)
  call gadf_set('to', w1, .false.)
  call gadf_set('step', 0.1_kp, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output="/dev/null")
  call gadf_fit(1.0, max_iter=6)
  ok = gadf_iterations == 6
  do i = 1, 6
     write(*, '(a, i0, a, es25.17, a, es10.2)') 'par ', i, ' = ', fitfuncs(1)%pars(i)%val, '   rel. dev. ', &
          & abs(fitfuncs(1)%pars(i)%val - expected(i))/abs(expected(i))
     ok = ok .and. abs(fitfuncs(1)%pars(i)%val - expected(i)) <= 1e-9_kp*abs(expected(i))
  end do
  ok = ok .and. abs(fitfuncs(1)%pars(6)%val - 0.5_kp) < 2e-3_kp          ! the step was seen by its two points
  call gadf_close()
  if (ok) then
     print '(a)', 'PASS'
  else
     print '(a)', 'FAIL'
     error stop 1
  end if
end program fit_rare_branch
