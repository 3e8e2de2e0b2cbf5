! A feature of eval() that only a recorder which visits EVERY abscissa can find: N = 400001 points, a window three points wide whose
! bounds are plain reals of the module -- `if (x > w_from .and. x < w_to)` is invisible to operator overloading (no comparison of an
! AD variable for the device to decide), and none of the 2^17 evenly spaced abscissas of a sampled capture falls inside.  The
! reference evaluates eval() afresh at every point (gadfit.F90:679-690) and fits the step inside the window; gadf_fit records eval()
! at every abscissa by default (GADFIT_HIP_VERIFY=sample: the 2^17-point sample of rounds 1-3 -- the captured model then has no
! node for the step at all, its column of the Jacobian is zero and the fit stops in the Cholesky factorization; with the step
! passive it would be frozen silently).  Expected values: the oracle's fit (tests/golden/make_branching_goldens.py, case
! narrow_window), same data by the same formula.
module window_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  real(kp) :: w_from = 0.0_kp, w_to = 0.0_kp        ! plain reals: set by the program before the fit
  type, extends(fitfunc) :: win_t
   contains
     procedure :: init => w_init
     procedure :: eval => w_eval
  end type win_t
contains
  subroutine w_init(this)
    class(win_t), intent(out) :: this
    allocate(this%pars(4))
    call this%set(1, 'amp'); call this%set(2, 'tau'); call this%set(3, 'bgr'); call this%set(4, 'step')
  end subroutine w_init

  type(advar) function w_eval(this, x) result(y)
    class(win_t), intent(in) :: this
    real(kp), intent(in) :: x
    y = this%pars(1)*exp(-(x/this%pars(2))) + this%pars(3)
    if (x > w_from .and. x < w_to) y = y + this%pars(4)
  end function w_eval
end module window_model

program fit_narrow_window
  use window_model
  use gadfit
  implicit none
  integer, parameter :: n = 400001
  type(win_t) :: f
  real(kp), allocatable :: x(:), y(:)
  real(kp), parameter :: expected(4) = [4.999999983387668_kp, 20.000000086288651_kp, 0.99999999248330085_kp, 0.49916892583932754_kp]
  integer :: i
  logical :: ok
  allocate(x(n), y(n))
  do i = 1, n
     x(i) = 100.0_kp*real(i - 1, kp)/real(n - 1, kp)
  end do
  w_from = 0.5_kp*(x(200001) + x(200002)); w_to = 0.5_kp*(x(200004) + x(200005))      ! points 200002 .. 200004 (1-based) lie inside
  do i = 1, n
     y(i) = 5.0_kp*exp(-(x(i)/20.0_kp)) + 1.0_kp + 1.0e-3_kp*sin(real(mod(37*(i - 1), 1000), kp))
     if (x(i) > w_from .and. x(i) < w_to) y(i) = y(i) + 0.5_kp
  end do
  ! (argument 'sample': gadf_init(f, record_every_abscissa=.false.) -- the sampled capture asked for in the program's source, round 6;
  ! this program is the one it is wrong for: the window lies between two samples)
  if (command_argument_count() >= 1) then
     call gadf_init(f, record_every_abscissa=.false.)
  else
     call gadf_init(f)
  end if
  call gadf_add_dataset(x, y)
  call gadf_set('amp', 4.5_kp, .true.)
  call gadf_set('tau', 22.0_kp, .true.)
  call gadf_set('bgr', 1.2_kp, .true.)
  call gadf_set('step', 0.1_kp, .true.)
  call gadf_set_errors(NONE)
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
end program fit_narrow_window
