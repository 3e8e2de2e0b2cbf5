! An eval() that is NOT thread-safe: it parks an intermediate in a module variable (the reference calls eval() from one image at
! a time, so nothing forbids that).  gadf_fit calls eval() from several threads while it tabulates per-point columns; the two
! passes of that tabulation disagree, the layer warns, falls back to serial recordings, and the fit is the serial one to the bit.
! usage: fit_stateful_eval [N [keyword]]; prints the parameters with 17 digits.  'keyword': gadf_init(f, eval_is_thread_safe=.false.,
! force_outcomes=.false.) -- the program says in its source that its eval() must be called from one thread (round 6).
module stateful_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  real(kp) :: parked = 0.0_kp              ! shared by every thread that calls eval()
  type, extends(fitfunc) :: st_t
   contains
     procedure :: init => st_init
     procedure :: eval => st_eval
  end type st_t
contains
  subroutine st_init(this)
    class(st_t), intent(out) :: this
    allocate(this%pars(4))
    call this%set(1, 'amp'); call this%set(2, 'tau'); call this%set(3, 'osc'); call this%set(4, 'bgr')
  end subroutine st_init

  type(advar) function st_eval(this, x) result(y)
    class(st_t), intent(in) :: this
    real(kp), intent(in) :: x
    real(kp) :: acc
    integer :: k
    parked = 0.05_kp*x
    acc = 0.0_kp
    do k = 1, 40                          ! (some work between the write and the read)
       acc = acc + sqrt(real(k, kp) + x)
    end do
    y = this%pars(1)*exp(-(x/this%pars(2))) + this%pars(3)*sin(parked)**2 + this%pars(4) + 0.0_kp*acc
  end function st_eval
end module stateful_model

program fit_stateful_eval
  use stateful_model
  use gadfit
  implicit none
  type(st_t) :: f
  real(kp), allocatable :: x(:), y(:)
  real(kp), parameter :: truth(4) = [5.0_kp, 20.0_kp, 0.7_kp, 1.0_kp]
  integer :: n, i
  character(len=32) :: arg
  logical :: ok
  n = 200000
  if (command_argument_count() >= 1) then; call get_command_argument(1, arg); read(arg, *) n; end if
  allocate(x(n), y(n))
  do i = 1, n
     x(i) = 100.0_kp*(real(i, kp) - 0.5_kp)/real(n, kp)
     y(i) = truth(1)*exp(-(x(i)/truth(2))) + truth(3)*sin(0.05_kp*x(i))**2 + truth(4) + 1.0e-3_kp*sin(real(mod(37*(i - 1), 1000), kp))
  end do
  if (command_argument_count() >= 2) then
     call gadf_init(f, eval_is_thread_safe=.false., force_outcomes=.false.)
  else
     call gadf_init(f)
  end if
  call gadf_add_dataset(x, y)
  call gadf_set('amp', 4.6_kp, .true.); call gadf_set('tau', 22.0_kp, .true.); call gadf_set('osc', 0.6_kp, .true.)
  call gadf_set('bgr', 1.1_kp, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output="/dev/null")
  call gadf_fit(1.0, max_iter=6)
  ok = .true.
  do i = 1, 4
     write(*, '(a, i0, a, es25.17)') 'par ', i, ' = ', fitfuncs(1)%pars(i)%val
     ok = ok .and. abs(fitfuncs(1)%pars(i)%val - truth(i)) < 2e-3_kp*abs(truth(i))
  end do
  call gadf_close()
  if (ok) then
     print '(a)', 'PASS'
  else
     print '(a)', 'FAIL'
     error stop 1
  end if
end program fit_stateful_eval
