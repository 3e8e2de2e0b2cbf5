! An eval() with a MEMO CACHE in module variables -- `if (x /= cached_x) then; cached_x = x; cached_s = f(x); end if` -- correct and
! deterministic under the reference (eval() is called from one image at a time), racy when gadf_fit calls eval() from several
! threads while it tabulates per-point columns: between one thread's test and its read another thread may have replaced the cache.
! Unlike a module variable written at every call the damage is sporadic.  Whatever the interleaving, every fit must either notice
! (the two threaded passes disagree, or the serial re-verification does: warning, serial tabulation) or come out right: all cycles
! print the bits of the serial fit.   usage: fit_racy_cache [N] [cycles]; per cycle the parameters with 17 digits.
module racy_cache_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  real(kp) :: cached_x = -1.0_kp, cached_s = 0.0_kp      ! shared by every thread that calls eval()
  type, extends(fitfunc) :: rc_t
   contains
     procedure :: init => rc_init
     procedure :: eval => rc_eval
  end type rc_t
contains
  subroutine rc_init(this)
    class(rc_t), intent(out) :: this
    allocate(this%pars(4))
    call this%set(1, 'amp'); call this%set(2, 'tau'); call this%set(3, 'osc'); call this%set(4, 'bgr')
  end subroutine rc_init

  type(advar) function rc_eval(this, x) result(y)
    class(rc_t), intent(in) :: this
    real(kp), intent(in) :: x
    if (x /= cached_x) then
       cached_x = x
       cached_s = sin(0.05_kp*x)**2
    end if
    y = this%pars(1)*exp(-(x/this%pars(2))) + this%pars(3)*cached_s + this%pars(4)
  end function rc_eval
end module racy_cache_model

program fit_racy_cache
  use racy_cache_model
  use gadfit
  implicit none
  type(rc_t) :: f
  real(kp), allocatable :: x(:), y(:)
  real(kp), parameter :: truth(4) = [5.0_kp, 20.0_kp, 0.7_kp, 1.0_kp]
  integer :: n, i, cycles, c
  character(len=32) :: arg
  logical :: ok
  n = 20000; cycles = 1
  if (command_argument_count() >= 1) then; call get_command_argument(1, arg); read(arg, *) n; end if
  if (command_argument_count() >= 2) then; call get_command_argument(2, arg); read(arg, *) cycles; end if
  allocate(x(n), y(n))
  do i = 1, n
     x(i) = 100.0_kp*(real(i, kp) - 0.5_kp)/real(n, kp)
     y(i) = truth(1)*exp(-(x(i)/truth(2))) + truth(3)*sin(0.05_kp*x(i))**2 + truth(4) + 1.0e-3_kp*sin(real(mod(37*(i - 1), 1000), kp))
  end do
  ok = .true.
  do c = 1, cycles
     call gadf_init(f)
     call gadf_add_dataset(x, y)
     call gadf_set('amp', 4.6_kp, .true.); call gadf_set('tau', 22.0_kp, .true.); call gadf_set('osc', 0.6_kp, .true.)
     call gadf_set('bgr', 1.1_kp, .true.)
     call gadf_set_errors(NONE)
     call gadf_set_verbosity(output="/dev/null")
     call gadf_fit(1.0, max_iter=6)
     write(*, '(a, i0, 4(1x, es25.17))') 'cycle ', c, (fitfuncs(1)%pars(i)%val, i = 1, 4)
     do i = 1, 4
        ok = ok .and. abs(fitfuncs(1)%pars(i)%val - truth(i)) < 2e-3_kp*abs(truth(i))
     end do
     call gadf_close()
  end do
  if (ok) then
     print '(a)', 'PASS'
  else
     print '(a)', 'FAIL'
     error stop 1
  end if
end program fit_racy_cache
