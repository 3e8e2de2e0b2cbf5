! An eval() that branches on a comparison of the abscissa with a FITTED parameter AND carries plain real(kp) arithmetic on x on one
! side (a per-point column): while the breakpoint moves, points change sides, so the column is tabulated for the points of BOTH paths
! -- each point recorded along its own path and, with the comparison forced, along the other (gadfit.F90: tabulate; on threads, the
! forced outcomes in thread-local storage).  usage: bench_guard_aux [N] [max_iter]; prints the parameters with 17 digits (the test
! compares the threaded tabulation with the serial one bit for bit).
module guard_aux_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  real(kp), parameter :: brk = 37.3_kp
  type, extends(fitfunc) :: ga_t
   contains
     procedure :: init => ga_init
     procedure :: eval => ga_eval
  end type ga_t
contains
  subroutine ga_init(this)
    class(ga_t), intent(out) :: this
    allocate(this%pars(4))
    call this%set(1, 'top'); call this%set(2, 'slope'); call this%set(3, 'tau'); call this%set(4, 'break')
  end subroutine ga_init

  type(advar) function ga_eval(this, x) result(y)
    class(ga_t), intent(in) :: this
    real(kp), intent(in) :: x
    real(kp) :: g
    if (x < this%pars(4)) then
       y = this%pars(1) + this%pars(2)*(x - this%pars(4))
    else
       g = 1.0_kp/(1.0_kp + 1.0e-4_kp*x**2)
       y = this%pars(1)*exp(-((x - this%pars(4))/this%pars(3)))*(g*(1.0_kp + 1.0e-4_kp*brk**2))
    end if
  end function ga_eval
end module guard_aux_model

program bench_guard_aux
  use guard_aux_model
  use gadfit
  use, intrinsic :: iso_fortran_env, only: int64
  implicit none
  type(ga_t) :: f
  real(kp), allocatable :: x(:), y(:)
  real(kp), parameter :: truth(4) = [4.0_kp, 0.08_kp, 11.0_kp, 37.3_kp]
  integer :: n, iters, i
  integer(int64) :: c0, c1, rate
  character(len=32) :: arg
  logical :: ok
  n = 100000; iters = 6
  if (command_argument_count() >= 1) then; call get_command_argument(1, arg); read(arg, *) n; end if
  if (command_argument_count() >= 2) then; call get_command_argument(2, arg); read(arg, *) iters; end if
  allocate(x(n), y(n))
  do i = 1, n
     x(i) = 100.0_kp*(real(i, kp) - 0.5_kp)/real(n, kp)
     if (x(i) < brk) then
        y(i) = truth(1) + truth(2)*(x(i) - brk)
     else
        y(i) = truth(1)*exp(-((x(i) - brk)/truth(3)))*(1.0_kp + 1.0e-4_kp*brk**2)/(1.0_kp + 1.0e-4_kp*x(i)**2)
     end if
     y(i) = y(i) + 1.0e-3_kp*sin(real(mod(37*(i - 1), 1000), kp))
  end do
  call gadf_init(f)
  call gadf_add_dataset(x, y)
  call gadf_set('top', 4.3_kp, .true.); call gadf_set('slope', 0.07_kp, .true.); call gadf_set('tau', 10.0_kp, .true.)
  call gadf_set('break', 39.0_kp, .true.)      ! (above the true break: points between them start on the first path and come to need the second path's column)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output="/dev/null")
  call system_clock(c0, rate)
  call gadf_fit(1.0, max_iter=iters)
  call system_clock(c1)
  write(*, '(a, i0, a, f10.3, a)') 'N = ', n, '   gadf_fit: ', 1e3*real(c1 - c0)/real(rate), ' ms'
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
end program bench_guard_aux
