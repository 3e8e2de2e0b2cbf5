! An eval() that branches on the plain real x (a literal breakpoint: no comparison of AD variables, so nothing for the recorder to
! see) with real(kp) arithmetic on x on one side: the device needs, per data point, which of the two recorded paths it takes and
! the value of the real factor -- two per-point columns that gadf_fit tabulates over ALL points (gadfit.F90: tabulate; on several
! threads while every path is straight-line).  usage: bench_hidden_branch [N] [max_iter]; prints the parameters with 17 digits (the
! test compares the threaded tabulation with the serial one bit for bit).
module hidden_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  real(kp), parameter :: brk = 37.3_kp
  type, extends(fitfunc) :: hb_t
   contains
     procedure :: init => hb_init
     procedure :: eval => hb_eval
  end type hb_t
contains
  subroutine hb_init(this)
    class(hb_t), intent(out) :: this
    allocate(this%pars(3))
    call this%set(1, 'top'); call this%set(2, 'slope'); call this%set(3, 'tau')
  end subroutine hb_init

  type(advar) function hb_eval(this, x) result(y)
    class(hb_t), intent(in) :: this
    real(kp), intent(in) :: x
    real(kp) :: g
    if (x < brk) then
       y = this%pars(1) + this%pars(2)*(x - brk)
    else
       g = 1.0_kp/(1.0_kp + 1.0e-4_kp*(x - brk)**2)
       y = this%pars(1)*exp(-((x - brk)/this%pars(3)))*g
    end if
  end function hb_eval
end module hidden_model

program bench_hidden_branch
  use hidden_model
  use gadfit
  use, intrinsic :: iso_fortran_env, only: int64
  implicit none
  type(hb_t) :: f
  real(kp), allocatable :: x(:), y(:)
  real(kp), parameter :: truth(3) = [4.0_kp, 0.08_kp, 11.0_kp]
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
        y(i) = truth(1)*exp(-((x(i) - brk)/truth(3)))/(1.0_kp + 1.0e-4_kp*(x(i) - brk)**2)
     end if
     y(i) = y(i) + 1.0e-3_kp*sin(real(mod(37*(i - 1), 1000), kp))
  end do
  call gadf_init(f)
  call gadf_add_dataset(x, y)
  call gadf_set('top', 4.3_kp, .true.); call gadf_set('slope', 0.07_kp, .true.); call gadf_set('tau', 10.0_kp, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output="/dev/null")
  call system_clock(c0, rate)
  call gadf_fit(1.0, max_iter=iters)
  call system_clock(c1)
  write(*, '(a, i0, a, f10.3, a)') 'N = ', n, '   gadf_fit: ', 1e3*real(c1 - c0)/real(rate), ' ms'
  ok = .true.
  do i = 1, 3
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
end program bench_hidden_branch
