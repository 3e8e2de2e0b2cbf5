! A model with plain real(kp) arithmetic on the abscissa inside eval() -- sin(0.05 x)**2 and x**2 are invisible to the recorder and
! reach the device as per-point columns that gadf_fit tabulates over ALL data points (gadfit.F90: tabulate; on several threads
! when eval() is one straight-line path).  usage: bench_real_x [N] [max_iter] [fits]; (fits > 1: further gadf_fit calls from perturbed
! values -- they tabulate the columns again from the layer's own copy of the abscissas); prints the fitted parameters with 17 digits (the test
! compares the threaded tabulation with the serial one bit for bit) and, with GADFIT_HIP_SETUP_TIMES=1, the phases of gadf_fit.
module real_x_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: rx_t
   contains
     procedure :: init => rx_init
     procedure :: eval => rx_eval
  end type rx_t
contains
  subroutine rx_init(this)
    class(rx_t), intent(out) :: this
    allocate(this%pars(5))
    call this%set(1, 'amp'); call this%set(2, 'tau'); call this%set(3, 'osc'); call this%set(4, 'curv'); call this%set(5, 'bgr')
  end subroutine rx_init

  type(advar) function rx_eval(this, x) result(y)
    class(rx_t), intent(in) :: this
    real(kp), intent(in) :: x
    y = this%pars(1)*exp(-(x/this%pars(2))) + this%pars(3)*sin(0.05_kp*x)**2 + this%pars(4)*(1.0e-4_kp*x**2) + this%pars(5)
  end function rx_eval
end module real_x_model

program bench_real_x
  use real_x_model
  use gadfit
  use, intrinsic :: iso_fortran_env, only: int64
  implicit none
  type(rx_t) :: f
  real(kp), allocatable :: x(:), y(:)
  real(kp), parameter :: truth(5) = [5.0_kp, 20.0_kp, 0.7_kp, 1.3_kp, 1.0_kp]
  integer :: n, iters, i, fits, k
  integer(int64) :: c0, c1, rate
  character(len=32) :: arg
  logical :: ok
  n = 200000; iters = 8
  if (command_argument_count() >= 1) then; call get_command_argument(1, arg); read(arg, *) n; end if
  if (command_argument_count() >= 2) then; call get_command_argument(2, arg); read(arg, *) iters; end if
  fits = 1
  if (command_argument_count() >= 3) then; call get_command_argument(3, arg); read(arg, *) fits; end if
  allocate(x(n), y(n))
  do i = 1, n
     x(i) = 100.0_kp*(real(i, kp) - 0.5_kp)/real(n, kp)
     y(i) = truth(1)*exp(-(x(i)/truth(2))) + truth(3)*sin(0.05_kp*x(i))**2 + truth(4)*(1.0e-4_kp*x(i)**2) + truth(5) &
          & + 1.0e-3_kp*sin(real(mod(37*(i - 1), 1000), kp))
  end do
  call gadf_init(f)
  call gadf_add_dataset(x, y)
  call gadf_set('amp', 4.6_kp, .true.); call gadf_set('tau', 22.0_kp, .true.); call gadf_set('osc', 0.6_kp, .true.)
  call gadf_set('curv', 1.5_kp, .true.); call gadf_set('bgr', 1.1_kp, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output="/dev/null")
  call system_clock(c0, rate)
  call gadf_fit(1.0, max_iter=iters)
  call system_clock(c1)
  do k = 2, fits
     call gadf_set('amp', 4.6_kp, .true.); call gadf_set('osc', 0.6_kp, .true.)
     call gadf_fit(1.0, max_iter=iters)
  end do
  write(*, '(a, i0, a, f10.3, a)') 'N = ', n, '   gadf_fit: ', 1e3*real(c1 - c0)/real(rate), ' ms'
  ok = .true.
  do i = 1, 5
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
end program bench_real_x
