! BASELINE config 4 through the Fortran API: the reference's integral model (fortran/tests/2_integral_single.F90:
! pi * int_0^x t**a exp(-b t**2) dt, Gauss-Kronrod 15, rel_error 1e-10) over N points on (0, 2], both parameters fitted; times
! gadf_init ... gadf_set and the first / a later gadf_fit on the host clock.   usage: bench_integral [N] [max_iter]   (1000000, 6)
module integral_bench_model
  use ad
  use fitfunction
  use gadf_constants
  use numerical_integration
  implicit none
  type, extends(fitfunc) :: integral_t
   contains
     procedure :: init => integral_init
     procedure :: eval => integral_eval
  end type integral_t
contains
  subroutine integral_init(this)
    class(integral_t), intent(out) :: this
    allocate(this%pars(2))
    call this%set(1, 'a'); call this%set(2, 'b')
  end subroutine integral_init

  type(advar) function integral_eval(this, x) result(y)
    class(integral_t), intent(in) :: this
    real(kp), intent(in) :: x
    type(advar) :: q(2)
    q(1) = this%pars(1); q(2) = this%pars(2)
    y = pi*integrate(kernel, q, 0.0_kp, x)
  end function integral_eval

  type(advar) function kernel(t, q) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: q(:)
    y = t**q(1)*exp(-q(2)*t**2)
  end function kernel
end module integral_bench_model

program bench_integral
  use integral_bench_model
  use gadfit
  use, intrinsic :: iso_fortran_env, only: int64
  implicit none
  type(integral_t) :: f
  real(kp), allocatable, target :: x(:), y(:)
  integer :: n, iters, i
  integer(int64) :: c0, c1, c2, c3, c4, rate
  character(len=32) :: arg
  n = 1000000; iters = 6
  if (command_argument_count() >= 1) then; call get_command_argument(1, arg); read(arg, *) n; end if
  if (command_argument_count() >= 2) then; call get_command_argument(2, arg); read(arg, *) iters; end if
  allocate(x(n), y(n))
  do i = 1, n
     x(i) = 2.0_kp*real(i, kp)/real(n, kp)
     ! (a smooth stand-in for the integral with a = 2, b = 1.5: the fit only has to run its iterations)
     y(i) = pi*(x(i)**3/3.0_kp)*exp(-0.9_kp*x(i)**2)*(1.0_kp + 0.3_kp*x(i)**2)
  end do
  call system_clock(c0, rate)
  call gadf_init(f, rel_error=1e-10_kp)
  call gadf_add_dataset(x, y)
  call gadf_set('a', 2.2_kp, .true.); call gadf_set('b', 1.3_kp, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output='/dev/null')
  call system_clock(c1)
  call gadf_fit(1.0, max_iter=iters)
  call system_clock(c2)
  call gadf_set('a', 2.2_kp, .true.); call gadf_set('b', 1.3_kp, .true.)
  call system_clock(c3)
  call gadf_fit(1.0, max_iter=iters)
  call system_clock(c4)
  write(*, '(a, i0, a, i0)') 'N = ', n, '  iterations = ', gadf_iterations
  write(*, '(a, f10.3, a)') 'gadf_init + add_dataset + set : ', 1e3*real(c1 - c0, kp)/real(rate, kp), ' ms'
  write(*, '(a, f10.3, a, i0, a)') 'first gadf_fit                : ', 1e3*real(c2 - c1, kp)/real(rate, kp), ' ms  (', iters, ' iterations)'
  write(*, '(a, f10.3, a, f8.4, a)') 'gadf_fit                      : ', 1e3*real(c4 - c3, kp)/real(rate, kp), ' ms = ', &
       & 1e3*real(c4 - c3, kp)/real(rate, kp)/max(1, gadf_iterations), ' ms per LM iteration'
  write(*, '(a, 2es14.6)') 'a, b = ', fitfuncs(1)%pars(1)%val, fitfuncs(1)%pars(2)%val
  call gadf_close()
  print '(a)', 'DONE'
end program bench_integral
