! The headline workload (8 skewed Gaussians = 32 active parameters, N points, sigma given) through the Fortran API end to end:
! gadf_init / gadf_add_dataset / gadf_set / gadf_set_errors(USER) / gadf_fit, timing each phase on the host clock.  Shows what the
! Fortran layer adds around the device path (model capture, data hand-over) and the time of an LM iteration as a Fortran user sees
! it.  usage: bench_headline [N] [max_iter]   (defaults 1000000, 10).  Deterministic data (a small LCG + Box-Muller).
module gauss8_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: gauss8_t
   contains
     procedure :: init => g8_init
     procedure :: eval => g8_eval
  end type gauss8_t
contains
  subroutine g8_init(this)
    class(gauss8_t), intent(out) :: this
    allocate(this%pars(32))
  end subroutine g8_init

  type(advar) function g8_eval(this, x) result(y)
    class(gauss8_t), intent(in) :: this
    real(kp), intent(in) :: x
    integer :: k
    y = 0.0_kp
    do k = 0, 7
       y = y + this%pars(4*k+1)*exp(-(((x - this%pars(4*k+2))/this%pars(4*k+3))**2))*(1.0_kp + this%pars(4*k+4)*(x - this%pars(4*k+2)))
    end do
  end function g8_eval
end module gauss8_model

program bench_headline
  use gauss8_model
  use gadfit
  use, intrinsic :: iso_fortran_env, only: int64
  implicit none
  type(gauss8_t) :: f
  real(kp), allocatable, target :: x(:), y(:), s(:)
  real(kp) :: truth(32), t, u1, u2, fx, d
  integer :: n, iters, i, k
  integer(int64) :: c0, c1, c2, c3, c4, rate, seed
  character(len=32) :: arg
  n = 1000000; iters = 10
  if (command_argument_count() >= 1) then; call get_command_argument(1, arg); read(arg, *) n; end if
  if (command_argument_count() >= 2) then; call get_command_argument(2, arg); read(arg, *) iters; end if
  do k = 0, 7
     truth(4*k+1) = 1.0_kp + 4.0_kp*k/7.0_kp; truth(4*k+2) = 6.0_kp + 12.0_kp*k
     truth(4*k+3) = 2.0_kp + 2.0_kp*k/7.0_kp; truth(4*k+4) = 0.01_kp*(1 + mod(k, 3))
  end do
  allocate(x(n), y(n), s(n))
  seed = 20240601_int64
  do i = 1, n
     x(i) = 100.0_kp*(real(i, kp) - 0.5_kp)/real(n, kp)
     fx = 0.0_kp
     do k = 0, 7
        d = x(i) - truth(4*k+2)
        fx = fx + truth(4*k+1)*exp(-((d/truth(4*k+3))**2))*(1.0_kp + truth(4*k+4)*d)
     end do
     s(i) = 0.01_kp*(1.0_kp + abs(fx))
     seed = iand(seed*6364136223846793005_int64 + 1442695040888963407_int64, huge(seed)); u1 = (real(ishft(seed, -11), kp) + 1.0_kp)/4503599627370497.0_kp
     seed = iand(seed*6364136223846793005_int64 + 1442695040888963407_int64, huge(seed)); u2 = real(ishft(seed, -11), kp)/4503599627370496.0_kp
     y(i) = fx + s(i)*sqrt(-2.0_kp*log(u1))*cos(6.283185307179586_kp*u2)
  end do
  call system_clock(c0, rate)
  call gadf_init(f)
  call gadf_add_dataset(x, y, s)
  do k = 1, 32
     t = truth(k)*(1.0_kp + 0.05_kp*(-1)**k)
     call gadf_set(k, t, .true.)
  end do
  call gadf_set_errors(USER)
  call gadf_set_verbosity(output='/dev/null')
  call system_clock(c1)
  call gadf_fit(1.0, max_iter=iters)              ! first call: model capture, kernel load (cache), data hand-over, the iterations
  call system_clock(c2)
  do k = 1, 32
     t = truth(k)*(1.0_kp + 0.05_kp*(-1)**k)
     call gadf_set(k, t, .true.)
  end do
  call system_clock(c3)
  call gadf_fit(1.0, max_iter=iters)
  call system_clock(c4)
  write(*, '(a, i0, a, i0)') 'N = ', n, '  iterations = ', gadf_iterations
  write(*, '(a, f10.3, a)') 'gadf_init + add_dataset + set : ', 1e3*real(c1 - c0, kp)/real(rate, kp), ' ms'
  write(*, '(a, f10.3, a, i0, a)') 'first gadf_fit                : ', 1e3*real(c2 - c1, kp)/real(rate, kp), ' ms  (', iters, ' iterations)'
  write(*, '(a, f10.3, a, f8.4, a)') 'gadf_fit                      : ', 1e3*real(c4 - c3, kp)/real(rate, kp), ' ms = ', &
       & 1e3*real(c4 - c3, kp)/real(rate, kp)/max(1, gadf_iterations), ' ms per LM iteration'
  write(*, '(a, es12.5, a, es12.5)') 'chi2/dof = ', gadf_chi2/real(n - 32, kp), '   A_1 = ', fitfuncs(1)%pars(1)%val
  if (abs(fitfuncs(1)%pars(1)%val - truth(1)) > 0.05_kp*truth(1)) error stop 'fit is off'
  call gadf_close()
  print '(a)', 'DONE'
end program bench_headline
