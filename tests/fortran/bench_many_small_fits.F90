! Many small fits in one process -- a batch of spectra, each fitted on its own: gadf_init ... gadf_fit ... gadf_close per spectrum
! (1000 points, a Gaussian on a background, 4 parameters), what a whole cycle costs after the first.
! usage: bench_many_small_fits [fits] [points]     (defaults 20, 1000)
module small_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: small_t
   contains
     procedure :: init => small_init
     procedure :: eval => small_eval
  end type small_t
contains
  subroutine small_init(this)
    class(small_t), intent(out) :: this
    allocate(this%pars(4))
  end subroutine small_init

  type(advar) function small_eval(this, x) result(y)
    class(small_t), intent(in) :: this
    real(kp), intent(in) :: x
    y = this%pars(1)*exp(-((x - this%pars(2))/this%pars(3))**2) + this%pars(4)
  end function small_eval
end module small_model

program bench_many_small_fits
  use small_model
  use gadfit
  use, intrinsic :: iso_fortran_env, only: int64
  implicit none
  type(small_t) :: f
  real(kp), allocatable, target :: x(:), y(:)
  integer :: nfits, n, i, k
  integer(int64) :: c0, c1, ca, cb, cc, rate
  real(kp) :: first_ms, later_ms, pos, t_init, t_fit, t_close
  character(len=32) :: arg
  nfits = 20; n = 1000
  if (command_argument_count() >= 1) then; call get_command_argument(1, arg); read(arg, *) nfits; end if
  if (command_argument_count() >= 2) then; call get_command_argument(2, arg); read(arg, *) n; end if
  allocate(x(n), y(n))
  later_ms = 0.0_kp; t_init = 0.0_kp; t_fit = 0.0_kp; t_close = 0.0_kp
  do k = 1, nfits
     pos = 4.0_kp + 0.05_kp*k
     do i = 1, n
        x(i) = 10.0_kp*(real(i, kp) - 0.5_kp)/real(n, kp)
        y(i) = 3.0_kp*exp(-((x(i) - pos)/0.8_kp)**2) + 0.5_kp + 1.0e-3_kp*sin(977.0_kp*x(i) + k)
     end do
     call system_clock(c0, rate)
     call gadf_init(f)
     call gadf_add_dataset(x, y)
     call gadf_set(1, 2.5_kp, .true.); call gadf_set(2, 4.3_kp, .true.); call gadf_set(3, 1.0_kp, .true.); call gadf_set(4, 0.3_kp, .true.)
     call gadf_set_errors(NONE)
     call gadf_set_verbosity(output='/dev/null')
     call system_clock(ca)
     call gadf_fit(1.0, max_iter=30)
     call system_clock(cb)
     if (abs(fitfuncs(1)%pars(2)%val - pos) > 1.0e-2_kp) error stop 'fit is off'
     call gadf_close()
     call system_clock(c1)
     if (k > 1) then
        t_init = t_init + 1e3*real(ca - c0, kp)/real(rate, kp); t_fit = t_fit + 1e3*real(cb - ca, kp)/real(rate, kp)
        t_close = t_close + 1e3*real(c1 - cb, kp)/real(rate, kp)
     end if
     if (k == 1) then
        first_ms = 1e3*real(c1 - c0, kp)/real(rate, kp)
     else
        later_ms = later_ms + 1e3*real(c1 - c0, kp)/real(rate, kp)
     end if
  end do
  write(*, '(a, f10.3, a)') 'first cycle (gadf_init ... gadf_close): ', first_ms, ' ms'
  write(*, '(a, f10.3, a, i0, a)') 'later cycles                          : ', later_ms/max(1, nfits - 1), ' ms each (', nfits - 1, ')'
  write(*, '(a, 3f9.3, a)') '  of which gadf_init ... gadf_set, gadf_fit, gadf_close: ', t_init/max(1, nfits - 1), t_fit/max(1, nfits - 1), &
       & t_close/max(1, nfits - 1), ' ms'
  print '(a)', 'DONE'
end program bench_many_small_fits
