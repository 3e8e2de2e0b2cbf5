! BASELINE config 3 through the Fortran API: a GLOBAL fit of 64 curves x 1e5 points, 4 local + 3 global parameters per curve (the
! shape of the reference's example 4, fortran/examples/4_multiple_curves.F90: amplitudes and background per curve, decay times
! shared), timing gadf_init ... gadf_set and the first / a later gadf_fit on the host clock.
! usage: bench_global [curves] [points per curve] [max_iter] [files]      (defaults 64, 100000, 10; a 4th argument "files": the
! curves are first written to text files under /tmp and handed to gadf_add_dataset by path)
module decay3_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: decay3_t
   contains
     procedure :: init => d3_init
     procedure :: eval => d3_eval
  end type decay3_t
contains
  subroutine d3_init(this)
    class(decay3_t), intent(out) :: this
    allocate(this%pars(7))
  end subroutine d3_init

  type(advar) function d3_eval(this, x) result(y)
    class(decay3_t), intent(in) :: this
    real(kp), intent(in) :: x
    y = this%pars(1)*exp(-x/this%pars(5)) + this%pars(2)*exp(-x/this%pars(6)) + this%pars(3)*exp(-x/this%pars(7)) + this%pars(4)
  end function d3_eval
end module decay3_model

program bench_global
  use decay3_model
  use gadfit
  use, intrinsic :: iso_fortran_env, only: int64
  implicit none
  type(decay3_t) :: f
  real(kp), allocatable, target :: x(:,:), y(:,:)
  real(kp), parameter :: tau(3) = [2.0_kp, 9.0_kp, 40.0_kp]
  real(kp) :: amp(4)
  integer :: nc, n, iters, i, c, k
  integer(int64) :: c0, c1, c2, c3, c4, rate
  character(len=32) :: arg
  character(len=64) :: path
  logical :: files
  integer :: u
  nc = 64; n = 100000; iters = 10; files = .false.
  if (command_argument_count() >= 4) then; call get_command_argument(4, arg); files = trim(arg) == 'files'; end if
  if (command_argument_count() >= 1) then; call get_command_argument(1, arg); read(arg, *) nc; end if
  if (command_argument_count() >= 2) then; call get_command_argument(2, arg); read(arg, *) n; end if
  if (command_argument_count() >= 3) then; call get_command_argument(3, arg); read(arg, *) iters; end if
  allocate(x(n, nc), y(n, nc))
  do c = 1, nc
     amp = [5.0_kp + 0.5_kp*sin(real(c, kp)), 3.0_kp + 0.3_kp*cos(real(c, kp)), 1.0_kp + 0.1_kp*sin(2.0_kp*c), 0.2_kp + 0.02_kp*cos(3.0_kp*c)]
     do i = 1, n
        x(i, c) = 100.0_kp*(real(i, kp) - 0.5_kp)/real(n, kp)
        y(i, c) = amp(1)*exp(-x(i, c)/tau(1)) + amp(2)*exp(-x(i, c)/tau(2)) + amp(3)*exp(-x(i, c)/tau(3)) + amp(4) &
             & + 1.0e-3_kp*sin(12345.0_kp*x(i, c) + c)
     end do
  end do
  if (files) then
     do c = 1, nc
        write(path, '(a, i0, a)') '/tmp/gadfit_bench_global_', c, '.txt'
        open(newunit=u, file=trim(path), action='write', status='replace')
        write(u, '(a)') '# x y'
        do i = 1, n
           write(u, '(2es25.17)') x(i, c), y(i, c)
        end do
        close(u)
     end do
  end if
  call system_clock(c0, rate)
  call gadf_init(f, nc)
  do c = 1, nc
     if (files) then
        write(path, '(a, i0, a)') '/tmp/gadfit_bench_global_', c, '.txt'
        call gadf_add_dataset(trim(path))
     else
        call gadf_add_dataset(x(:, c), y(:, c))
     end if
  end do
  call set_start()
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output='/dev/null')
  call system_clock(c1)
  call gadf_fit(1.0, max_iter=iters)
  call system_clock(c2)
  call set_start()
  call system_clock(c3)
  call gadf_fit(1.0, max_iter=iters)
  call system_clock(c4)
  write(*, '(a, i0, a, i0, a, i0)') 'curves = ', nc, '  points per curve = ', n, '  iterations = ', gadf_iterations
  write(*, '(a, f10.3, a)') 'gadf_init + add_dataset + set : ', 1e3*real(c1 - c0, kp)/real(rate, kp), ' ms'
  write(*, '(a, f10.3, a, i0, a)') 'first gadf_fit                : ', 1e3*real(c2 - c1, kp)/real(rate, kp), ' ms  (', iters, ' iterations)'
  write(*, '(a, f10.3, a, f8.4, a)') 'gadf_fit                      : ', 1e3*real(c4 - c3, kp)/real(rate, kp), ' ms = ', &
       & 1e3*real(c4 - c3, kp)/real(rate, kp)/max(1, gadf_iterations), ' ms per LM iteration'
  write(*, '(a, 3es14.6)') 'tau = ', fitfuncs(1)%pars(5)%val, fitfuncs(1)%pars(6)%val, fitfuncs(1)%pars(7)%val
  if (any(abs([(fitfuncs(1)%pars(4 + k)%val, k = 1, 3)] - tau) > 0.02_kp*tau)) error stop 'fit is off'
  call gadf_close()
  if (files) then
     do c = 1, nc
        write(path, '(a, i0, a)') '/tmp/gadfit_bench_global_', c, '.txt'
        open(newunit=u, file=trim(path), status='old'); close(u, status='delete')
     end do
  end if
  print '(a)', 'DONE'
contains
  subroutine set_start()
    integer :: cc
    do cc = 1, nc
       call gadf_set(cc, 1, 5.0_kp, .true.); call gadf_set(cc, 2, 3.0_kp, .true.)
       call gadf_set(cc, 3, 1.0_kp, .true.); call gadf_set(cc, 4, 0.3_kp, .true.)
    end do
    call gadf_set(5, 2.2_kp, .true.); call gadf_set(6, 8.0_kp, .true.); call gadf_set(7, 44.0_kp, .true.)
  end subroutine set_start
end program bench_global
