! A branch on the PLAIN REAL abscissa hidden BEHIND a comparison of AD variables:
!     if (x < brk) then                   ! brk = pars(4): fitted -- the device decides this per point at the current parameters
!        if (x < 20.0_kp) then  A(x)      ! plain real: invisible to operator overloading
!        else                   B(x)
!     else                      C(x)
! With the start value of brk no data point with x >= 20 lies on the first side, so the data alone only ever show A behind
! "x < brk"; the fit moves brk beyond 20 and points enter B.  The reference runs eval() afresh at every point (gadfit.F90:679-690)
! and takes B there.  The capture forces the recorded outcomes on eval() at every data point (gadfit.F90: cross_check), meets B,
! and the per-point variant column tells A from B.  (Before round 4 the device sent those points into A, silently.)
! usage: fit_fork_behind_guard [N]; prints the parameters with 17 digits; tests/test_fortran_binding.py compares them with the fit
! of the same model through the Python API, where both comparisons are recorded.
module fork_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: fork_t
   contains
     procedure :: init => fk_init
     procedure :: eval => fk_eval
  end type fork_t
contains
  subroutine fk_init(this)
    class(fork_t), intent(out) :: this
    allocate(this%pars(4))
    call this%set(1, 'top'); call this%set(2, 'slope'); call this%set(3, 'tau'); call this%set(4, 'break')
  end subroutine fk_init

  type(advar) function fk_eval(this, x) result(y)
    class(fork_t), intent(in) :: this
    real(kp), intent(in) :: x
    if (x < this%pars(4)) then
       if (x < 20.0_kp) then
          y = this%pars(1) + this%pars(2)*(x - 20.0_kp)
       else
          y = this%pars(1) + 3.0_kp*this%pars(2)*(x - 20.0_kp)
       end if
    else
       y = (this%pars(1) + 3.0_kp*this%pars(2)*(this%pars(4) - 20.0_kp))*exp(-((x - this%pars(4))/this%pars(3)))
    end if
  end function fk_eval
end module fork_model

program fit_fork_behind_guard
  use fork_model
  use gadfit
  implicit none
  type(fork_t) :: f
  integer :: n, i
  character(len=32) :: arg
  real(kp), allocatable, target :: xs(:), ys(:)
  real(kp) :: x, t
  n = 2000
  if (command_argument_count() >= 1) then
     call get_command_argument(1, arg); read(arg, *) n
  end if
  allocate(xs(n), ys(n))
  do i = 1, n                              ! the data of top = 4, slope = 0.08, tau = 11, break = 27.3, a ripple on top
     x = 60.0_kp*(real(i, kp) - 0.5_kp)/real(n, kp)
     xs(i) = x
     if (x < 27.3_kp) then
        t = 4.0_kp + merge(0.08_kp, 0.24_kp, x < 20.0_kp)*(x - 20.0_kp)
     else
        t = (4.0_kp + 0.24_kp*7.3_kp)*exp(-(x - 27.3_kp)/11.0_kp)
     end if
     ys(i) = t*(1.0_kp + 0.01_kp*sin(12.9898_kp*real(i, kp)))
  end do
  call gadf_init(f)
  call gadf_add_dataset(xs, ys)
  call gadf_set('top', 4.2_kp, .true.)
  call gadf_set('slope', 0.07_kp, .true.)
  call gadf_set('tau', 10.0_kp, .true.)
  call gadf_set('break', 17.0_kp, .true.)          ! (below 20: at the capture no point with x >= 20 is on the first side)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output='/dev/null')
  call gadf_fit(1.0, max_iter=12)
  do i = 1, 4
     write(*, '(a, i0, a, es25.17)') 'par ', i, ' = ', fitfuncs(1)%pars(i)%val
  end do
  write(*, '(a, i0)') 'iterations = ', gadf_iterations
  write(*, '(a, es25.17)') 'chi2 = ', gadf_chi2
  call gadf_close()
  print '(a)', 'DONE'
end program fit_fork_behind_guard
