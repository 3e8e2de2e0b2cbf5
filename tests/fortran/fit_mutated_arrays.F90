! gadf_add_dataset copies its arrays AT THE CALL, whatever their size (round 6: one semantic; rounds 4-5 copied up to 2^20 points and
! borrowed above).  The program adds x, y, overwrites both, and fits: the fit is of what the arrays held when they were added.
! usage: fit_mutated_arrays N   (tests/test_fortran_binding.py runs N = 2^20 - 1 and 2^20 + 1: the same answer on either side)
module mutated_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: mu_t
   contains
     procedure :: init => mu_init
     procedure :: eval => mu_eval
  end type mu_t
contains
  subroutine mu_init(this)
    class(mu_t), intent(out) :: this
    allocate(this%pars(3))
    call this%set(1, 'amp'); call this%set(2, 'tau'); call this%set(3, 'bgr')
  end subroutine mu_init

  type(advar) function mu_eval(this, x) result(y)
    class(mu_t), intent(in) :: this
    real(kp), intent(in) :: x
    y = this%pars(1)*exp(-(x/this%pars(2))) + this%pars(3)
  end function mu_eval
end module mutated_model

program fit_mutated_arrays
  use mutated_model
  use gadfit
  implicit none
  type(mu_t) :: f
  real(kp), allocatable :: x(:), y(:)
  real(kp), parameter :: truth(3) = [4.0_kp, 15.0_kp, 0.5_kp]
  integer :: n, i
  character(len=32) :: arg
  logical :: ok
  n = 1000
  if (command_argument_count() >= 1) then; call get_command_argument(1, arg); read(arg, *) n; end if
  allocate(x(n), y(n))
  do i = 1, n
     x(i) = 60.0_kp*(real(i, kp) - 0.5_kp)/real(n, kp)
     y(i) = truth(1)*exp(-(x(i)/truth(2))) + truth(3) + 1.0e-3_kp*sin(real(mod(37*(i - 1), 1000), kp))
  end do
  call gadf_init(f)
  call gadf_add_dataset(x, y)
  x = -1.0_kp; y = 0.0_kp                  ! (the arrays are the program's again)
  call gadf_set('amp', 3.6_kp, .true.); call gadf_set('tau', 17.0_kp, .true.); call gadf_set('bgr', 0.6_kp, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output="/dev/null")
  call gadf_fit(1.0, max_iter=8)
  ok = .true.
  do i = 1, 3
     write(*, '(a, i0, a, es25.17)') 'par ', i, ' = ', fitfuncs(1)%pars(i)%val
     ok = ok .and. abs(fitfuncs(1)%pars(i)%val - truth(i)) < 1e-3_kp*abs(truth(i))
  end do
  call gadf_close()
  if (ok) then
     print '(a)', 'PASS'
  else
     print '(a)', 'FAIL'
     error stop 1
  end if
end program fit_mutated_arrays
