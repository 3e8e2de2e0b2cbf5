! GPU parity test through the Fortran API: global fit of two decay curves sharing tau
! (the reference's 4_multiple_curves known answers, fortran/tests/4_multiple_curves.F90:56-61),
! data read from two-column text files like the reference's example program.
module decay_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: decay_t
   contains
     procedure :: init => decay_init
     procedure :: eval => decay_eval
  end type decay_t
contains
  subroutine decay_init(this)
    class(decay_t), intent(out) :: this
    allocate(this%pars(3))
    call this%set(1, 'I0'); call this%set(2, 'tau'); call this%set(3, 'bgr')
  end subroutine decay_init

  type(advar) function decay_eval(this, x) result(y)
    class(decay_t), intent(in) :: this
    real(kp), intent(in) :: x
    y = this%pars(1)*exp(-x/this%pars(2)) + this%pars(3)
  end function decay_eval
end module decay_model

program fit_two_curves
  use decay_model
  use gadfit
  implicit none
  type(decay_t) :: f
  character(len=512) :: p1, p2
  real(kp), parameter :: golden(3,2) = reshape([ &
       & 46.980695087179093_kp, 21.367028663570494_kp, 8.9528433588272360_kp, &
       & 150.03361724451275_kp, 21.367028663570494_kp, 4.3777353718042322_kp], [3, 2])
  integer :: i, j
  call get_command_argument(1, p1)
  call get_command_argument(2, p2)
  call gadf_init(f, 2)
  call gadf_add_dataset(trim(p1))
  call gadf_add_dataset(trim(p2))
  call gadf_set(1, 'I0', 1.0, .true.)
  call gadf_set(2, 'I0', 1.0, .true.)
  call gadf_set(1, 3, 1.0, .true.)
  call gadf_set(2, 3, 1.0, .true.)
  call gadf_set('tau', 1.0, .true.)
  call gadf_set_errors(SQRT_Y)
  call gadf_set_verbosity(output='/dev/null')
  call gadf_fit(lambda=10.0, accth=0.9, max_iter=4)
  do i = 1, 2
     do j = 1, 3
        write(*, '(2(i0, 1x), es25.17)') i, j, fitfuncs(i)%pars(j)%val
        if (abs(fitfuncs(i)%pars(j)%val - golden(j, i)) > 1e-10_kp*abs(golden(j, i))) &
             & error stop 'parameter differs from the reference golden value'
     end do
  end do
  call gadf_close()
  print '(a)', 'PASS'
end program fit_two_curves
