! GPU parity test through the Fortran API: the C++ solver's robust cost functions
! (c++/tests/lm_solver.cpp:499-565 "Loss functions") via gadf_set_loss.  Start values and the
! expected results come from the caller (tests/golden/goldens.py CXX_LOSS), the two decay curves
! from the same two-column files as fit_two_curves.
!   fit_robust_loss curve1 curve2 loss iterations I0_1 bgr_1 I0_2 bgr_2 tau
module decay_model_loss
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
end module decay_model_loss

program fit_robust_loss
  use decay_model_loss
  use gadfit
  implicit none
  type(decay_t) :: f
  character(len=512) :: p1, p2, arg
  real(kp) :: v(5)
  integer :: i, j, loss, iters
  call get_command_argument(1, p1)
  call get_command_argument(2, p2)
  call get_command_argument(3, arg); read(arg, *) loss
  call get_command_argument(4, arg); read(arg, *) iters
  do i = 1, 5
     call get_command_argument(4 + i, arg); read(arg, *) v(i)
  end do
  call gadf_init(f, 2)
  call gadf_add_dataset(trim(p1))
  call gadf_add_dataset(trim(p2))
  call gadf_set(1, 'I0', v(1), .true.)
  call gadf_set(1, 'bgr', v(2), .true.)
  call gadf_set(2, 'I0', v(3), .true.)
  call gadf_set(2, 'bgr', v(4), .true.)
  call gadf_set('tau', v(5), .true.)
  call gadf_set_verbosity(output='/dev/null')
  call gadf_set_loss(loss)
  call gadf_fit(lambda=1.0, lam_incs=3, max_iter=iters)
  do i = 1, 2
     do j = 1, 3
        write(*, '(a, 2(i0, 1x), es25.17)') 'PAR ', i, j, fitfuncs(i)%pars(j)%val
     end do
  end do
  call gadf_close()
  print '(a)', 'DONE'
end program fit_robust_loss
