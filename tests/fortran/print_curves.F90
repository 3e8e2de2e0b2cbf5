! gadf_print without a fit (doc/user_guide.tex:886: "one can ... comment out the call to gadf_fit"): the curves of
! two datasets on a grid, grouped and per dataset, linear and logarithmic spacing.  Runs without a GPU
! (GADFIT_HIP_DEVICE=-1): the curves are evaluated on the host.  Output prefix = argument 1.
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

program print_curves
  use decay_model
  use gadfit
  implicit none
  type(decay_t) :: f
  character(len=512) :: prefix
  call get_command_argument(1, prefix)
  call gadf_init(f, 2)
  call gadf_set(1, 'I0', 5.0_kp, .true.);  call gadf_set(2, 'I0', 7.0_kp, .true.)
  call gadf_set(1, 'bgr', 1.0_kp, .true.); call gadf_set(2, 'bgr', 2.0_kp, .true.)
  call gadf_set('tau', 4.0_kp, .true.)
  call gadf_print(begin=0.0, end=10.0, points=5, output=trim(prefix)//'_a')
  call gadf_print(begin=0.0, end=10.0, points=5, output=trim(prefix)//'_b', grouped=.false.)
  call gadf_print(begin_kp=1.0_kp, end_kp=100.0_kp, points=3, output=trim(prefix)//'_c', logplot=.true.)
  ! the fitfunc class used directly (doc/user_guide.tex:1718): finite-difference helpers against the analytic answers
  block
    type(decay_t) :: g
    real(kp) :: grad(3), d2
    call g%init()
    call g%set(1, 5.0_kp); call g%set('tau', 4.0_kp); call g%set(3, 1.0_kp)
    call g%grad_finite(2.0_kp, [1, 2, 3], grad)
    d2 = g%dir_deriv_2nd_finite(2.0_kp, [1, 2], [0.0_kp, 1.0_kp])
    write(*, '(a, 4es25.16)') 'fd ', grad, d2
    call g%info()
    call g%destroy()
    if (allocated(g%pars)) error stop 'destroy left pars allocated'
  end block
  call gadf_close()
  print '(a)', 'DONE'
end program print_curves
