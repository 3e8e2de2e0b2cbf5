! gadf_init(ws_size=...) is the user's quadrature workspace (fortran/gadfit/gadfit.F90:133-172 -> init_integration,
! numerical_integration.F90:114-135): the number of intervals an adaptive integral may use before the reference stops with
! "Number of iterations was insufficient" (NI:251, 282-283).  The integrand -- six narrow Lorentzians -- needs between 300 and
! 500 intervals at rel_error = 1e-13.  Argument 1: the workspace size (0 = leave it to the default, 1000).  Prints chi2 after one
! LM iteration; with a workspace of 300 or less gadf_fit must stop with the reference's message.
module peaks_model
  use ad
  use fitfunction
  use gadf_constants
  use numerical_integration
  implicit none
  type, extends(fitfunc) :: peaks_t
   contains
     procedure :: init => peaks_init
     procedure :: eval => peaks_eval
  end type peaks_t
contains
  subroutine peaks_init(this)
    class(peaks_t), intent(out) :: this
    allocate(this%pars(2))
    call this%set(1, 'amp'); call this%set(2, 'pos')
  end subroutine peaks_init

  type(advar) function peaks_eval(this, x) result(y)
    class(peaks_t), intent(in) :: this
    real(kp), intent(in) :: x
    type(advar) :: q(2)
    q(1) = this%pars(1); q(2) = this%pars(2)
    y = integrate(kernel, q, 0.0_kp, x)*1.0e-5_kp
  end function peaks_eval

  type(advar) function kernel(t, q) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: q(:)
    integer :: k
    y = q(1)/((t - q(2))**2 + 1.0e-10_kp)
    do k = 1, 5
       y = y + q(1)/((t - (q(2) + 0.13_kp*k))**2 + 1.0e-10_kp)
    end do
  end function kernel
end module peaks_model

program ws_size
  use peaks_model
  use gadfit
  implicit none
  type(peaks_t) :: f
  integer, parameter :: n = 120
  real(kp), target, save :: xs(n), ys(n)
  real(kp), parameter :: base(3) = [0.5_kp, 0.8_kp, 1.0_kp]
  character(len=32) :: arg
  integer :: ws, i
  call get_command_argument(1, arg)
  read(arg, *) ws
  do i = 1, n
     xs(i) = base(mod(i - 1, 3) + 1) + 1.0e-3_kp*(i - 1)
     ys(i) = 1.0_kp
  end do
  if (ws > 0) then
     call gadf_init(f, ws_size=ws, rel_error=1e-13_kp)
  else
     call gadf_init(f, rel_error=1e-13_kp)
  end if
  call gadf_add_dataset(xs, ys)
  call gadf_set('amp', 1.0_kp, .true.)
  call gadf_set('pos', 0.111_kp, .false.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output='/dev/null')
  call gadf_fit(1.0, max_iter=1)
  write(*, '(a, es25.17)') 'chi2 = ', gadf_chi2
  call gadf_close()
  print '(a)', 'DONE'
end program ws_size
