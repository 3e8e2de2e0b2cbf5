! An integrand that compares AD variables: the function handed to integrate() has a kink at a FITTED position (t > q(2):
! advar > advar inside the integrand, automatic_differentiation.F90:315-318).  The reference's integrand takes the branch anew at
! every abscissa the quadrature calls it at; here gadf_fit records the integrand with its integration variable at a dozen places
! of its range, every path through it is a recording of its own, the library pools them into the one call site and the device
! picks the recording per evaluation (gadfit.F90: discover, ad_theta; libgadfit_hip: Model::alts, emit_family).
! Expected values: the CPU oracle's fit of the same data (tests/golden/make_branching_goldens.py, case kinked_integrand).
module kinked_model
  use ad
  use fitfunction
  use gadf_constants
  use numerical_integration
  implicit none
  type, extends(fitfunc) :: kinked_t
   contains
     procedure :: init => k_init
     procedure :: eval => k_eval
  end type kinked_t
contains
  subroutine k_init(this)
    class(kinked_t), intent(out) :: this
    allocate(this%pars(4))
    call this%set(1, 'amp'); call this%set(2, 'kink'); call this%set(3, 'tau'); call this%set(4, 'bgr')
  end subroutine k_init

  type(advar) function k_eval(this, x) result(y)
    class(kinked_t), intent(in) :: this
    real(kp), intent(in) :: x
    type(advar) :: q(3)
    q(1) = this%pars(1); q(2) = this%pars(2); q(3) = this%pars(3)
    y = integrate(kernel, q, 0.0_kp, x) + this%pars(4)
  end function k_eval

  type(advar) function kernel(t, q) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: q(:)
    if (t > q(2)) then
       y = q(1)*exp(-((t - q(2))/q(3)))
    else
       y = q(1)*(1.0_kp + 0.5_kp*(t - q(2)))
    end if
  end function kernel
end module kinked_model

program fit_kinked_integrand
  use kinked_model
  use gadfit
  implicit none
  type(kinked_t) :: f
  character(len=512) :: path
  real(kp), parameter :: expected(4) = [1.3003170667725117_kp, 1.1991875769513196_kp, &
       & 0.79987099713442222_kp, 0.099906800259076514_kp]
  ! second argument "far": the kink starts beyond every range of integration -- one path through the integrand at the start, the
  ! other first met on the device inside the fit (the layer records the integrands again at the parameters of that pass)
  real(kp), parameter :: expected_far(4) = [1.2976213824316831_kp, 1.1977326017725713_kp, &
       & 0.80000000000000004_kp, 0.10248468801110946_kp]
  real(kp) :: want(4)
  character(len=16) :: mode
  integer :: i, want_iterations
  logical :: ok
  call get_command_argument(1, path)
  mode = ''
  if (command_argument_count() >= 2) call get_command_argument(2, mode)
  call gadf_init(f, rel_error=1e-10_kp)
  call gadf_add_dataset(trim(path))
  if (trim(mode) == 'far') then
     call gadf_set('amp', 1.326_kp, .true.)
     call gadf_set('kink', 4.32_kp, .true.)
     call gadf_set('tau', 0.8_kp, .false.)
     call gadf_set('bgr', 0.09000000000000001_kp, .true.)
     call gadf_set_errors(USER)
     call gadf_set_verbosity(output="/dev/null")
     call gadf_fit(1.0, max_iter=6)
     want = expected_far; want_iterations = 6
  else
  call gadf_set('amp', 1.3650000000000002_kp, .true.)
  call gadf_set('kink', 1.116_kp, .true.)
  call gadf_set('tau', 0.8480000000000001_kp, .true.)
  call gadf_set('bgr', 0.08000000000000002_kp, .true.)
  call gadf_set_errors(USER)
  call gadf_set_verbosity(output="/dev/null")
  call gadf_fit(1.0, accth=0.9, max_iter=6)
  want = expected; want_iterations = 5
  end if
  ok = gadf_iterations == want_iterations
  do i = 1, 4
     write(*, '(a, i0, a, es25.17, a, es10.2)') 'par ', i, ' = ', fitfuncs(1)%pars(i)%val, '   rel. dev. ', &
          & abs(fitfuncs(1)%pars(i)%val - want(i))/abs(want(i))
     ok = ok .and. abs(fitfuncs(1)%pars(i)%val - want(i)) <= 1e-8_kp*abs(want(i))
  end do
  call gadf_close()
  if (ok) then
     print '(a)', 'PASS'
  else
     print '(a)', 'FAIL'
     error stop 1
  end if
end program fit_kinked_integrand
