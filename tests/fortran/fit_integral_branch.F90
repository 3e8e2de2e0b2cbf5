! A quadrature on either side of a fitted breakpoint: int_0^x t**a exp(-b t**2) dt up to the break, the integral up to the break
! plus a line after it.  Two integrate() call sites in two paths through eval() that share one integrand; the second has an ACTIVE
! upper bound (Leibniz term, numerical_integration.F90:413-417).  The comparison is real < advar (automatic_differentiation.F90:380-384).
! Expected values: the CPU oracle's fit of the same data (tests/golden/make_branching_goldens.py, case integral_branch).
module integral_branch_model
  use ad
  use fitfunction
  use gadf_constants
  use numerical_integration
  implicit none
  type, extends(fitfunc) :: ib_t
   contains
     procedure :: init => ib_init
     procedure :: eval => ib_eval
  end type ib_t
contains
  subroutine ib_init(this)
    class(ib_t), intent(out) :: this
    allocate(this%pars(4))
    call this%set(1, 'a'); call this%set(2, 'b'); call this%set(3, 'break'); call this%set(4, 'slope')
  end subroutine ib_init

  type(advar) function ib_eval(this, x) result(y)
    class(ib_t), intent(in) :: this
    real(kp), intent(in) :: x
    type(advar) :: q(2)
    q(1) = this%pars(1); q(2) = this%pars(2)
    if (x < this%pars(3)) then
       y = integrate(kernel, q, 0.0_kp, x)
    else
       y = integrate(kernel, q, 0.0_kp, this%pars(3)) + this%pars(4)*(x - this%pars(3))
    end if
  end function ib_eval

  type(advar) function kernel(t, q) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: q(:)
    y = t**q(1)*exp(-(q(2)*t**2))
  end function kernel
end module integral_branch_model

program fit_integral_branch
  use integral_branch_model
  use gadfit
  implicit none
  type(ib_t) :: f
  character(len=512) :: path
  real(kp), parameter :: expected(4) = [2.0010145850760912_kp, 0.90014469011544207_kp, 1.6015754003059182_kp, &
       & -0.050284599520950053_kp]
  integer :: i
  logical :: ok
  call get_command_argument(1, path)
  call gadf_init(f, rel_error=1e-10_kp)
  call gadf_add_dataset(trim(path))
  call gadf_set('a', 2.2_kp, .true.)
  call gadf_set('b', 0.8_kp, .true.)
  call gadf_set('break', 1.8_kp, .true.)
  call gadf_set('slope', -0.04_kp, .true.)
  call gadf_set_errors(USER)
  call gadf_set_verbosity(output="/dev/null")
  call gadf_fit(1.0, accth=0.9, max_iter=6)
  ok = gadf_iterations == 6
  do i = 1, 4
     write(*, '(a, i0, a, es25.17, a, es10.2)') 'par ', i, ' = ', fitfuncs(1)%pars(i)%val, '   rel. dev. ', &
          & abs(fitfuncs(1)%pars(i)%val - expected(i))/abs(expected(i))
     ok = ok .and. abs(fitfuncs(1)%pars(i)%val - expected(i)) <= 1e-8_kp*abs(expected(i))
  end do
  call gadf_close()
  if (ok) then
     print '(a)', 'PASS'
  else
     print '(a)', 'FAIL'
     error stop 1
  end if
end program fit_integral_branch
