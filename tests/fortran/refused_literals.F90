! What the recorder cannot capture must stop loudly, never fit something else.  eval() may read %val into plain real arithmetic in
! the reference (the components of advar are public, automatic_differentiation.F90:65-80) -- the number then simply does not take
! part in the differentiation.  On the device a real number formed on the host is a literal of the recorded model:
!   argument 1 = 'tval'  an integrand multiplies by cos(t%val): the literal would have to follow the abscissas of the quadrature,
!                        which exist only on the device -> error naming the integration variable
!   argument 1 = 'tfix'  the same over the fixed range [0, 1]: the literal is the same at every data point, only a second recording with
!                        the integration variable elsewhere shows it (gadfit.F90: probe_theta) -> the same error
!   argument 1 = 'ipvx'  an integrand multiplies by cos(q(2)%val*xmod), xmod a module variable that eval() sets to the abscissa: a real
!                        that follows a fitted parameter AND the abscissa inside an integrand (in eval() itself such a real is a per-point
!                        column tabulated anew before every pass since round 5: fit_param_val_x.F90) -> error naming %val
!   argument 1 = 'fdival' an INTEGRAND multiplies by sin(q(2)%val) -- carried as a pseudo-parameter under AD (fit_integrand_param_val.F90)
!                        -- and the program asks for use_ad=.false.: the reference's finite differences move that number with the
!                        parameter (fitfunction.F90:155-174), the device's would not -> error naming use_ad.  (In eval() itself such
!                        reals ride sets of columns under use_ad=.false. since round 5: fit_param_val_x.F90 'fd'.)
!   argument 1 = 'fdacc' eval() multiplies by exp(-pars(2)%val*x), use_ad=.false. AND geodesic acceleration: the central difference
!                        of fitfunction.F90:188-203 would need the columns at p +- h*delta -> error naming accth
!   argument 1 = 'good'  the same two models written with advar arithmetic: fits, prints DONE
module literal_models
  use ad
  use fitfunction
  use gadf_constants
  use numerical_integration
  implicit none
  character(len=8) :: mode = 'good'
  real(kp) :: xmod = 0.0_kp
  type, extends(fitfunc) :: lit_t
   contains
     procedure :: init => lit_init
     procedure :: eval => lit_eval
  end type lit_t
contains
  subroutine lit_init(this)
    class(lit_t), intent(out) :: this
    allocate(this%pars(2))
    call this%set(1, 'amp'); call this%set(2, 'rate')
  end subroutine lit_init

  type(advar) function lit_eval(this, x) result(y)
    class(lit_t), intent(in) :: this
    real(kp), intent(in) :: x
    type(advar) :: q(2)
    q(1) = this%pars(1); q(2) = this%pars(2)
    select case (trim(mode))
    case ('tval')
       y = integrate(weighted_val, q, 0.0_kp, x)
    case ('tfix')
       y = integrate(weighted_val, q, 0.0_kp, 1.0_kp)*x
    case ('ipvx')
       xmod = x
       y = integrate(weighted_pvx, q, 0.0_kp, 1.0_kp)
    case ('fdacc')
       y = this%pars(1)*exp(-this%pars(2)%val*x)
    case ('fdival')
       y = integrate(weighted_pv, q, 0.0_kp, x)
    case default
       y = integrate(weighted, q, 0.0_kp, x) + this%pars(1)*exp(-this%pars(2)*x)
    end select
  end function lit_eval

  type(advar) function weighted(t, q) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: q(:)
    y = q(1)*exp(-q(2)*t)*cos(t)
  end function weighted

  type(advar) function weighted_val(t, q) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: q(:)
    y = q(1)*exp(-q(2)*t)*cos(t%val)
  end function weighted_val

  type(advar) function weighted_pv(t, q) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: q(:)
    y = q(1)*exp(-q(2)*t)*sin(q(2)%val)
  end function weighted_pv

  type(advar) function weighted_pvx(t, q) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: q(:)
    y = q(1)*exp(-q(2)*t)*cos(q(2)%val*xmod)
  end function weighted_pvx
end module literal_models

program refused_literals
  use literal_models
  use gadfit
  implicit none
  type(lit_t) :: f
  integer, parameter :: n = 200
  real(kp), target, save :: xs(n), ys(n)
  integer :: i
  call get_command_argument(1, mode)
  do i = 1, n
     xs(i) = 0.02_kp*i
     ys(i) = 1.0_kp + 0.1_kp*sin(xs(i))
  end do
  call gadf_init(f)
  call gadf_add_dataset(xs, ys)
  call gadf_set('amp', 1.0_kp, .true.)
  call gadf_set('rate', 0.7_kp, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output='/dev/null')
  if (trim(mode) == 'fdacc') then
     call gadf_fit(1.0, max_iter=3, use_ad=.false., accth=0.9)
  else if (trim(mode) == 'fdival') then
     call gadf_fit(1.0, max_iter=3, use_ad=.false.)
  else
     call gadf_fit(1.0, max_iter=3)
  end if
  write(*, '(a)') 'DONE'
  call gadf_close()
end program refused_literals
