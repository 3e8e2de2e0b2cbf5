! integrate() OUTSIDE gadf_fit (gadf_print, a program calling eval() itself): evaluated on the host by module
! numerical_integration through module ad's arithmetic (host_integral).  Prints "name value" lines that
! tests/test_fortran_binding.py holds against closed forms; needs no GPU.
module host_integrands
  use ad
  use gadf_constants
  use numerical_integration
  implicit none
contains
  type(advar) function power_gauss(t, pars) result(y)      ! t**a exp(-b t**2): the integrand of 2_integral_single
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: pars(:)
    y = t**pars(1)*exp(-pars(2)*t**2)
  end function power_gauss
  type(advar) function decay(t, pars) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: pars(:)
    y = exp(-pars(1)*t)
  end function decay
  type(advar) function bell(t, pars) result(y)
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: pars(:)
    y = exp(-pars(1)*t**2)
  end function bell
  type(advar) function outer(t, pars) result(y)            ! int_0^t exp(-p s) ds, itself an integral
    type(advar), intent(in) :: t
    type(advar), intent(in out) :: pars(:)
    y = integrate(decay, pars, 0.0_kp, t)
  end function outer
end module host_integrands

program host_integrate
  use host_integrands
  implicit none
  type(advar) :: p(2), q(1), y, lo, hi
  p(1) = 1.5_kp; p(2) = 0.7_kp
  y = integrate(power_gauss, p, 0.0_kp, 2.0_kp, rel_error=1e-12_kp)
  print '(a, 1x, es24.16)', 'power_gauss_0_2', y%val
  q(1) = 1.3_kp
  y = integrate(decay, q, 0.0_kp, INFINITY)
  print '(a, 1x, es24.16)', 'decay_0_inf', y%val
  y = integrate(decay, q, 1.0_kp, INFINITY)
  print '(a, 1x, es24.16)', 'decay_1_inf', y%val
  y = integrate(bell, q, -INFINITY, INFINITY)
  print '(a, 1x, es24.16)', 'bell_inf_inf', y%val
  y = integrate(bell, q, -INFINITY, 0.5_kp)
  print '(a, 1x, es24.16)', 'bell_inf_half', y%val
  y = integrate(bell, q, INFINITY, 0.5_kp)                  ! reversed: minus the integral from 0.5 to +inf
  print '(a, 1x, es24.16)', 'bell_reversed', y%val
  ! nested: int_0^2 (int_0^t exp(-p s) ds) dt
  call init_integration_dbl(rel_error_inner=1e-13_kp, rel_error_outer=1e-12_kp)
  y = integrate(outer, q, 0.0_kp, 2.0_kp)
  print '(a, 1x, es24.16)', 'nested', y%val
  call free_integration()
  ! forward mode through the integrand: d/dp int_0^1 exp(-p t) dt
  q(1)%d = 1.0_kp; q(1)%index = 1
  y = integrate(decay, q, 0.0_kp, 1.0_kp)
  print '(a, 1x, es24.16)', 'forward_value', y%val
  print '(a, 1x, es24.16)', 'forward_d', y%d
  q(1)%d = 0.0_kp; q(1)%index = 0
  ! an active upper bound, forward mode: d/du int_0^u exp(-p t) dt = exp(-p u)
  hi = 0.8_kp; hi%d = 1.0_kp; hi%index = 1
  y = integrate(decay, q, 0.0_kp, hi)
  print '(a, 1x, es24.16)', 'bound_value', y%val
  print '(a, 1x, es24.16)', 'bound_d', y%d
  lo = 0.2_kp; lo%d = 1.0_kp; lo%index = 1
  y = integrate(decay, q, lo, 0.8_kp)
  print '(a, 1x, es24.16)', 'lower_bound_d', y%d
  ! another rule
  call set_integration_rule(GAUSS_KRONROD_41P)
  y = integrate(power_gauss, p, 0.0_kp, 2.0_kp, rel_error=1e-12_kp)
  print '(a, 1x, es24.16)', 'power_gauss_41', y%val
  print '(a)', 'DONE'
end program host_integrate
