! Host-side forward mode of module ad (AD:454-1459, "not reverse_mode" branches): value, first and second
! directional derivative of one expression over every elemental, two active operands and a passive one.
! The Python test evaluates the same expression with the oracle's forward mode and compares.  No GPU needed.
program forward_mode
  use ad
  use gadf_constants
  implicit none
  type(advar) :: a, b, c, f
  a%val = 1.3_kp; a%d = 0.7_kp;  a%dd = 0.2_kp; a%index = 1
  b%val = 2.1_kp; b%d = -0.4_kp; b%dd = 0.1_kp; b%index = 1
  c%val = 0.8_kp
  f = sin(a*b)/sqrt(b) + exp(-a)*log(b) + a**b + b**3 + 2.0_kp**a + a**1.5_kp + atan(a/b) + tanh(a) + erf(b) &
       & + abs(-a) + cos(a + c) + tan(0.3_kp*a) + asin(a/3.0_kp) + acos(b/3.0_kp) + sinh(a - b) + cosh(b*c) &
       & + asinh(a) + acosh(b + 1.0_kp) + atanh(a/4.0_kp) + (a + 2.0_kp)/(b - 0.5_kp) + 3.0_kp/a - b/2.0_kp + c**a
  write(*, '(a, 3es26.17, 1x, i0)') 'fwd ', f%val, f%d, f%dd, f%index
  f = c*2.0_kp + exp(c)            ! passive operands only: no derivative, index 0
  write(*, '(a, 3es26.17, 1x, i0)') 'pas ', f%val, f%d, f%dd, f%index
end program forward_mode
