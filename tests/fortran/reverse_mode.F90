! Host-side reverse mode of module ad (public names of the reference: ad_init_reverse, forward_values,
! index_count, ad_grad, adjoints, trace, ad_memory_report, ad_close; AD:233-313, 1476-1690).
! (1) the sizes ad_init_reverse derives from a memory string -- the known answers of the reference's own test
!     (fortran/tests/ad_reverse_mode.F90:9-21: 4, 108, 108, 1108 trace entries);
! (2) value and gradient of one expression over every elemental with two active operands and a passive one, a
!     repeated evaluation (counters reset by ad_grad) and a different active set.  The Python test evaluates the
!     same expression with the oracle's reverse tape and compares.  No GPU needed.
program reverse_mode_test
  use ad
  use gadf_constants
  implicit none
  type(advar) :: a, b, c, f
  integer :: rep
  call ad_init_reverse('64 B');       write(*, '(a, i0)') 'trace_size ', size(trace); call ad_close()
  call ad_init_reverse('1 kB');       write(*, '(a, i0)') 'trace_size ', size(trace); call ad_close()
  call ad_init_reverse('0.001 MB');   write(*, '(a, i0)') 'trace_size ', size(trace); call ad_close()
  call ad_init_reverse('0.00001 GB'); write(*, '(a, i0)') 'trace_size ', size(trace); call ad_close()
  call ad_init_reverse(sweep_size=1000, trace_size=1000, const_size=100)
  write(*, '(a, 3(i0, 1x), l1)') 'sizes ', size(forward_values), size(trace), size(ad_constants), reverse_mode
  a%val = 1.3_kp; b%val = 2.1_kp; c%val = 0.8_kp
  do rep = 1, 2
     a%index = 1; b%index = 2; c%index = 0
     forward_values(1) = a%val; forward_values(2) = b%val
     index_count = 2
     f = expr(a, b, c)
     if (f%index /= index_count) error stop 'the result is not the last value written'
     call ad_grad(2)
     write(*, '(a, 3es26.17, 3(1x, i0))') 'rev ', f%val, adjoints(1), adjoints(2), index_count, trace_count, const_count
  end do
  ! only b active: slot 1
  a%index = 0; b%index = 1
  forward_values(1) = b%val
  index_count = 1
  f = expr(a, b, c)
  call ad_grad(1)
  write(*, '(a, 2es26.17)') 'revb ', f%val, adjoints(1)
  ! nothing active: a plain evaluation, nothing recorded
  b%index = 0; index_count = 0
  f = expr(a, b, c)
  write(*, '(a, es26.17, 3(1x, i0))') 'pas ', f%val, f%index, trace_count, index_count
  call ad_memory_report()
  call ad_close()
  reverse_mode = .false.
contains
  type(advar) function expr(a, b, c) result(f)
    type(advar), intent(in) :: a, b, c
    f = sin(a*b)/sqrt(b) + exp(-a)*log(b) + a**b + b**3 + 2.0_kp**a + a**1.5_kp + atan(a/b) + tanh(a) + erf(b) &
         & + abs(-a) + cos(a + c) + tan(0.3_kp*a) + asin(a/3.0_kp) + acos(b/3.0_kp) + sinh(a - b) + cosh(b*c) &
         & + asinh(a) + acosh(b + 1.0_kp) + atanh(a/4.0_kp) + (a + 2.0_kp)/(b - 0.5_kp) + 3.0_kp/a - b/2.0_kp + c**a &
         & + 2*a - b*3 + a/c + c/b
  end function expr
end program reverse_mode_test
