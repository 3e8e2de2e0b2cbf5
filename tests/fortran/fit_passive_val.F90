! eval() reads the %val of a PASSIVE parameter into plain arithmetic: an integer exponent n = nint(pars(4)%val) of a polynomial
! term.  For the reference that is ordinary Fortran; here it is a constant of the captured model -- legitimate, because a passive
! parameter keeps its value for the whole fit -- and gadf_fit captures the model again when such a value (or the active set) has
! changed between fits: the second fit below runs with another exponent and must find the other data's parameters.
module passive_val_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: pv_t
   contains
     procedure :: init => pv_init
     procedure :: eval => pv_eval
  end type pv_t
contains
  subroutine pv_init(this)
    class(pv_t), intent(out) :: this
    allocate(this%pars(4))
    call this%set(1, 'amp'); call this%set(2, 'tau'); call this%set(3, 'coef'); call this%set(4, 'order')
  end subroutine pv_init

  type(advar) function pv_eval(this, x) result(y)
    class(pv_t), intent(in) :: this
    real(kp), intent(in) :: x
    integer :: n
    n = nint(this%pars(4)%val)                     ! (a passive parameter used as a switch)
    y = this%pars(1)*exp(-(x/this%pars(2))) + this%pars(3)*(0.1_kp*x)**n
  end function pv_eval
end module passive_val_model

program fit_passive_val
  use passive_val_model
  use gadfit
  implicit none
  integer, parameter :: n = 2000
  type(pv_t) :: f
  real(kp) :: x(n), y2(n), y3(n)
  real(kp), parameter :: truth(3) = [5.0_kp, 2.0_kp, 0.7_kp]
  integer :: i, order
  logical :: ok
  do i = 1, n
     x(i) = 10.0_kp*(real(i, kp) - 0.5_kp)/real(n, kp)
     y2(i) = truth(1)*exp(-(x(i)/truth(2))) + truth(3)*(0.1_kp*x(i))**2 + 1.0e-4_kp*sin(real(mod(37*(i - 1), 1000), kp))
     y3(i) = truth(1)*exp(-(x(i)/truth(2))) + truth(3)*(0.1_kp*x(i))**3 + 1.0e-4_kp*sin(real(mod(37*(i - 1), 1000), kp))
  end do
  ok = .true.
  call gadf_init(f)
  call gadf_add_dataset(x, y2)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output="/dev/null")
  do order = 2, 3
     if (order == 3) then
        ! other data for the same model (a new gadf_init would do as well; here only the passive value and the data values change)
        call gadf_close()
        call gadf_init(f)
        call gadf_add_dataset(x, y3)
        call gadf_set_errors(NONE)
        call gadf_set_verbosity(output="/dev/null")
     end if
     call gadf_set('amp', 4.5_kp, .true.); call gadf_set('tau', 2.3_kp, .true.); call gadf_set('coef', 0.5_kp, .true.)
     call gadf_set('order', real(order, kp), .false.)
     call gadf_fit(1.0, max_iter=8)
     do i = 1, 3
        write(*, '(a, i0, a, i0, a, es25.17)') 'order ', order, '  par ', i, ' = ', fitfuncs(1)%pars(i)%val
        ok = ok .and. abs(fitfuncs(1)%pars(i)%val - truth(i)) < 2e-3_kp*abs(truth(i))
     end do
     ! and again with the WRONG exponent on the same data, without a new gadf_init: the model must be captured again, and the fit
     ! of a wrong model cannot reach the true coefficient
     call gadf_set('amp', 4.5_kp, .true.); call gadf_set('tau', 2.3_kp, .true.); call gadf_set('coef', 0.5_kp, .true.)
     call gadf_set('order', real(5 - order, kp), .false.)
     call gadf_fit(1.0, max_iter=8)
     write(*, '(a, i0, a, es25.17)') 'order ', 5 - order, ' on the other data: coef = ', fitfuncs(1)%pars(3)%val
     ok = ok .and. abs(fitfuncs(1)%pars(3)%val - truth(3)) > 2e-2_kp
  end do
  call gadf_close()
  if (ok) then
     print '(a)', 'PASS'
  else
     print '(a)', 'FAIL'
     error stop 1
  end if
end program fit_passive_val
