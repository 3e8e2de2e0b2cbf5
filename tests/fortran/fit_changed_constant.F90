! eval() may read anything it likes -- a module variable of the user's, say -- and the reference calls it afresh at every point of
! every fit (gadfit.F90:679-690), so a value changed BETWEEN two calls of gadf_fit takes effect in the second.  Here such a value sits
! in the captured model as a literal: a later gadf_fit records every path once more at its first abscissa, finds the literal
! changed and captures the model again.  Model amp*exp(-rate*stretch*x), data made with rate*stretch = 0.5: the first fit
! (stretch = 1) finds rate = 0.5, the second (stretch = 2) must find 0.25.
module stretch_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  real(kp) :: stretch = 1.0_kp
  type, extends(fitfunc) :: stretch_t
   contains
     procedure :: init => stretch_init
     procedure :: eval => stretch_eval
  end type stretch_t
contains
  subroutine stretch_init(this)
    class(stretch_t), intent(out) :: this
    allocate(this%pars(2))
    call this%set(1, 'amp'); call this%set(2, 'rate')
  end subroutine stretch_init

  type(advar) function stretch_eval(this, x) result(y)
    class(stretch_t), intent(in) :: this
    real(kp), intent(in) :: x
    y = this%pars(1)*exp(-this%pars(2)*(stretch*x))
  end function stretch_eval
end module stretch_model

program fit_changed_constant
  use stretch_model
  use gadfit
  implicit none
  type(stretch_t) :: f
  integer, parameter :: n = 500
  real(kp), target, save :: xs(n), ys(n)
  integer :: i
  do i = 1, n
     xs(i) = 0.01_kp*i
     ys(i) = 3.0_kp*exp(-0.5_kp*xs(i))
  end do
  call gadf_init(f)
  call gadf_add_dataset(xs, ys)
  call gadf_set('amp', 2.0_kp, .true.)
  call gadf_set('rate', 0.3_kp, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output='/dev/null')
  call gadf_fit(1.0, max_iter=50)
  write(*, '(a, 2es25.17)') 'fit 1: ', fitfuncs(1)%pars(1)%val, fitfuncs(1)%pars(2)%val
  stretch = 2.0_kp
  call gadf_fit(1.0, max_iter=50)
  write(*, '(a, 2es25.17)') 'fit 2: ', fitfuncs(1)%pars(1)%val, fitfuncs(1)%pars(2)%val
  call gadf_close()
  write(*, '(a)') 'DONE'
end program fit_changed_constant
