! GPU parity test through the Fortran API for an eval() that does PLAIN REAL ARITHMETIC on the abscissa
! (x**2, sin(0.05*x)): operator overloading cannot see inside it (x is real(kp), fitfunction.F90:59-63),
! so the recorder finds literals that are not affine in x and tabulates them once per data point as
! auxiliary columns (GFH_AUX, gfh_set_aux).  The fitted parameters are printed with 17 digits; the Python
! test repeats the fit through the Python API, where the same arithmetic on a symbolic x is recorded
! directly, and compares.  Data: tests/golden/gaussian_xy.txt (path = argument 1).
module real_x_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: rx_t
   contains
     procedure :: init => rx_init
     procedure :: eval => rx_eval
  end type rx_t
contains
  subroutine rx_init(this)
    class(rx_t), intent(out) :: this
    allocate(this%pars(6))
    call this%set(1, 'fmax'); call this%set(2, 'x0'); call this%set(3, 'a'); call this%set(4, 'bgr')
    call this%set(5, 'quad'); call this%set(6, 'wave')
  end subroutine rx_init

  type(advar) function rx_eval(this, x) result(y)
    class(rx_t), intent(in) :: this
    real(kp), intent(in) :: x
    real(kp) :: q, s
    q = x**2*1.0e-4_kp                ! real arithmetic: invisible to the recorder
    s = sin(0.05_kp*x)
    y = this%pars(1)*exp(-((x-this%pars(2))/this%pars(3))**2) + this%pars(4) + this%pars(5)*q + this%pars(6)*s
  end function rx_eval
end module real_x_model

program fit_real_x_functions
  use real_x_model
  use gadfit
  implicit none
  type(rx_t) :: f
  character(len=512) :: path
  integer :: i
  call get_command_argument(1, path)
  call gadf_init(f)
  call gadf_add_dataset(trim(path))
  call gadf_set('fmax', 1.0, .true.)
  call gadf_set('x0', 1e-12_kp, .false.)
  call gadf_set('a', 1.0, .true.)
  call gadf_set('bgr', 1.0, .true.)
  call gadf_set('quad', 0.1, .true.)
  call gadf_set('wave', 0.1, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(timings=.true., output="/dev/null")
  call gadf_fit(0.1, accth=0.9, max_iter=4)
  do i = 1, 6
     write(*, '(a, i0, a, es25.17)') 'par ', i, ' = ', fitfuncs(1)%pars(i)%val
  end do
  write(*, '(a, i0, a, es25.17)') 'iterations = ', gadf_iterations, ' chi2 = ', gadf_chi2
  call gadf_print(points=11, output='/tmp/gadfit_real_x_print')      ! curve + _parameters + _log after the fit
  call gadf_close()
  print '(a)', 'DONE'
end program fit_real_x_functions
