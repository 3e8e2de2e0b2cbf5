! A real that eval() forms from the %val of a FITTED parameter TOGETHER with the abscissa: s = cos(rate%val*x) enters the model
! through plain real arithmetic.  The reference recomputes it at every point of every pass (gadfit.F90:679-690) and its AD never sees
! it: the residuals follow the parameter through s, the Jacobian does not.  No pseudo-parameter can carry a real that has another
! value at every point; the recorder finds the literal that moves with the abscissa (observe: a per-point column) AND with the
! parameters (probe_pars -- at the path's second abscissa: at x = 0, the first one of these data, s is 1 whatever the rate) and
! on_pars tabulates the column anew -- eval() at every data point, the reference's way -- before every pass whose parameters differ
! from those of the last tabulation.  Rounds 1-4 refused this program (refused_literals.F90, mode 'pval').
! Argument 1 = number of data points (default 500: the oracle's case; from GADFIT_HIP_THREADS_FROM on the columns are read off
! recordings made on several threads); argument 2 = 'accel': with geodesic acceleration (STEP 3 runs at the parameters of the sweep after
! a trial chi2() at other parameters: the column is tabulated back); 'fd': use_ad=.false. -- the reference's forward differences call
! eval() at p + step e_j, where s has moved (fitfunction.F90:155-174): the columns go to the device in 1 + n_active sets, one per
! evaluation (gfh_set_fd_column_sets; tabulate_all); 'blackbox': use_ad=.false. and eval() does EVERYTHING in plain real arithmetic
! on %val and assigns the result -- the case use_ad=.false. exists for (a function the AD types cannot express): the whole model is
! one column, the device forms the differences, J^T J and the sums.  (Expected values of both: the oracle's finite-difference fit,
! cases param_val_x_fd and param_x_blackbox_fd -- the same numbers, finite differences see only the function.)
! 'intermediate': the real is taken from the %val of an INTERMEDIATE AD variable (t = rate*x; s = cos(t%val)) -- the same number,
! but a recorder that skips the values of its nodes finds t%val = 0; 'stateful': the real waits in a module variable between two
! statements of eval() (legal under the reference's one call at a time) -- the same number when eval() is called from one thread.
! Both at 2e5 points (tests/test_fortran_binding.py): the columns are refreshed before every pass, on threads only where that is safe.
! Expected values (500 points): the oracle's fit of the same model written with value() = GFH_VAL
! (tests/golden/make_branching_goldens.py, case param_val_x); same data by the same formula.
module param_val_x_model
  use ad
  use fitfunction
  use gadf_constants
  implicit none
  logical :: blackbox = .false., branchy = .false., intermediate = .false., stateful = .false.
  real(kp) :: parked = 0.0_kp
  type, extends(fitfunc) :: pvx_t
   contains
     procedure :: init => pvx_init
     procedure :: eval => pvx_eval
  end type pvx_t
contains
  subroutine pvx_init(this)
    class(pvx_t), intent(out) :: this
    allocate(this%pars(3))
    call this%set(1, 'amp'); call this%set(2, 'rate'); call this%set(3, 'bgr')
  end subroutine pvx_init

  type(advar) function pvx_eval(this, x) result(y)
    class(pvx_t), intent(in) :: this
    real(kp), intent(in) :: x
    real(kp) :: s, acc
    type(advar) :: t
    integer :: k
    if (blackbox) then
       s = this%pars(1)%val*exp(-(this%pars(2)%val*x))*(1.0_kp + 0.1_kp*cos(this%pars(2)%val*x)) + this%pars(3)%val
       y = s
       return
    end if
    if (intermediate) then
       t = this%pars(2)*x
       s = cos(t%val)
    else if (stateful) then
       parked = this%pars(2)%val*x
       acc = 0.0_kp
       do k = 1, 40                          ! (some work between the write and the read)
          acc = acc + sqrt(real(k, kp) + x)
       end do
       s = cos(parked) + 0.0_kp*acc
    else
       s = cos(this%pars(2)%val*x)
    end if
    y = this%pars(1)*exp(-(this%pars(2)*x))*(1.0_kp + 0.1_kp*s) + this%pars(3)
    ! ('branch': a comparison with a fitted parameter that points cross during the fit, the far side the same function written with
    ! one more operation -- two paths through eval(), each with the column; the numbers of the fit stay what they are)
    if (branchy) then
       if (x > 2.0_kp*this%pars(2)) y = y*1.0_kp
    end if
  end function pvx_eval
end module param_val_x_model

program fit_param_val_x
  use param_val_x_model
  use gadfit
  implicit none
  integer :: n
  type(pvx_t) :: f
  real(kp), allocatable :: x(:), y(:)
  real(kp), parameter :: truth(3) = [3.0_kp, 0.8_kp, 0.5_kp]
  real(kp) :: expected(3)
  logical :: accel, fd
  real(kp) :: tol
  character(len=32) :: arg
  integer :: i
  logical :: ok
  n = 500; arg = ''
  if (command_argument_count() >= 1) then
     call get_command_argument(1, arg); read(arg, *) n
  end if
  accel = .false.
  if (command_argument_count() >= 2) then
     call get_command_argument(2, arg); accel = trim(arg) == 'accel'
  end if
  expected = [2.9999991148716143_kp, 0.79997790331478069_kp, 0.4999758446630409_kp]
  if (accel) expected = [2.9999989926128823_kp, 0.79997740905480808_kp, 0.49997566873843419_kp]
  fd = trim(arg) == 'fd' .or. trim(arg) == 'blackbox'; blackbox = trim(arg) == 'blackbox'
  branchy = trim(arg) == 'branch'
  intermediate = trim(arg) == 'intermediate'; stateful = trim(arg) == 'stateful'
  if (fd) expected = [2.9999997213407092_kp, 0.79997922369558339_kp, 0.49997679026851183_kp]
  tol = merge(1e-6_kp, 1e-10_kp, fd)           ! (finite differences divide the last bits of a value by sqrt(epsilon)*p)
  allocate(x(n), y(n))
  do i = 1, n
     x(i) = 5.0_kp*real(i - 1, kp)/real(n - 1, kp)
     y(i) = truth(1)*exp(-(truth(2)*x(i)))*(1.0_kp + 0.1_kp*cos(truth(2)*x(i))) + truth(3) + 1.0e-3_kp*sin(real(mod(37*(i - 1), 1000), kp))
  end do
  call gadf_init(f)
  call gadf_add_dataset(x, y)
  call gadf_set('amp', 2.5_kp, .true.)
  call gadf_set('rate', 0.9_kp, .true.)
  call gadf_set('bgr', 0.3_kp, .true.)
  call gadf_set_errors(NONE)
  call gadf_set_verbosity(output="/dev/null")
  if (fd) then
     call gadf_fit(1.0, max_iter=6, use_ad=.false.)
  else if (accel) then
     call gadf_fit(1.0, max_iter=6, accth=0.9)
  else
     call gadf_fit(1.0, max_iter=6)
  end if
  ok = gadf_iterations == 6
  do i = 1, 3
     if (n == 500) then
        write(*, '(a, i0, a, es25.17, a, es10.2)') 'par ', i, ' = ', fitfuncs(1)%pars(i)%val, '   rel. dev. ', &
             & abs(fitfuncs(1)%pars(i)%val - expected(i))/abs(expected(i))
        ok = ok .and. abs(fitfuncs(1)%pars(i)%val - expected(i)) <= tol*abs(expected(i))
     else          ! (other sizes: other noise, the same truth)
        write(*, '(a, i0, a, es25.17)') 'par ', i, ' = ', fitfuncs(1)%pars(i)%val
        ok = ok .and. abs(fitfuncs(1)%pars(i)%val - truth(i)) <= 1e-3_kp*abs(truth(i))
     end if
  end do
  call gadf_close()
  if (ok) then
     print '(a)', 'PASS'
  else
     print '(a)', 'FAIL'
     error stop 1
  end if
end program fit_param_val_x
