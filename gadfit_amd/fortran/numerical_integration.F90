! module numerical_integration -- integrate(f, pars, lower, upper [, rel_error, abs_error])
! with the reference's nine bound combinations (fortran/gadfit/numerical_integration.F90:53-58)
! and its constants (INFINITY, GAUSS_KRONROD_*P), as a RECORDER: during model capture the
! integrand is run once with a recording integration variable and recording pars(:), which
! yields an integrand sub-tape; the adaptive Gauss-Kronrod rule itself (NI:193-284, 636-664)
! executes per data point inside the generated HIP kernels (libgadfit_hip codegen).
! OUTSIDE a recording -- gadf_print drawing the fitted curve, a program calling eval() itself after the fit (the reference's
! own 2_integral_single / 3_integral_double do the former) -- integrate() is evaluated on the host through module ad's
! overloaded arithmetic (host_integral below): a handful of abscissas, never part of gadf_fit, whose every pass is the device's.
module numerical_integration

  use, intrinsic :: iso_c_binding
  use ad
  use gadf_constants, only: kp
  use messaging

  implicit none

  private
  public :: integrate, INFINITY, GAUSS_KRONROD_15P, GAUSS_KRONROD_21P, GAUSS_KRONROD_31P, &
       & GAUSS_KRONROD_41P, GAUSS_KRONROD_51P, GAUSS_KRONROD_61P, init_integration, &
       & init_integration_dbl, set_integration_rule, free_integration, &
       & int_rel_error_outer, int_rel_error_inner, int_rule, int_ws_size, int_ws_size_inner, &
       & workspace, workspace_init, workspace_destroy, numerical_integration_memory_report

  ! NI:26-37
  integer, parameter :: INFINITY = 521207248
  integer, parameter :: GAUSS_KRONROD_15P = 15, GAUSS_KRONROD_21P = 21, GAUSS_KRONROD_31P = 31, &
       & GAUSS_KRONROD_41P = 41, GAUSS_KRONROD_51P = 51, GAUSS_KRONROD_61P = 61

  ! Default relative errors of the inner and outer integral (NI:61-62) and the rule; they
  ! travel with the model tape (gfh_tape.rel_error_outer / rel_error_inner / gk_points).
  real(kp) :: int_rel_error_inner = 1e2_kp*epsilon(1.0_kp)
  real(kp) :: int_rel_error_outer = 1e2_kp*epsilon(1.0_kp)
  integer :: int_rule = GAUSS_KRONROD_15P
  logical :: have_inner_ws = .false.
  ! Workspace sizes = the number of intervals an adaptive integral may use before "Number of iterations was insufficient"
  ! (NI:40, 84-98, 251, 282-283); 0 = the reference's default of 1000.  They travel with the tape (gfh_tape.ws_size / ws_size_inner).
  integer :: int_ws_size = 0, int_ws_size_inner = 0

  abstract interface
     type(advar) function integrand(x, pars)
       import advar
       type(advar), intent(in) :: x
       type(advar), intent(in out) :: pars(:)
     end function integrand
  end interface

  interface
     integer(c_int) function gfh_gk_rule(points, roots, wg, wk) bind(c, name='gfh_gk_rule')     ! include/gadfit_hip.h
       import c_int, c_double
       integer(c_int), value :: points
       real(c_double), intent(out) :: roots(*), wg(*), wk(*)
     end function gfh_gk_rule
  end interface

  ! host side (host_integral): the rule's tables as the library holds them, one workspace per nesting level (NI:40-51, 70)
  real(c_double), allocatable, save :: gk_roots(:), gk_wg(:), gk_wk(:)
  type workspace                                         ! NI:40-51, public there as here
     type(advar), allocatable :: sums(:)
     real(kp), allocatable :: lower(:), upper(:), abs_error(:)
   contains
     procedure :: init => workspace_init
     procedure :: destroy => workspace_destroy
  end type workspace
  type(workspace), save :: hws(2)
  integer, save :: host_depth = 0, host_used(2) = 0

  interface integrate
     module procedure integrate_real_real, integrate_real_inf, integrate_inf_real, &
          & integrate_inf_inf, integrate_advar_advar, integrate_advar_real, &
          & integrate_real_advar, integrate_advar_inf, integrate_inf_advar
  end interface integrate

contains

  ! NI:114-123.  Without an inner workspace the outer default equals the inner one.
  subroutine init_integration(rel_error, workspace_size, integration_rule)
    real(kp), intent(in), optional :: rel_error
    integer, intent(in), optional :: workspace_size, integration_rule
    if (.not. have_inner_ws) int_rel_error_outer = int_rel_error_inner
    if (present(rel_error)) int_rel_error_outer = rel_error
    int_ws_size = 0                                   ! ws(1)%init(workspace_size), NI:120
    if (present(workspace_size)) int_ws_size = workspace_size
    call set_integration_rule(integration_rule)
  end subroutine init_integration

  ! NI:127-135
  subroutine init_integration_dbl(rel_error_inner, rel_error_outer, ws_size_inner, ws_size_outer, integration_rule)
    real(kp), intent(in), optional :: rel_error_inner, rel_error_outer
    integer, intent(in), optional :: ws_size_inner, ws_size_outer, integration_rule
    if (present(rel_error_inner)) int_rel_error_inner = rel_error_inner
    int_ws_size_inner = 0                             ! ws(2)%init(ws_size_inner), NI:134
    if (present(ws_size_inner)) int_ws_size_inner = ws_size_inner
    have_inner_ws = .true.
    int_rel_error_outer = 1e3_kp*epsilon(1.0_kp)
    call init_integration(rel_error_outer, ws_size_outer, integration_rule)
  end subroutine init_integration_dbl

  ! NI:139-171
  subroutine set_integration_rule(rule)
    integer, intent(in), optional :: rule
    int_rule = GAUSS_KRONROD_15P
    if (present(rule)) then
       select case (rule)
       case (15, 21, 31, 41, 51, 61)
          int_rule = rule
       case default
          call error(__FILE__, __LINE__, 'Invalid input. The following rules are available: &
               &GAUSS_KRONROD_15p, GAUSS_KRONROD_21p, GAUSS_KRONROD_31p, GAUSS_KRONROD_41p, &
               &GAUSS_KRONROD_51p, and GAUSS_KRONROD_61p.')
       end select
    end if
  end subroutine set_integration_rule

  subroutine free_integration()
    int_rel_error_inner = 1e2_kp*epsilon(1.0_kp)
    int_rel_error_outer = 1e2_kp*epsilon(1.0_kp)
    int_rule = GAUSS_KRONROD_15P
    have_inner_ws = .false.
    int_ws_size = 0; int_ws_size_inner = 0
    call hws%destroy()
    host_used = 0
  end subroutine free_integration

  ! The common recorder.  lo_node/up_node: nodes in the enclosing sub-tape (ignored when the
  ! matching *_inf /= 0); probe: a finite abscissa at which the integrand is run once.
  type(advar) function record_integral(f, pars, lo_node, up_node, lo_inf, up_inf, probe, &
       & rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    integer, intent(in) :: lo_node, up_node, lo_inf, up_inf
    real(kp), intent(in) :: probe
    real(kp), intent(in), optional :: rel_error, abs_error
    type(advar) :: xi, yi
    type(advar), allocatable :: ip(:)
    type(gfh_integral), allocatable :: tmp(:)
    integer, allocatable :: itmp(:)
    integer(c_int32_t), allocatable :: ntmp(:)
    integer :: parent, s, j, off, idx
    real(kp) :: rel_, abs_
    if (.not. ad_recording) call error(__FILE__, __LINE__, 'integrate() runs on the device: it is &
         &only meaningful inside a fitting function handed to gadf_fit.')
    if (ad_thread_check) then
       ! recordings on several threads at once (gadfit.F90: discover / tabulate): nothing of module ad's capture state is written;
       ! the call site's bookkeeping and its comparison with the known recording live in the thread's own storage (ad_tls.c)
       if (gfh_adchk_depth() >= 2) call error(__FILE__, __LINE__, 'Integrals can be nested at most twice.')
       allocate(ntmp(size(pars)))
       do j = 1, size(pars)
          ntmp(j) = anode(pars(j))
       end do
       call gfh_adchk_ipar(int(size(pars), c_int), ntmp)
       s = gfh_adchk_sub_enter()
       xi%val = probe
       xi%node = ad_emit(GFH_IVAR, -1, -1, 0, 0.0_kp)
       allocate(ip(size(pars)))
       do j = 1, size(pars)
          ip(j)%val = pars(j)%val
          ip(j)%node = ad_emit(GFH_IPARAM, j - 1, -1, 0, 0.0_kp)
       end do
       yi = f(xi, ip)
       call gfh_adchk_sub_leave(int(anode(yi), c_int))
       if (present(rel_error)) then; rel_ = rel_error; else; rel_ = -1.0_kp; end if
       if (present(abs_error)) then; abs_ = abs_error; else; abs_ = -1.0_kp; end if
       idx = gfh_adchk_integral(int(s, c_int), int(lo_node, c_int), int(up_node, c_int), int(lo_inf, c_int), int(up_inf, c_int), &
            & int(size(pars), c_int), rel_, abs_)
       y%val = 0.0_kp
       y%node = ad_emit(GFH_INTEGRATE, idx - 1, -1, 0, 0.0_kp)
       return
    end if
    if (ad_depth >= 2) call error(__FILE__, __LINE__, 'Integrals can be nested at most twice.')   ! ws(2), NI:70
    if (ad_nsub >= AD_MAX_SUB) call error(__FILE__, __LINE__, 'Too many integrate() call sites.')
    ! bindings: the pars(:) of this call as nodes of the enclosing sub-tape
    off = ad_n_ipar
    if (off + size(pars) > size(ad_ipar_nodes)) then
       allocate(ntmp(2*(off + size(pars))))
       ntmp(:off) = ad_ipar_nodes(:off)
       call move_alloc(ntmp, ad_ipar_nodes)
    end if
    do j = 1, size(pars)
       ad_ipar_nodes(off + j) = anode(pars(j))
    end do
    ad_n_ipar = off + size(pars)
    ! record the integrand into its own sub-tape
    parent = ad_cur
    ad_nsub = ad_nsub + 1
    s = ad_nsub
    ad_cur = s
    ad_depth = ad_depth + 1
    xi%val = probe
    xi%node = ad_emit(GFH_IVAR, -1, -1, 0, 0.0_kp)
    allocate(ip(size(pars)))
    do j = 1, size(pars)
       ip(j)%val = pars(j)%val
       ip(j)%node = ad_emit(GFH_IPARAM, j - 1, -1, 0, 0.0_kp)
    end do
    yi = f(xi, ip)
    ad_sub_result(s) = anode(yi)
    ad_depth = ad_depth - 1
    ad_cur = parent
    ! the call site
    if (ad_n_integrals == size(ad_integrals)) then
       allocate(tmp(2*size(ad_integrals)), itmp(2*size(ad_integrals)))
       tmp(:ad_n_integrals) = ad_integrals(:ad_n_integrals)
       itmp(:ad_n_integrals) = ad_int_sub(:ad_n_integrals)
       call move_alloc(tmp, ad_integrals)
       call move_alloc(itmp, ad_int_sub)
    end if
    ad_n_integrals = ad_n_integrals + 1
    idx = ad_n_integrals
    ad_integrals(idx)%integrand = s
    ad_integrals(idx)%lower = lo_node
    ad_integrals(idx)%upper = up_node
    ad_integrals(idx)%lower_inf = lo_inf
    ad_integrals(idx)%upper_inf = up_inf
    ad_integrals(idx)%n_ipars = size(pars)
    ad_integrals(idx)%ipar_off = off
    ad_integrals(idx)%depth = ad_depth + 1
    ad_integrals(idx)%rel_error = -1.0_kp
    ad_integrals(idx)%abs_error = -1.0_kp
    if (present(rel_error)) ad_integrals(idx)%rel_error = rel_error
    if (present(abs_error)) ad_integrals(idx)%abs_error = abs_error
    ad_int_sub(idx) = parent
    y%val = 0.0_kp
    y%node = ad_emit(GFH_INTEGRATE, idx - 1, -1, 0, 0.0_kp)
  end function record_integral

  ! the abscissa at which an integrand is recorded: ad_theta of the way through the range, in the variable the quadrature runs over
  ! (NI:314-318, 347-351 for the half-infinite ranges; both infinite: the tangent map).  gadf_fit records an integrand that
  ! compares AD variables at several ad_theta, so that every path through it is met.
  real(kp) function probe_at(lo, hi, lo_inf, up_inf) result(t)
    real(kp), intent(in) :: lo, hi
    integer, intent(in) :: lo_inf, up_inf
    real(kp) :: th
    th = min(max(ad_theta, 1e-9_kp), 1.0_kp - 1e-9_kp)
    if (lo_inf == 0 .and. up_inf == 0) then
       t = lo + th*(hi - lo)
    else if (lo_inf == 0) then
       t = merge(lo - 1.0_kp + 1.0_kp/(1.0_kp - th), lo + 1.0_kp - 1.0_kp/(1.0_kp - th), up_inf > 0)
    else if (up_inf == 0) then
       t = merge(hi + 1.0_kp - 1.0_kp/(1.0_kp - th), hi - 1.0_kp + 1.0_kp/(1.0_kp - th), lo_inf < 0)
    else
       t = tan(3.14159265358979323846_kp*(th - 0.5_kp))
    end if
  end function probe_at

  integer function inf_flag(v, what) result(s)
    integer, intent(in) :: v
    character(*), intent(in) :: what
    if (v == INFINITY) then
       s = 1
    else if (v == -INFINITY) then
       s = -1
    else
       s = 0
       call error(__FILE__, __LINE__, 'Incorrect '//what//' bound. Use either (+-)INFINITY or a real number.')
    end if
  end function inf_flag

  ! NI:84-106: the interval arrays of one nesting level (1000 intervals unless a size is given)
  subroutine workspace_init(this, size)
    class(workspace), intent(in out) :: this
    integer, intent(in), optional :: size
    integer :: n
    n = 1000
    if (present(size)) n = size
    if (allocated(this%lower)) call this%destroy()
    allocate(this%lower(n), this%upper(n), this%sums(n), this%abs_error(n))
  end subroutine workspace_init

  impure elemental subroutine workspace_destroy(this)
    class(workspace), intent(in out) :: this
    if (allocated(this%lower)) deallocate(this%lower)
    if (allocated(this%upper)) deallocate(this%upper)
    if (allocated(this%sums)) deallocate(this%sums)
    if (allocated(this%abs_error)) deallocate(this%abs_error)
  end subroutine workspace_destroy

  ! NI:669-716: the intervals asked for and, for the integrals the HOST evaluated (gadf_print, eval() outside gadf_fit), used.
  ! What the device's passes used is in gfh_counters / gadf_set_verbosity(memory=.true.).
  subroutine numerical_integration_memory_report(io_unit)
    use, intrinsic :: iso_fortran_env, only: output_unit
    integer, intent(in), optional :: io_unit
    integer :: u
    u = output_unit
    if (present(io_unit)) u = io_unit
    write(u, '(1x, g0)') 'Numerical integration memory usage'
    write(u, '(1x, g0)') '=================================='
    write(u, '(2x, g0, i0, g0, i0, g0)') 'Requested: 4x', merge(int_ws_size, 1000, int_ws_size > 0), ' intervals (outer), 4x', &
         & merge(int_ws_size_inner, 1000, int_ws_size_inner > 0), ' (inner)'
    write(u, '(7x, g0, i0, g0, i0, g0)') 'Used on the host: 4x', host_used(1), ' (outer), 4x', host_used(2), ' (inner)'
  end subroutine numerical_integration_memory_report

  ! ---- integrate() outside a recording: the reference's adaptive rule (NI:193-284) on the host, through module ad's arithmetic.
  ! kind = 0: the range [lower, upper].  kind = +1 / -1: the half-infinite range anchor .. +inf / -inf .. anchor mapped to (0, 1]
  ! by x = anchor - 1 + 1/t / x = anchor + 1 - 1/t, integrand f(x)/t**2 (NI:310-318, 343-351); lower = 0, upper = 1 then.
  ! The interval with the largest error estimate is halved until the estimates add up to less than the bound; the parameters are
  ! passive while that goes on (their indices set to zero, NI:238-239) and the intervals are summed once more with the indices
  ! back, so that a caller in forward or reverse mode gets the derivatives of the final sum, as in the reference (NI:268-275).
  type(advar) recursive function host_integral(f, pars, lower, upper, kind, anchor, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    real(kp), intent(in) :: lower, upper, anchor
    integer, intent(in) :: kind
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: level, n_ws, n, k, i
    integer, allocatable :: saved(:)
    real(kp) :: rel_, abs_, a, b, mid, err_total, sum_total, e_
    if (host_depth >= 2) call error(__FILE__, __LINE__, 'Integrals can be nested at most twice.')
    host_depth = host_depth + 1
    level = host_depth
    if (.not. allocated(gk_roots)) allocate(gk_roots(1))
    if (size(gk_roots) /= int_rule) then
       deallocate(gk_roots); if (allocated(gk_wg)) deallocate(gk_wg, gk_wk)
       allocate(gk_roots(int_rule), gk_wg(int_rule/2), gk_wk(int_rule))
       if (gfh_gk_rule(int(int_rule, c_int), gk_roots, gk_wg, gk_wk) /= 0) call error(__FILE__, __LINE__, 'Unknown Gauss-Kronrod rule.')
    end if
    n_ws = merge(int_ws_size, int_ws_size_inner, level == 1)
    if (n_ws <= 0) n_ws = 1000                           ! NI:40
    if (allocated(hws(level)%lower)) then
       if (size(hws(level)%lower) /= n_ws) call hws(level)%destroy()
    end if
    if (.not. allocated(hws(level)%lower)) call hws(level)%init(n_ws)
    if (present(rel_error)) then                         ! NI:227-235
       rel_ = rel_error
    else
       rel_ = merge(int_rel_error_outer, int_rel_error_inner, level == 1)
    end if
    abs_ = 0.0_kp
    if (present(abs_error)) abs_ = abs_error
    saved = pars%index
    pars%index = 0
    associate(w => hws(level))
      w%lower(1) = lower; w%upper(1) = upper
      w%sums(1) = host_panel(f, pars, lower, upper, kind, anchor, w%abs_error(1))
      do n = 1, n_ws - 1
         k = maxloc(w%abs_error(:n), 1)
         a = w%lower(k); b = w%upper(k); mid = (a + b)/2
         w%sums(k) = host_panel(f, pars, a, mid, kind, anchor, w%abs_error(k))
         w%sums(n+1) = host_panel(f, pars, mid, b, kind, anchor, w%abs_error(n+1))
         w%upper(k) = mid; w%lower(n+1) = mid; w%upper(n+1) = b
         err_total = sum(w%abs_error(:n+1)); sum_total = sum(w%sums(:n+1)%val)
         if (err_total < abs_ .or. err_total/sum_total < rel_) then
            pars%index = saved
            host_used(level) = max(host_used(level), n)
            y = 0.0_kp
            do i = 1, n + 1
               y = y + host_panel(f, pars, w%lower(i), w%upper(i), kind, anchor, e_)
            end do
            host_depth = host_depth - 1
            return
         end if
      end do
    end associate
    pars%index = saved
    host_depth = host_depth - 1
    call error(__FILE__, __LINE__, 'Number of iterations was insufficient. Increase either workspace size or the error bound(s).')
  end function host_integral

  ! One Gauss-Kronrod panel over [a, b] (NI:636-664): the Kronrod sum through module ad's arithmetic, the Gauss sum of the
  ! values at the even positions beside it; their difference is the panel's error estimate.
  type(advar) recursive function host_panel(f, pars, a, b, kind, anchor, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    real(kp), intent(in) :: a, b, anchor
    integer, intent(in) :: kind
    real(kp), intent(out) :: abs_error
    type(advar) :: arg, fv
    real(kp) :: half, centre, gauss, t
    integer :: i
    half = (b - a)/2; centre = (a + b)/2
    gauss = 0.0_kp
    y = 0.0_kp
    do i = 1, size(gk_roots)
       t = half*gk_roots(i) + centre
       if (kind == 0) then
          arg%val = t
          fv = f(arg, pars)
       else
          arg%val = merge(anchor - 1.0_kp + 1.0_kp/t, anchor + 1.0_kp - 1.0_kp/t, kind > 0)
          fv = f(arg, pars)/(t*t)
       end if
       if (mod(i, 2) == 0) gauss = gauss + gk_wg(i/2)*fv%val
       y = y + gk_wk(i)*fv
    end do
    y = half*y
    abs_error = abs(y%val - half*gauss)
  end function host_panel

  ! the derivative of an integral with respect to an ACTIVE bound, host side, forward mode (NI:427-440): the integrand at the
  ! bound times the bound's derivative, sign = +1 for the upper bound, -1 for the lower one
  subroutine host_bound_terms(f, pars, bound, sign, y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    type(advar), intent(in) :: bound
    real(kp), intent(in) :: sign
    type(advar), intent(in out) :: y
    type(advar) :: arg, at_value, along
    if (bound%index == 0) return
    if (reverse_mode) call error(__FILE__, __LINE__, 'integrate() with an active bound in reverse mode is recorded for the &
         &device only: outside a fitting function handed to gadf_fit use the forward mode.')
    arg%val = bound%val
    at_value = f(arg, pars)
    along = f(bound, pars)
    y%d = y%d + sign*bound%d*at_value%val
    y%dd = y%dd + sign*(bound%dd*at_value%val + bound%d*(along%d + at_value%d))
    if (y%index == 0) y%index = 1
  end subroutine host_bound_terms

  ! ---- the nine specifics (NI:193-630)
  type(advar) recursive function integrate_real_real(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    real(kp), intent(in) :: lower, upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: ln, un
    if (.not. ad_recording) then
       y = host_integral(f, pars, lower, upper, 0, 0.0_kp, rel_error, abs_error)
       return
    end if
    ln = rnode(lower); un = rnode(upper)
    y = record_integral(f, pars, ln, un, 0, 0, probe_at(lower, upper, 0, 0), rel_error, abs_error)
  end function integrate_real_real

  type(advar) recursive function integrate_real_inf(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    real(kp), intent(in) :: lower
    integer, intent(in) :: upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: ln
    if (.not. ad_recording) then
       if (inf_flag(upper, 'upper') > 0) then
          y = host_integral(f, pars, 0.0_kp, 1.0_kp, 1, lower, rel_error, abs_error)
       else                                             ! NI:308: minus the integral from -inf to lower
          y = -host_integral(f, pars, 0.0_kp, 1.0_kp, -1, lower, rel_error, abs_error)
       end if
       return
    end if
    ln = rnode(lower)
    y = record_integral(f, pars, ln, -1, 0, inf_flag(upper, 'upper'), probe_at(lower, 0.0_kp, 0, inf_flag(upper, 'upper')), rel_error, abs_error)
  end function integrate_real_inf

  type(advar) recursive function integrate_inf_real(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    integer, intent(in) :: lower
    real(kp), intent(in) :: upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: un
    if (.not. ad_recording) then
       if (inf_flag(lower, 'lower') < 0) then
          y = host_integral(f, pars, 0.0_kp, 1.0_kp, -1, upper, rel_error, abs_error)
       else                                             ! NI:341: minus the integral from upper to +inf
          y = -host_integral(f, pars, 0.0_kp, 1.0_kp, 1, upper, rel_error, abs_error)
       end if
       return
    end if
    un = rnode(upper)
    y = record_integral(f, pars, -1, un, inf_flag(lower, 'lower'), 0, probe_at(0.0_kp, upper, inf_flag(lower, 'lower'), 0), rel_error, abs_error)
  end function integrate_inf_real

  type(advar) recursive function integrate_inf_inf(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    integer, intent(in) :: lower, upper
    real(kp), intent(in), optional :: rel_error, abs_error
    if (.not. ad_recording) then                        ! NI:367-368: the two halves about zero
       y = integrate_inf_real(f, pars, lower, 0.0_kp, rel_error, abs_error) + integrate_real_inf(f, pars, 0.0_kp, upper, rel_error, abs_error)
       return
    end if
    y = record_integral(f, pars, -1, -1, inf_flag(lower, 'lower'), inf_flag(upper, 'upper'), probe_at(0.0_kp, 0.0_kp, -1, 1), &
         & rel_error, abs_error)
  end function integrate_inf_inf

  type(advar) recursive function integrate_advar_advar(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    type(advar), intent(in) :: lower, upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: ln, un
    if (.not. ad_recording) then
       y = host_integral(f, pars, lower%val, upper%val, 0, 0.0_kp, rel_error, abs_error)
       call host_bound_terms(f, pars, lower, -1.0_kp, y); call host_bound_terms(f, pars, upper, 1.0_kp, y)
       return
    end if
    ln = anode(lower); un = anode(upper)
    y = record_integral(f, pars, ln, un, 0, 0, probe_at(lower%val, upper%val, 0, 0), rel_error, abs_error)
  end function integrate_advar_advar

  type(advar) recursive function integrate_advar_real(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    type(advar), intent(in) :: lower
    real(kp), intent(in) :: upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: ln, un
    if (.not. ad_recording) then
       y = host_integral(f, pars, lower%val, upper, 0, 0.0_kp, rel_error, abs_error)
       call host_bound_terms(f, pars, lower, -1.0_kp, y)
       return
    end if
    ln = anode(lower); un = rnode(upper)
    y = record_integral(f, pars, ln, un, 0, 0, probe_at(lower%val, upper, 0, 0), rel_error, abs_error)
  end function integrate_advar_real

  type(advar) recursive function integrate_real_advar(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    real(kp), intent(in) :: lower
    type(advar), intent(in) :: upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: ln, un
    if (.not. ad_recording) then
       y = host_integral(f, pars, lower, upper%val, 0, 0.0_kp, rel_error, abs_error)
       call host_bound_terms(f, pars, upper, 1.0_kp, y)
       return
    end if
    ln = rnode(lower); un = anode(upper)
    y = record_integral(f, pars, ln, un, 0, 0, probe_at(lower, upper%val, 0, 0), rel_error, abs_error)
  end function integrate_real_advar

  type(advar) recursive function integrate_advar_inf(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    type(advar), intent(in) :: lower
    integer, intent(in) :: upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: ln
    if (.not. ad_recording) then
       y = integrate_real_inf(f, pars, lower%val, upper, rel_error, abs_error)
       call host_bound_terms(f, pars, lower, -1.0_kp, y)
       return
    end if
    ln = anode(lower)
    y = record_integral(f, pars, ln, -1, 0, inf_flag(upper, 'upper'), probe_at(lower%val, 0.0_kp, 0, inf_flag(upper, 'upper')), rel_error, abs_error)
  end function integrate_advar_inf

  type(advar) recursive function integrate_inf_advar(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    integer, intent(in) :: lower
    type(advar), intent(in) :: upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: un
    if (.not. ad_recording) then
       y = integrate_inf_real(f, pars, lower, upper%val, rel_error, abs_error)
       call host_bound_terms(f, pars, upper, 1.0_kp, y)
       return
    end if
    un = anode(upper)
    y = record_integral(f, pars, -1, un, inf_flag(lower, 'lower'), 0, probe_at(0.0_kp, upper%val, inf_flag(lower, 'lower'), 0), rel_error, abs_error)
  end function integrate_inf_advar
end module numerical_integration
