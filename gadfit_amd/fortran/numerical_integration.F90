! module numerical_integration -- integrate(f, pars, lower, upper [, rel_error, abs_error])
! with the reference's nine bound combinations (fortran/gadfit/numerical_integration.F90:53-58)
! and its constants (INFINITY, GAUSS_KRONROD_*P), as a RECORDER: during model capture the
! integrand is run once with a recording integration variable and recording pars(:), which
! yields an integrand sub-tape; the adaptive Gauss-Kronrod rule itself (NI:193-284, 636-664)
! executes per data point inside the generated HIP kernels (libgadfit_hip codegen).
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
       & int_rel_error_outer, int_rel_error_inner, int_rule, int_ws_size, int_ws_size_inner

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

  ! ---- the nine specifics (NI:193-630)
  type(advar) function integrate_real_real(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    real(kp), intent(in) :: lower, upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: ln, un
    ln = rnode(lower); un = rnode(upper)
    y = record_integral(f, pars, ln, un, 0, 0, probe_at(lower, upper, 0, 0), rel_error, abs_error)
  end function integrate_real_real

  type(advar) function integrate_real_inf(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    real(kp), intent(in) :: lower
    integer, intent(in) :: upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: ln
    ln = rnode(lower)
    y = record_integral(f, pars, ln, -1, 0, inf_flag(upper, 'upper'), probe_at(lower, 0.0_kp, 0, inf_flag(upper, 'upper')), rel_error, abs_error)
  end function integrate_real_inf

  type(advar) function integrate_inf_real(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    integer, intent(in) :: lower
    real(kp), intent(in) :: upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: un
    un = rnode(upper)
    y = record_integral(f, pars, -1, un, inf_flag(lower, 'lower'), 0, probe_at(0.0_kp, upper, inf_flag(lower, 'lower'), 0), rel_error, abs_error)
  end function integrate_inf_real

  type(advar) function integrate_inf_inf(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    integer, intent(in) :: lower, upper
    real(kp), intent(in), optional :: rel_error, abs_error
    y = record_integral(f, pars, -1, -1, inf_flag(lower, 'lower'), inf_flag(upper, 'upper'), probe_at(0.0_kp, 0.0_kp, -1, 1), &
         & rel_error, abs_error)
  end function integrate_inf_inf

  type(advar) function integrate_advar_advar(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    type(advar), intent(in) :: lower, upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: ln, un
    ln = anode(lower); un = anode(upper)
    y = record_integral(f, pars, ln, un, 0, 0, probe_at(lower%val, upper%val, 0, 0), rel_error, abs_error)
  end function integrate_advar_advar

  type(advar) function integrate_advar_real(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    type(advar), intent(in) :: lower
    real(kp), intent(in) :: upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: ln, un
    ln = anode(lower); un = rnode(upper)
    y = record_integral(f, pars, ln, un, 0, 0, probe_at(lower%val, upper, 0, 0), rel_error, abs_error)
  end function integrate_advar_real

  type(advar) function integrate_real_advar(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    real(kp), intent(in) :: lower
    type(advar), intent(in) :: upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: ln, un
    ln = rnode(lower); un = anode(upper)
    y = record_integral(f, pars, ln, un, 0, 0, probe_at(lower, upper%val, 0, 0), rel_error, abs_error)
  end function integrate_real_advar

  type(advar) function integrate_advar_inf(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    type(advar), intent(in) :: lower
    integer, intent(in) :: upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: ln
    ln = anode(lower)
    y = record_integral(f, pars, ln, -1, 0, inf_flag(upper, 'upper'), probe_at(lower%val, 0.0_kp, 0, inf_flag(upper, 'upper')), rel_error, abs_error)
  end function integrate_advar_inf

  type(advar) function integrate_inf_advar(f, pars, lower, upper, rel_error, abs_error) result(y)
    procedure(integrand) :: f
    type(advar), intent(in out) :: pars(:)
    integer, intent(in) :: lower
    type(advar), intent(in) :: upper
    real(kp), intent(in), optional :: rel_error, abs_error
    integer :: un
    un = anode(upper)
    y = record_integral(f, pars, -1, un, inf_flag(lower, 'lower'), 0, probe_at(0.0_kp, upper%val, inf_flag(lower, 'lower'), 0), rel_error, abs_error)
  end function integrate_inf_advar
end module numerical_integration
