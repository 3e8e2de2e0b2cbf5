! ISO_C_BINDING interface to libgadfit_hip.so (include/gadfit_hip.h).  This is the only
! place where the Fortran driver crosses into the HIP library: no CUDA shims, no dual paths.
module gadfit_hip_c
  use, intrinsic :: iso_c_binding
  implicit none
  public

  type, bind(c) :: gfh_subtape_c
     integer(c_int32_t) :: n_nodes, result
     type(c_ptr) :: nodes
  end type gfh_subtape_c

  type, bind(c) :: gfh_tape_c
     integer(c_int32_t) :: n_pars, n_subtapes
     type(c_ptr) :: sub
     integer(c_int32_t) :: n_integrals
     type(c_ptr) :: integrals, ipar_nodes
     integer(c_int32_t) :: gk_points, n_aux
     real(c_double) :: rel_error_outer, rel_error_inner
     integer(c_int32_t) :: ws_size, ws_size_inner
  end type gfh_tape_c

  type, bind(c) :: gfh_fit_options_c
     real(c_double) :: lambda, lam_up, lam_down, accth, grad_chi2, cos_phi, rel_error, &
          & rel_error_global, chi2_rel, chi2_abs
     integer(c_int) :: has_lambda, has_lam_up, has_lam_down, has_accth, has_grad_chi2, &
          & has_cos_phi, has_rel_error, has_rel_error_global, has_chi2_rel, has_chi2_abs
     type(c_ptr) :: DTD_min
     integer(c_int) :: lam_incs, has_lam_incs, uphill, has_uphill, max_iter, has_max_iter, &
          & damp_max, has_damp_max, nielsen, has_nielsen, umnigh, has_umnigh, verbosity
     real(c_double) :: umnigh_a
  end type gfh_fit_options_c

  type, bind(c) :: gfh_fit_result_c
     integer(c_int) :: iterations, dim, dof, exit_reason
     real(c_double) :: lambda, chi2
     integer(c_int) :: n_sweeps, n_chi2, n_omega, n_lookahead
     real(c_double) :: seconds
  end type gfh_fit_result_c

  interface
     integer(c_int) function gfh_create(device, ctx) bind(c, name='gfh_create')
       import c_int, c_ptr
       integer(c_int), value :: device
       type(c_ptr), intent(out) :: ctx
     end function gfh_create
     ! ... returning at once, the device part of the creation on a thread of the library (gadfit_hip.h)
     integer(c_int) function gfh_create_begin(device, ctx) bind(c, name='gfh_create_begin')
       import c_int, c_ptr
       integer(c_int), value :: device
       type(c_ptr), intent(out) :: ctx
     end function gfh_create_begin

     ! single-process device group: one member context and host thread per GPU behind one handle
     integer(c_int) function gfh_create_group(n_devices, devices, ctx) bind(c, name='gfh_create_group')
       import c_int, c_ptr
       integer(c_int), value :: n_devices
       type(c_ptr), value :: devices
       type(c_ptr), intent(out) :: ctx
     end function gfh_create_group

     integer(c_int) function gfh_comm_init_from_env(ctx) bind(c, name='gfh_comm_init_from_env')
       import c_int, c_ptr
       type(c_ptr), value :: ctx
     end function gfh_comm_init_from_env

     subroutine gfh_destroy(ctx) bind(c, name='gfh_destroy')
       import c_ptr
       type(c_ptr), value :: ctx
     end subroutine gfh_destroy

     type(c_ptr) function gfh_last_error(ctx) bind(c, name='gfh_last_error')
       import c_ptr
       type(c_ptr), value :: ctx
     end function gfh_last_error

     integer(c_int) function gfh_set_data(ctx, n_total, x, y, w, n_datasets, data_positions) &
          & bind(c, name='gfh_set_data')
       import c_int, c_int64_t, c_double, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int64_t), value :: n_total
       real(c_double), intent(in) :: x(*), y(*), w(*)
       integer(c_int), value :: n_datasets
       integer(c_int64_t), intent(in) :: data_positions(*)
     end function gfh_set_data

     ! the same, returning at once: the copies run on a thread of the library until the next call on the context
     integer(c_int) function gfh_set_data_begin(ctx, n_total, x, y, w, n_datasets, data_positions) &
          & bind(c, name='gfh_set_data_begin')
       import c_int, c_int64_t, c_double, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int64_t), value :: n_total
       real(c_double), intent(in) :: x(*), y(*), w(*)
       integer(c_int), value :: n_datasets
       integer(c_int64_t), intent(in) :: data_positions(*)
     end function gfh_set_data_begin

     integer(c_int) function gfh_queue_host_copy(ctx, dst, src, bytes) bind(c, name='gfh_queue_host_copy')
       import c_int, c_int64_t, c_ptr
       type(c_ptr), value :: ctx, dst, src
       integer(c_int64_t), value :: bytes
     end function gfh_queue_host_copy

     ! the text reader of gadf_add_dataset(path) (include/gadfit_hip.h): parse once, take the columns over
     integer(c_int) function gfh_read_columns(path, n_columns, cols, n_points) bind(c, name='gfh_read_columns')
       import c_int, c_int64_t, c_ptr, c_char
       character(kind=c_char), intent(in) :: path(*)
       integer(c_int), value :: n_columns
       type(c_ptr), intent(out) :: cols
       integer(c_int64_t), intent(out) :: n_points
     end function gfh_read_columns
     integer(c_int) function gfh_take_columns(cols, x, y, w) bind(c, name='gfh_take_columns')
       import c_int, c_ptr, c_double
       type(c_ptr), value :: cols
       real(c_double), intent(out) :: x(*), y(*), w(*)
     end function gfh_take_columns
     subroutine gfh_free_columns(cols) bind(c, name='gfh_free_columns')
       import c_ptr
       type(c_ptr), value :: cols
     end subroutine gfh_free_columns
     integer(c_int) function gfh_get_abscissas(ctx, x_out) bind(c, name='gfh_get_abscissas')
       import c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: x_out(*)
     end function gfh_get_abscissas
     integer(c_int) function gfh_wait_host_copy(ctx) bind(c, name='gfh_wait_host_copy')
       import c_int, c_ptr
       type(c_ptr), value :: ctx
     end function gfh_wait_host_copy

     integer(c_int) function gfh_set_keep_jacobian(ctx, mode) bind(c, name='gfh_set_keep_jacobian')
       import c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: mode
     end function gfh_set_keep_jacobian

     integer(c_int) function gfh_set_load_balancing(ctx, on) bind(c, name='gfh_set_load_balancing')
       import c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: on
     end function gfh_set_load_balancing

     integer(c_int) function gfh_set_use_ad(ctx, on) bind(c, name='gfh_set_use_ad')
       import c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: on
     end function gfh_set_use_ad
     integer(c_int) function gfh_set_fd_column_sets(ctx, on) bind(c, name='gfh_set_fd_column_sets')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: on
     end function gfh_set_fd_column_sets

     integer(c_int) function gfh_set_loss(ctx, loss) bind(c, name='gfh_set_loss')
       import c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: loss
     end function gfh_set_loss
     integer(c_int) function gfh_set_aux(ctx, n_aux, aux) bind(c, name='gfh_set_aux')
       import c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_aux
       real(c_double), intent(in) :: aux(*)                        ! [n_total, n_aux]: column k contiguous
     end function gfh_set_aux
     integer(c_int) function gfh_init_weights(ctx, error_type) bind(c, name='gfh_init_weights')
       import c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: error_type
     end function gfh_init_weights

     integer(c_int) function gfh_set_model(ctx, tape) bind(c, name='gfh_set_model')
       import c_int, c_ptr, gfh_tape_c
       type(c_ptr), value :: ctx
       type(gfh_tape_c), intent(in) :: tape
     end function gfh_set_model

     ! a branching eval(): one tape per recorded path (include/gadfit_hip.h); tapes = array of pointers to gfh_tape_c
     integer(c_int) function gfh_set_model_variants(ctx, n_variants, tapes, hint_aux) bind(c, name='gfh_set_model_variants')
       import c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_variants, hint_aux
       type(c_ptr), intent(in) :: tapes(*)
     end function gfh_set_model_variants
     integer(c_int) function gfh_model_needs_hint(ctx) bind(c, name='gfh_model_needs_hint')
       import c_int, c_ptr
       type(c_ptr), value :: ctx
     end function gfh_model_needs_hint
     ! one per-point variant column per set of outcomes (gadfit_hip.h): for the next gfh_set_model_variants
     integer(c_int) function gfh_set_variant_hint_columns(ctx, n_tapes, cols) bind(c, name='gfh_set_variant_hint_columns')
       import c_int, c_int32_t, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_tapes
       integer(c_int32_t), intent(in) :: cols(*)
     end function gfh_set_variant_hint_columns
     ! compile (or load from the cache) the kernels of the current model for an active set, without launching: needs no GPU
     integer(c_int) function gfh_model_prepare(ctx, n_act, active_pars) bind(c, name='gfh_model_prepare')
       import c_int, c_int32_t, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_act
       integer(c_int32_t), intent(in) :: active_pars(*)
     end function gfh_model_prepare
     ! per-thread state of the recorder's checking mode (include/gadfit_hip.h)
     ! host CPUs the process may keep busy: affinity mask cut down to the cgroup's CPU quota (ad_tls.c)
     integer(c_int) function gfh_host_cpu_budget() bind(c, name='gfh_host_cpu_budget')
       import c_int
     end function gfh_host_cpu_budget
     subroutine gfh_adchk_load(n, op, a, b, flags, cls, c, alpha, beta) bind(c, name='gfh_adchk_load')
       import c_int, c_int32_t, c_double
       integer(c_int), value :: n
       integer(c_int32_t), intent(in) :: op(*), a(*), b(*), flags(*), cls(*)
       real(c_double), intent(in) :: c(*), alpha(*), beta(*)
     end subroutine gfh_adchk_load
     subroutine gfh_adchk_load_path(k, n, op, a, b, flags, cls, c, alpha, beta) bind(c, name='gfh_adchk_load_path')
       import c_int, c_int32_t, c_double
       integer(c_int), value :: k, n
       integer(c_int32_t), intent(in) :: op(*), a(*), b(*), flags(*), cls(*)
       real(c_double), intent(in) :: c(*), alpha(*), beta(*)
     end subroutine gfh_adchk_load_path
     subroutine gfh_adchk_load_ints(k, nsub, nint, nip, sub, ipar, sub_result, i_integrand, i_lower, i_upper, i_linf, i_uinf, i_nip, &
          & i_rel, i_abs) bind(c, name='gfh_adchk_load_ints')
       import c_int, c_int32_t, c_double
       integer(c_int), value :: k, nsub, nint, nip
       integer(c_int32_t), intent(in) :: sub(*), ipar(*), sub_result(*), i_integrand(*), i_lower(*), i_upper(*), i_linf(*), i_uinf(*), i_nip(*)
       real(c_double), intent(in) :: i_rel(*), i_abs(*)
     end subroutine gfh_adchk_load_ints
     subroutine gfh_adchk_script(n, bits) bind(c, name='gfh_adchk_script')
       import c_int, c_int64_t
       integer(c_int), value :: n
       integer(c_int64_t), value :: bits
     end subroutine gfh_adchk_script
     subroutine gfh_adchk_use(k) bind(c, name='gfh_adchk_use')
       import c_int
       integer(c_int), value :: k
     end subroutine gfh_adchk_use
     subroutine gfh_adchk_begin(x, n_params) bind(c, name='gfh_adchk_begin')
       import c_int, c_double
       real(c_double), value :: x
       integer(c_int), value :: n_params
     end subroutine gfh_adchk_begin
     subroutine gfh_adchk_end(n, diverged, litfail) bind(c, name='gfh_adchk_end')
       import c_int
       integer(c_int), intent(out) :: n, diverged, litfail
     end subroutine gfh_adchk_end
     integer(c_int) function gfh_adchk_aux(cap, vals, nodes) bind(c, name='gfh_adchk_aux')
       import c_int, c_int32_t, c_double
       integer(c_int), value :: cap
       real(c_double), intent(out) :: vals(*)
       integer(c_int32_t), intent(out) :: nodes(*)
     end function gfh_adchk_aux
     integer(c_int) function gfh_model_n_tapes(ctx) bind(c, name='gfh_model_n_tapes')
       import c_int, c_ptr
       type(c_ptr), value :: ctx
     end function gfh_model_n_tapes
     integer(c_int) function gfh_model_n_variants(ctx) bind(c, name='gfh_model_n_variants')
       import c_int, c_ptr
       type(c_ptr), value :: ctx
     end function gfh_model_n_variants
     integer(c_int) function gfh_set_pars_hook(ctx, fn, user) bind(c, name='gfh_set_pars_hook')
       import c_int, c_ptr, c_funptr
       type(c_ptr), value :: ctx, user
       type(c_funptr), value :: fn
     end function gfh_set_pars_hook
     integer(c_int) function gfh_set_unseen_handler(ctx, fn, user) bind(c, name='gfh_set_unseen_handler')
       import c_int, c_ptr, c_funptr
       type(c_ptr), value :: ctx, user
       type(c_funptr), value :: fn
     end function gfh_set_unseen_handler

     integer(c_int) function gfh_fit(ctx, pars, n_act, active_pars, is_global, opt, res) &
          & bind(c, name='gfh_fit')
       import c_int, c_int32_t, c_double, c_ptr, gfh_fit_options_c, gfh_fit_result_c
       type(c_ptr), value :: ctx
       real(c_double), intent(in out) :: pars(*)
       integer(c_int), value :: n_act
       integer(c_int32_t), intent(in) :: active_pars(*), is_global(*)
       type(gfh_fit_options_c), intent(in out) :: opt
       type(gfh_fit_result_c), intent(out) :: res
     end function gfh_fit

     integer(c_int) function gfh_get_timers(ctx, out8) bind(c, name='gfh_get_timers')
       import c_int, c_double, c_ptr
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: out8(8)
     end function gfh_get_timers
     subroutine gfh_reset_timers(ctx) bind(c, name='gfh_reset_timers')
       import c_ptr
       type(c_ptr), value :: ctx
     end subroutine gfh_reset_timers

     integer(c_int) function gfh_chi2(ctx, pars, chi2) bind(c, name='gfh_chi2')
       import c_int, c_double, c_ptr
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: pars(*)
       real(c_double), intent(out) :: chi2
     end function gfh_chi2

  end interface

contains

  ! C string -> Fortran string
  function c_message(p) result(s)
    type(c_ptr), intent(in) :: p
    character(:), allocatable :: s
    character(kind=c_char), pointer :: ch(:)
    integer :: n
    s = ''
    if (.not. c_associated(p)) return
    call c_f_pointer(p, ch, [4096])
    n = 0
    do while (n < 4096)
       if (ch(n+1) == c_null_char) exit
       n = n + 1
    end do
    allocate(character(n) :: s)
    s = transfer(ch(1:n), s)
  end function c_message
end module gadfit_hip_c
