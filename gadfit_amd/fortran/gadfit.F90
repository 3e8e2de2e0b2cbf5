! module gadfit -- the procedural driver API (drop-in for fortran/gadfit/gadfit.F90:41-58):
! gadf_init, gadf_add_dataset, gadf_set, gadf_set_errors, gadf_set_verbosity, gadf_fit,
! gadf_print, gadf_close and the readable fitfuncs(:).
!
! The Levenberg-Marquardt hot path -- STEP 1/2/3 and chi2() of the reference's gadf_fit
! (gadfit.F90:674-743, 1015-1034) -- runs on the GPU behind libgadfit_hip.so, reached
! through ISO_C_BINDING (gadfit_hip_c).  This module only marshals: it concatenates the
! datasets, captures the user's eval() once into a model tape, and hands parameter blocks
! to gfh_fit.  No coarrays: multi-GPU sharding lives in the library (RCCL).
module gadfit

  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: real32, output_unit
  use ad
  use fitfunction
  use gadf_constants
  use gadfit_hip_c
  use messaging
  use numerical_integration

  implicit none

  private
  public :: gadf_init, gadf_add_dataset, gadf_set, gadf_set_errors, gadf_set_verbosity, &
       & gadf_fit, gadf_print, gadf_close, fitfuncs, gadf_iterations, gadf_chi2, gadf_set_loss
  public :: LOSS_LINEAR, LOSS_CAUCHY, LOSS_HUBER
  public :: NONE, SQRT_Y, PROPTO_Y, INVERSE_Y, USER, GLOBAL, LOCAL, GLOBAL_AND_LOCAL

  ! data_error_type (gadfit.F90:45-48)
  integer, parameter :: NONE = 0, SQRT_Y = 1, PROPTO_Y = 2, INVERSE_Y = 3, USER = 4
  integer, parameter :: GLOBAL = 0, LOCAL = 1, GLOBAL_AND_LOCAL = 2
  ! robust cost functions of the C++ solver (c++/gadfit/lm_solver.h:76-83); not in the Fortran reference
  integer, parameter :: LOSS_LINEAR = 0, LOSS_CAUCHY = 1, LOSS_HUBER = 2
  integer :: loss_type = LOSS_LINEAR

  interface gadf_add_dataset
     module procedure gadf_add_dataset_file, gadf_add_dataset_data
  end interface gadf_add_dataset

  interface gadf_set
     module procedure set_int_local_real, set_int_local_real32, set_int_global_real, &
          & set_int_global_real32, set_char_local_real, set_char_local_real32, &
          & set_char_global_real, set_char_global_real32
  end interface gadf_set

  type data_pointer
     real(kp), pointer :: x_data(:) => null(), y_data(:) => null(), weights(:) => null()
     character(:), allocatable :: path
  end type data_pointer

  ! There are as many instances of the fitting function as there are datasets.
  class(fitfunc), allocatable, protected :: fitfuncs(:)
  integer, allocatable :: active_pars(:)          ! 1-based index or 0 (gadfit.F90:68)
  logical, allocatable :: is_global(:)
  real(kp), allocatable, target :: x_data(:), y_data(:), weights(:)
  integer(c_int64_t), allocatable :: data_positions(:)   ! 0-based offsets for the library
  type(data_pointer), allocatable :: data_pointers(:)
  integer :: n_added, data_error_type, set_count, verbosity
  logical :: show_timings = .false.
  integer :: last_n_omega = 0
  real(c_double) :: last_seconds = 0
  integer :: gadf_iterations
  real(kp) :: gadf_chi2
  real(kp) :: umnigh_a = 0.5_kp                    ! the SAVEd local of gadfit.F90:515
  type(c_ptr) :: ctx = c_null_ptr
  logical :: model_captured, data_uploaded, lb_on = .false.
  ! real(kp) functions of x that eval() forms in plain real arithmetic (invisible to the recorder): their
  ! positions in the raw recording; tabulated per data point by tabulate_aux (GFH_AUX columns)
  integer :: n_aux_cols = 0, n_raw_nodes = 0
  integer, allocatable :: aux_raw_k(:)
  ! how capture_model classified every real literal of the raw recording (verify_capture checks it against the data):
  ! 0 not a literal of eval(), 1 constant lit_c, 2 affine lit_alpha*x + lit_beta, 3 auxiliary column
  integer, allocatable :: lit_class(:), raw_op(:), raw_a(:), raw_b(:)
  real(kp), allocatable :: lit_c(:), lit_alpha(:), lit_beta(:)
  logical, allocatable :: force_aux(:)            ! literals verify_capture found to be neither: tabulated per point instead
  integer, parameter :: VERIFY_ALL_UP_TO = 131072

contains

  subroutine lib_check(rc, file, line)
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: file
    integer, intent(in) :: line
    if (rc /= 0) call error(file, line, c_message(gfh_last_error(ctx)))
  end subroutine lib_check

  ! gadfit.F90:133-184.  The AD / quadrature workspace sizes are accepted for source
  ! compatibility; the tape is recorded once per model, not per point.
  subroutine gadf_init(f, num_datasets, sweep_size, trace_size, const_size, ws_size, &
       & ws_size_inner, integration_rule, ad_memory, rel_error_inner, rel_error)
    class(fitfunc), intent(in) :: f
    integer, intent(in), optional :: num_datasets, sweep_size, trace_size, const_size, &
         & ws_size, ws_size_inner, integration_rule
    character(*), intent(in), optional :: ad_memory
    real(kp), intent(in), optional :: rel_error_inner, rel_error
    integer :: i, n, device, stat, n_group
    character(len=16) :: env
    if (allocated(fitfuncs)) call gadf_close()
    ! gadfit.F90:166-172: quadrature tolerances / rule
    call free_integration()
    if (present(rel_error_inner) .or. present(ws_size_inner)) then
       call init_integration_dbl(rel_error_inner, rel_error, ws_size_inner, ws_size, integration_rule)
    else if (present(rel_error) .or. present(ws_size) .or. present(integration_rule)) then
       call init_integration(rel_error, ws_size, integration_rule)
    end if
    n = 1
    if (present(num_datasets)) n = num_datasets
    allocate(fitfuncs(n), mold=f)
    do i = 1, n
       call fitfuncs(i)%init()
    end do
    allocate(active_pars(size(fitfuncs(1)%pars)), is_global(size(fitfuncs(1)%pars)), &
         & data_pointers(n), data_positions(n+1))
    active_pars = 0; is_global = .false.
    data_positions = 0
    n_added = 0; set_count = 0; data_error_type = NONE; verbosity = 1
    gadf_iterations = 0; gadf_chi2 = 0.0_kp
    model_captured = .false.; data_uploaded = .false.; lb_on = .false.
    if (allocated(force_aux)) deallocate(force_aux)
    device = 0
    call get_environment_variable('GADFIT_HIP_DEVICE', env, status=stat)
    if (stat == 0) read(env, *, iostat=stat) device
    ! GADFIT_HIP_DEVICES = n (or 'all'): this one process drives n GPUs, one image per GPU as a
    ! host thread inside the library (device group; replaces num_images() images without a launcher)
    n_group = 0
    call get_environment_variable('GADFIT_HIP_DEVICES', env, status=stat)
    if (stat == 0) then
       if (trim(adjustl(env)) == 'all') then
          n_group = -1
       else
          read(env, *, iostat=stat) n_group
          if (stat /= 0 .or. n_group < 1) call error(__FILE__, __LINE__, &
               & 'GADFIT_HIP_DEVICES must be a positive number of devices or "all".')
       end if
    end if
    if (n_group /= 0) then
       call lib_check(gfh_create_group(int(max(n_group, 0), c_int), c_null_ptr, ctx), __FILE__, __LINE__)
    else
       call lib_check(gfh_create(int(device, c_int), ctx), __FILE__, __LINE__)
    end if
    ! The Jacobian has no reader behind this API (JacobianT is private in the reference, gadfit.F90:60-64): the fused
    ! STEP 1+2 kernel writes it only for the fits whose options read it back (gfh_set_keep_jacobian mode 2; same
    ! J^T J / J^T r / chi2 bit for bit).  GADFIT_HIP_KEEP_J overrides.
    call get_environment_variable('GADFIT_HIP_KEEP_J', env, status=stat)
    if (stat /= 0 .and. (device >= 0 .or. n_group /= 0)) call lib_check(gfh_set_keep_jacobian(ctx, 2_c_int), __FILE__, __LINE__)
    ! one process per GPU, started by a plain shell loop: GADFIT_HIP_NRANKS / _RANK / _IDFILE
    ! (replaces num_images()/this_image(); no-op when unset)
    if (device >= 0) call lib_check(gfh_comm_init_from_env(ctx), __FILE__, __LINE__)
  end subroutine gadf_init

  ! gadfit.F90:189-222: the file is read in read_data
  subroutine gadf_add_dataset_file(path)
    character(*), intent(in) :: path
    if (.not. allocated(fitfuncs)) call error(__FILE__, __LINE__, &
         & 'Number of datasets is undetermined. Call gadf_init first.')
    if (n_added >= size(fitfuncs)) call error(__FILE__, __LINE__, 'Too many calls to gadf_add_dataset.')
    n_added = n_added + 1
    data_pointers(n_added)%path = path
  end subroutine gadf_add_dataset_file

  ! gadfit.F90:226-246: user arrays are borrowed by pointer until the first gadf_fit
  subroutine gadf_add_dataset_data(x_data, y_data, weights)
    real(kp), intent(in), target :: x_data(:), y_data(:)
    real(kp), intent(in), target, optional :: weights(:)
    if (.not. allocated(fitfuncs)) call error(__FILE__, __LINE__, &
         & 'Number of datasets is undetermined. Call gadf_init first.')
    if (n_added >= size(fitfuncs)) call error(__FILE__, __LINE__, 'Too many calls to gadf_add_dataset.')
    n_added = n_added + 1
    data_pointers(n_added)%x_data => x_data
    data_pointers(n_added)%y_data => y_data
    if (present(weights)) data_pointers(n_added)%weights => weights
  end subroutine gadf_add_dataset_data

  ! gadfit.F90:255-273
  subroutine set_int_local_real(dataset_i, par, val, active)
    integer, intent(in) :: dataset_i, par
    real(kp), intent(in) :: val
    logical, intent(in), optional :: active
    if (.not. allocated(fitfuncs)) call error(__FILE__, __LINE__, &
         & 'Number of datasets is undetermined. Call gadf_init first.')
    if (dataset_i > size(fitfuncs)) call error(__FILE__, __LINE__, &
         & 'Invalid dataset index. Call gadf_init with the correct number of datasets.')
    is_global(par) = .false.
    call fitfuncs(dataset_i)%set(par, val)
    active_pars(par) = 0
    if (present(active)) then
       if (active) active_pars(par) = par
    end if
    set_count = set_count + 1
  end subroutine set_int_local_real

  subroutine set_int_local_real32(dataset_i, par, val, active)
    integer, intent(in) :: dataset_i, par
    real(real32), intent(in) :: val
    logical, intent(in), optional :: active
    call set_int_local_real(dataset_i, par, real(val, kp), active)
  end subroutine set_int_local_real32

  ! gadfit.F90:284-293
  subroutine set_int_global_real(par, val, active)
    integer, intent(in) :: par
    real(kp), intent(in) :: val
    logical, intent(in), optional :: active
    integer :: i
    if (.not. allocated(fitfuncs)) call error(__FILE__, __LINE__, &
         & 'Number of datasets is undetermined. Call gadf_init first.')
    do i = 1, size(fitfuncs)
       call set_int_local_real(i, par, val, active)
    end do
    is_global(par) = .true.
  end subroutine set_int_global_real

  subroutine set_int_global_real32(par, val, active)
    integer, intent(in) :: par
    real(real32), intent(in) :: val
    logical, intent(in), optional :: active
    call set_int_global_real(par, real(val, kp), active)
  end subroutine set_int_global_real32

  subroutine set_char_local_real(dataset_i, par, val, active)
    integer, intent(in) :: dataset_i
    character(*), intent(in) :: par
    real(kp), intent(in) :: val
    logical, intent(in), optional :: active
    if (.not. allocated(fitfuncs)) call error(__FILE__, __LINE__, &
         & 'Number of datasets is undetermined. Call gadf_init first.')
    call set_int_local_real(dataset_i, fitfuncs(1)%get_index(par), val, active)
  end subroutine set_char_local_real

  subroutine set_char_local_real32(dataset_i, par, val, active)
    integer, intent(in) :: dataset_i
    character(*), intent(in) :: par
    real(real32), intent(in) :: val
    logical, intent(in), optional :: active
    call set_char_local_real(dataset_i, par, real(val, kp), active)
  end subroutine set_char_local_real32

  subroutine set_char_global_real(par, val, active)
    character(*), intent(in) :: par
    real(kp), intent(in) :: val
    logical, intent(in), optional :: active
    if (.not. allocated(fitfuncs)) call error(__FILE__, __LINE__, &
         & 'Number of datasets is undetermined. Call gadf_init first.')
    call set_int_global_real(fitfuncs(1)%get_index(par), val, active)
  end subroutine set_char_global_real

  subroutine set_char_global_real32(par, val, active)
    character(*), intent(in) :: par
    real(real32), intent(in) :: val
    logical, intent(in), optional :: active
    call set_char_global_real(par, real(val, kp), active)
  end subroutine set_char_global_real32

  ! gadfit.F90:356-385: the per-iteration log switch and `timings` are honoured
  subroutine gadf_set_verbosity(scope, digits, timings, memory, workloads, delta1, delta2, &
       & cos_phi, grad_chi2, uphill, acc, output)
    integer, intent(in), optional :: scope, digits
    logical, intent(in), optional :: timings, memory, workloads, delta1, delta2, cos_phi, &
         & grad_chi2, uphill, acc
    character(*), intent(in), optional :: output
    verbosity = 1
    if (present(output)) then
       if (output == '/dev/null') verbosity = 0
    end if
    if (present(timings)) show_timings = timings
  end subroutine gadf_set_verbosity

  ! print_timings (gadfit.F90:1064-1137) for the device path: the reference's phase names with the time the GPU
  ! spent in their kernels (HIP events) and the wall time of the main loop
  subroutine print_device_timings(r)
    type(gfh_fit_result_c), intent(in) :: r
    last_n_omega = r%n_omega; last_seconds = r%seconds
    write(output_unit, '(a)') ''
    call write_device_timings(output_unit)
  end subroutine print_device_timings

  subroutine write_device_timings(u)
    integer, intent(in) :: u
    real(c_double) :: t(8)
    call lib_check(gfh_get_timers(ctx, t), __FILE__, __LINE__)
    write(u, '(1x, a)') 'Timings (device kernel time by phase, HIP events; wall time of the main loop)'
    write(u, '(1x, a, f12.6, a, i0, a)') 'Jacobian + J^T J / J^T r (STEP 1+2): ', t(1) + t(2), ' s  (', nint(t(7)), ' passes)'
    write(u, '(1x, a, f12.6, a, i0, a)') 'Chi2:                               ', t(5), ' s  (', nint(t(8)), ' passes)'
    write(u, '(1x, a, f12.6, a, i0, a)') 'Omega (STEP 3):                     ', t(6), ' s  (', last_n_omega, ' passes)'
    write(u, '(1x, a, f12.6, a)')       'Reduction / all-reduce:             ', t(3) + t(4), ' s'
    write(u, '(1x, a, f12.6, a, i0, a)') 'Main loop (wall):                   ', last_seconds, ' s  (', gadf_iterations, ' iterations)'
  end subroutine write_device_timings

  ! gadfit.F90:392-395
  subroutine gadf_set_errors(e)
    integer, intent(in) :: e
    data_error_type = e
  end subroutine gadf_set_errors

  ! The C++ solver's settings.loss (lm_solver.h:208): applies to the fits that follow.
  subroutine gadf_set_loss(loss)
    integer, intent(in) :: loss
    if (loss < LOSS_LINEAR .or. loss > LOSS_HUBER) call error(__FILE__, __LINE__, 'gadf_set_loss: unknown loss function')
    loss_type = loss
  end subroutine gadf_set_loss

  ! read_data (gadfit.F90:401-443): concatenates all datasets; for USER the third column /
  ! weights argument holds the uncertainties, which init_weights inverts ON THE DEVICE.
  subroutine read_data()
    integer :: i, n, io, u, stat
    integer(c_int64_t) :: j
    real(kp) :: a, b, c
    if (n_added /= size(fitfuncs)) call error(__FILE__, __LINE__, &
         & 'Some datasets are missing. gadf_add_dataset must be called for every dataset.')
    data_positions(1) = 0
    do i = 1, size(fitfuncs)
       if (associated(data_pointers(i)%x_data)) then
          n = size(data_pointers(i)%x_data)
       else
          n = 0
          open(newunit=u, file=data_pointers(i)%path, status='old', action='read', iostat=io)
          if (io /= 0) call error(__FILE__, __LINE__, 'Cannot open '//data_pointers(i)%path)
          do
             read(u, *, iostat=stat) a       ! gadfit.F90:212-215: lines without a number are skipped
             if (stat < 0) exit
             if (stat == 0) n = n + 1
          end do
          close(u)
          if (n == 0) call error(__FILE__, __LINE__, data_pointers(i)%path//' contains no valid data points.')
       end if
       data_positions(i+1) = data_positions(i) + n
    end do
    n = int(data_positions(size(fitfuncs)+1))
    allocate(x_data(n), y_data(n), weights(n))
    if (data_error_type /= USER) weights = 1.0_kp      ! (USER: every element is assigned below)
    do i = 1, size(fitfuncs)
       j = data_positions(i)
       if (associated(data_pointers(i)%x_data)) then
          x_data(j+1:data_positions(i+1)) = data_pointers(i)%x_data
          y_data(j+1:data_positions(i+1)) = data_pointers(i)%y_data
          if (data_error_type == USER) then
             if (.not. associated(data_pointers(i)%weights)) call error(__FILE__, __LINE__, &
                  & 'USER errors requested but no weights were given.')
             weights(j+1:data_positions(i+1)) = data_pointers(i)%weights
          end if
       else
          open(newunit=u, file=data_pointers(i)%path, status='old', action='read')
          do while (j < data_positions(i+1))
             if (data_error_type == USER) then
                read(u, *, iostat=stat) a, b, c
             else
                read(u, *, iostat=stat) a, b
                c = 1.0_kp
             end if
             if (stat < 0) exit
             if (stat == 0) then
                j = j + 1
                x_data(j) = a; y_data(j) = b; weights(j) = c
             end if
          end do
          close(u)
       end if
    end do
  end subroutine read_data

  ! Runs eval() under the recording advar at three abscissas and once with perturbed
  ! parameters, and turns the recorded operation sequence into the model tape:
  !  * structure (ops, operands, integrate() call sites) must be identical across probes
  !    (no data-dependent control flow);
  !  * a literal that is the same in all probes is a constant;
  !  * a literal of eval() that changes with x must be affine in x (covers x, -x, x-c, c*x ...
  !    which is how a real(kp) abscissa enters advar arithmetic) and is rebuilt from the X node;
  !  * a literal that changes when only the parameters change (use of %val) is refused;
  !  * literals inside integrands must not depend on x at all (x reaches an integrand
  !    through its pars(:), as in the reference's examples).
  subroutine capture_model()
    integer, parameter :: NPROBE = 4
    type(gfh_node), allocatable :: probes(:,:)
    integer, allocatable :: psub(:)
    type(gfh_integral), allocatable :: pints(:)
    integer, allocatable :: pint_sub(:), pipar(:)
    type(gfh_node), allocatable, target, save :: final(:)
    type(gfh_subtape_c), allocatable, target, save :: sub(:)
    type(gfh_integral), allocatable, target, save :: ints(:)
    integer(c_int32_t), allocatable, target, save :: ipar(:)
    type(gfh_tape_c) :: tape
    integer, allocatable :: remap(:), first(:), cnt(:), loc(:)
    real(kp), allocatable :: saved(:)
    real(kp) :: xp(NPROBE), alpha, beta, c1, c2, c3, scale
    type(advar) :: y
    integer :: ip, k, n, np, res_node(NPROBE), nf, xnode, i, s, nsub, nint, nip, base
    np = size(fitfuncs(1)%pars)
    allocate(saved(np))
    saved = fitfuncs(1)%pars%val
    ! three distinct abscissas from the data
    xp(1) = x_data(1); xp(2) = xp(1); xp(3) = xp(1)
    do i = 2, size(x_data)
       if (x_data(i) /= xp(1)) then
          xp(2) = x_data(i); exit
       end if
    end do
    do i = size(x_data), 1, -1
       if (x_data(i) /= xp(1) .and. x_data(i) /= xp(2)) then
          xp(3) = x_data(i); exit
       end if
    end do
    if (xp(2) == xp(1)) xp(2) = xp(1)*(1.0_kp + 1e-3_kp) + 1e-3_kp
    if (xp(3) == xp(1) .or. xp(3) == xp(2)) xp(3) = xp(2)*(1.0_kp + 2e-3_kp) + 2e-3_kp
    xp(4) = xp(1)
    n = 0; nsub = 0; nint = 0; nip = 0
    do ip = 1, NPROBE
       call ad_capture_begin()
       do k = 1, np
          call set_node(fitfuncs(1)%pars(k), ad_emit(GFH_PARAM, k-1, -1, 0, 0.0_kp))
       end do
       if (ip == 4) call set_vals(fitfuncs(1)%pars, saved*(1.0_kp + 1.0e-3_kp) + 1.0e-3_kp)
       y = fitfuncs(1)%eval(xp(ip))
       res_node(ip) = anode(y)
       call ad_capture_end()
       if (ad_capture_failed) call error(__FILE__, __LINE__, trim(ad_capture_msg))
       if (ip == 1) then
          n = ad_tape_n; nsub = ad_nsub; nint = ad_n_integrals; nip = ad_n_ipar
          allocate(probes(n, NPROBE), psub(n), pints(max(1, nint)), pint_sub(max(1, nint)), pipar(max(1, nip)))
          psub = ad_sub(:n)
          if (nint > 0) then
             pints(:nint) = ad_integrals(:nint); pint_sub(:nint) = ad_int_sub(:nint)
          end if
          if (nip > 0) pipar(:nip) = ad_ipar_nodes(:nip)
          allocate(first(0:nsub), cnt(0:nsub), loc(0:nsub))
          cnt = ad_sub_n(0:nsub)
       else
          if (ad_tape_n /= n .or. ad_nsub /= nsub .or. ad_n_integrals /= nint .or. ad_n_ipar /= nip) &
               & call control_flow_error()
          if (any(ad_sub(:n) /= psub)) call control_flow_error()
          if (nip > 0) then
             if (any(ad_ipar_nodes(:nip) /= pipar(:nip))) call control_flow_error()
          end if
          do i = 1, nint
             if (ad_integrals(i)%integrand /= pints(i)%integrand .or. ad_integrals(i)%lower /= pints(i)%lower .or. &
                  & ad_integrals(i)%upper /= pints(i)%upper .or. ad_integrals(i)%lower_inf /= pints(i)%lower_inf .or. &
                  & ad_integrals(i)%upper_inf /= pints(i)%upper_inf .or. ad_integrals(i)%n_ipars /= pints(i)%n_ipars) &
                  & call control_flow_error()
          end do
       end if
       probes(:, ip) = ad_tape(:n)
    end do
    call set_vals(fitfuncs(1)%pars, saved)
    do k = 1, np
       call set_node(fitfuncs(1)%pars(k), -1)
    end do
    do ip = 2, NPROBE
       if (res_node(ip) /= res_node(1) .or. any(probes(:,ip)%op /= probes(:,1)%op) .or. &
            & any(probes(:,ip)%a /= probes(:,1)%a) .or. any(probes(:,ip)%b /= probes(:,1)%b)) &
            & call control_flow_error()
    end do
    ! rebuild: sub-tape 0 with x-dependent literals expressed through the X node, integrand
    ! sub-tapes verbatim; all sub-tapes contiguous in `final`
    allocate(remap(0:max(cnt(0) - 1, 0)))
    if (allocated(final)) deallocate(final)
    if (allocated(sub)) deallocate(sub)
    if (allocated(ints)) deallocate(ints)
    if (allocated(ipar)) deallocate(ipar)
    allocate(final(4*n + 8), sub(nsub + 1), ints(max(1, nint)), ipar(max(1, nip)))
    nf = 0; xnode = -1
    n_aux_cols = 0; n_raw_nodes = n
    if (allocated(aux_raw_k)) deallocate(aux_raw_k)
    allocate(aux_raw_k(max(1, n)))
    if (allocated(lit_class)) deallocate(lit_class, lit_c, lit_alpha, lit_beta, raw_op, raw_a, raw_b)
    allocate(lit_class(n), lit_c(n), lit_alpha(n), lit_beta(n), raw_op(n), raw_a(n), raw_b(n))
    lit_class = 0; lit_c = 0.0_kp; lit_alpha = 0.0_kp; lit_beta = 0.0_kp
    raw_op = probes(:,1)%op; raw_a = probes(:,1)%a; raw_b = probes(:,1)%b
    if (allocated(force_aux)) then
       if (size(force_aux) /= n) deallocate(force_aux)
    end if
    if (.not. allocated(force_aux)) then
       allocate(force_aux(n)); force_aux = .false.
    end if
    do s = 0, nsub
       base = nf
       first(s) = nf
       loc(s) = 0
       do k = 1, n
          if (psub(k) /= s) cycle
          associate(nd => probes(k,1))
            if (nd%op == GFH_CONST) then
               c1 = probes(k,1)%c; c2 = probes(k,2)%c; c3 = probes(k,3)%c
               if (probes(k,4)%c /= c1 .and. .not. (c1 /= c1)) call error(__FILE__, __LINE__, &
                    & 'eval() forms a real number from parameter values (%val); such literals &
                    &cannot follow the parameters on the device. Keep them as advar.')
               if (c1 == c2 .and. c1 == c3 .and. .not. (s == 0 .and. force_aux(k))) then
                  call push(GFH_CONST, -1, -1, GFH_F_REAL, c1)
                  if (s == 0) remap(loc(s)) = nf - 1 - base
                  if (s == 0) then
                     lit_class(k) = 1; lit_c(k) = c1
                  end if
               else
                  if (s /= 0) call error(__FILE__, __LINE__, 'A real literal inside an integrand &
                       &depends on x; pass x to the integrand through its pars(:) array.')
                  alpha = (c2 - c1)/(xp(2) - xp(1))
                  if (abs(alpha - 1.0_kp) < 1e-13_kp) alpha = 1.0_kp
                  if (abs(alpha + 1.0_kp) < 1e-13_kp) alpha = -1.0_kp
                  beta = c1 - alpha*xp(1)
                  scale = abs(c1) + abs(alpha*xp(1))
                  if (abs(beta) <= 1e-13_kp*scale) beta = 0.0_kp
                  if (force_aux(k) .or. abs(alpha*xp(3) + beta - c3) > 1e-11_kp*(abs(c3) + abs(alpha*xp(3)) + abs(beta))) then
                     ! not affine in x (x**2, exp(-x), ... in plain real arithmetic): an auxiliary
                     ! per-point input, tabulated on the host once per data point
                     n_aux_cols = n_aux_cols + 1
                     aux_raw_k(n_aux_cols) = k
                     lit_class(k) = 3
                     call push(GFH_AUX, n_aux_cols - 1, -1, GFH_F_REAL, 0.0_kp)
                     remap(loc(s)) = nf - 1
                     loc(s) = loc(s) + 1
                     cycle
                  end if
                  lit_class(k) = 2; lit_alpha(k) = alpha; lit_beta(k) = beta
                  if (xnode < 0) then
                     call push(GFH_X, -1, -1, GFH_F_REAL, 0.0_kp)
                     xnode = nf - 1
                  end if
                  remap(loc(s)) = xnode
                  if (alpha == -1.0_kp) then
                     call push(GFH_NEG, remap(loc(s)), -1, GFH_F_REAL, 0.0_kp)
                     remap(loc(s)) = nf - 1
                  else if (alpha /= 1.0_kp) then
                     call push(GFH_CONST, -1, -1, GFH_F_REAL, alpha)
                     call push(GFH_MUL, nf - 1, remap(loc(s)), GFH_F_REAL, 0.0_kp)
                     remap(loc(s)) = nf - 1
                  end if
                  if (beta /= 0.0_kp) then
                     call push(GFH_CONST, -1, -1, GFH_F_REAL, beta)
                     call push(GFH_ADD, remap(loc(s)), nf - 1, GFH_F_REAL, 0.0_kp)
                     remap(loc(s)) = nf - 1
                  end if
               end if
            else if (s /= 0) then
               call push(nd%op, nd%a, nd%b, nd%flags, 0.0_kp)        ! integrand nodes: verbatim
            else
               select case (nd%op)
               case (GFH_PARAM, GFH_INTEGRATE)
                  call push(nd%op, nd%a, -1, nd%flags, 0.0_kp)
               case (GFH_POWI)
                  call push(nd%op, remap(nd%a), nd%b, nd%flags, 0.0_kp)
               case (GFH_ADD, GFH_SUB, GFH_MUL, GFH_DIV, GFH_POW)
                  call push(nd%op, remap(nd%a), remap(nd%b), nd%flags, 0.0_kp)
               case default
                  call push(nd%op, remap(nd%a), -1, nd%flags, 0.0_kp)
               end select
               remap(loc(s)) = nf - 1
            end if
          end associate
          loc(s) = loc(s) + 1
       end do
       sub(s+1)%n_nodes = nf - base
       sub(s+1)%nodes = c_loc(final(base+1))
       if (s == 0) then
          sub(s+1)%result = remap(res_node(1))
       else
          sub(s+1)%result = ad_sub_result(s)
       end if
    end do
    ! integrate() call sites; those of eval() refer to remapped nodes
    do i = 1, nint
       ints(i) = pints(i)
       if (pint_sub(i) == 0) then
          if (ints(i)%lower_inf == 0) ints(i)%lower = remap(pints(i)%lower)
          if (ints(i)%upper_inf == 0) ints(i)%upper = remap(pints(i)%upper)
       end if
       do k = 1, pints(i)%n_ipars
          if (pint_sub(i) == 0) then
             ipar(pints(i)%ipar_off + k) = remap(pipar(pints(i)%ipar_off + k))
          else
             ipar(pints(i)%ipar_off + k) = pipar(pints(i)%ipar_off + k)
          end if
       end do
    end do
    tape%n_pars = np; tape%n_subtapes = nsub + 1; tape%sub = c_loc(sub)
    tape%n_integrals = nint; tape%integrals = c_loc(ints); tape%ipar_nodes = c_loc(ipar)
    tape%gk_points = int_rule; tape%n_aux = n_aux_cols
    tape%rel_error_outer = int_rel_error_outer; tape%rel_error_inner = int_rel_error_inner
    call lib_check(gfh_set_model(ctx, tape), __FILE__, __LINE__)
    model_captured = .true.
  contains
    subroutine push(op, a, b, flags, c)
      integer, intent(in) :: op, a, b, flags
      real(kp), intent(in) :: c
      nf = nf + 1
      final(nf)%op = op; final(nf)%a = a; final(nf)%b = b; final(nf)%flags = flags; final(nf)%c = c
    end subroutine push
    subroutine control_flow_error()
      call error(__FILE__, __LINE__, 'eval() executes a different operation sequence for &
           &different x or parameters: data-dependent control flow cannot run on the device.')
    end subroutine control_flow_error
  end subroutine capture_model

  ! capture_model classifies the real literals of eval() from three abscissas.  A real function of x that eval() forms in plain
  ! real(kp) arithmetic and that happens to look constant or affine there -- a narrow bump exp(-(x-5)**2/0.01) that underflows at
  ! all three, a window, merge() on x -- would be baked into the tape as a constant, where the reference evaluates eval() at
  ! every point.  So the classification is checked against the DATA: eval() is recorded again at every abscissa (up to
  ! VERIFY_ALL_UP_TO points; beyond that at as many evenly spaced ones, both ends included: one recording costs about as much
  ! as the reference's own evaluation of a point, 1.5 us for the 32-parameter model, and a feature that falls between two
  ! samples spans fewer than N / VERIFY_ALL_UP_TO consecutive points, < 1e-5 of the data) and every literal must be what the
  ! tape says -- the same constant, or alpha*x + beta; the operation sequence must be the recorded one.  A literal that fails is
  ! promoted to an auxiliary per-point column (tabulated at EVERY point by tabulate_aux) and the tape is rebuilt; a different
  ! operation sequence is the control-flow error.  Returns .true. if something was promoted.
  logical function verify_capture() result(promoted)
    type(advar) :: y
    integer :: i, k, np, nchk, step, j
    real(kp) :: c, want
    promoted = .false.
    np = size(fitfuncs(1)%pars)
    nchk = size(x_data)
    step = 1
    if (nchk > VERIFY_ALL_UP_TO) step = (nchk + VERIFY_ALL_UP_TO - 1)/VERIFY_ALL_UP_TO
    i = 1
    do
       call ad_capture_begin()
       do k = 1, np
          call set_node(fitfuncs(1)%pars(k), ad_emit(GFH_PARAM, k-1, -1, 0, 0.0_kp))
       end do
       y = fitfuncs(1)%eval(x_data(i))
       call ad_capture_end()
       if (ad_capture_failed) call error(__FILE__, __LINE__, trim(ad_capture_msg))
       if (ad_tape_n /= n_raw_nodes) call error(__FILE__, __LINE__, 'eval() executes a different &
            &operation sequence for different x: data-dependent control flow cannot run on the device.')
       do j = 1, n_raw_nodes
          if (ad_tape(j)%op /= raw_op(j) .or. ad_tape(j)%a /= raw_a(j) .or. ad_tape(j)%b /= raw_b(j)) &
               & call error(__FILE__, __LINE__, 'eval() executes a different operation sequence for &
               &different x: data-dependent control flow cannot run on the device.')
          if (lit_class(j) == 1) then
             c = ad_tape(j)%c
             if (c /= lit_c(j) .and. .not. (c /= c .and. lit_c(j) /= lit_c(j))) then
                force_aux(j) = .true.; promoted = .true.
             end if
          else if (lit_class(j) == 2) then
             c = ad_tape(j)%c
             want = lit_alpha(j)*x_data(i) + lit_beta(j)
             if (.not. (abs(want - c) <= 1e-11_kp*(abs(c) + abs(lit_alpha(j)*x_data(i)) + abs(lit_beta(j))))) then
                force_aux(j) = .true.; promoted = .true.
             end if
          end if
       end do
       if (i == nchk) exit
       i = min(i + step, nchk)
    end do
    do k = 1, np
       call set_node(fitfuncs(1)%pars(k), -1)
    end do
  end function verify_capture

  ! Auxiliary per-point columns: eval() is recorded once per data point and the literals that
  ! capture_model found to be non-affine functions of x are read out of the recording.
  subroutine tabulate_aux()
    real(c_double), allocatable :: tab(:,:)
    type(advar) :: y
    integer :: i, j, k, np
    np = size(fitfuncs(1)%pars)
    allocate(tab(size(x_data), n_aux_cols))
    do i = 1, size(x_data)
       call ad_capture_begin()
       do k = 1, np
          call set_node(fitfuncs(1)%pars(k), ad_emit(GFH_PARAM, k-1, -1, 0, 0.0_kp))
       end do
       y = fitfuncs(1)%eval(x_data(i))
       call ad_capture_end()
       if (ad_capture_failed) call error(__FILE__, __LINE__, trim(ad_capture_msg))
       if (ad_tape_n /= n_raw_nodes) call error(__FILE__, __LINE__, 'eval() executes a different &
            &operation sequence for different x: data-dependent control flow cannot run on the device.')
       do j = 1, n_aux_cols
          tab(i, j) = ad_tape(aux_raw_k(j))%c
       end do
    end do
    do k = 1, np
       call set_node(fitfuncs(1)%pars(k), -1)
    end do
    call lib_check(gfh_set_aux(ctx, int(n_aux_cols, c_int), tab), __FILE__, __LINE__)
  end subroutine tabulate_aux

  ! fitfuncs is protected: these helpers live in this module so they may modify it
  subroutine set_node(p, node)
    type(advar), intent(in out) :: p
    integer, intent(in) :: node
    p%node = node
  end subroutine set_node

  subroutine set_vals(p, v)
    type(advar), intent(in out) :: p(:)
    real(kp), intent(in) :: v(:)
    p%val = v
  end subroutine set_vals

  ! gadfit.F90:502-1035.  Same optional arguments; the first ten are real(real32).
  subroutine gadf_fit(lambda, lam_up, lam_down, accth, grad_chi2, cos_phi, rel_error, &
       & rel_error_global, chi2_rel, chi2_abs, DTD_min, lam_incs, uphill, max_iter, damp_max, &
       & nielsen, umnigh, load_balancing, use_ad)
    real(real32), intent(in), optional :: lambda, lam_up, lam_down, accth, grad_chi2, cos_phi, &
         & rel_error, rel_error_global, chi2_rel, chi2_abs
    real(kp), intent(in), optional, target :: DTD_min(:)
    integer, intent(in), optional :: lam_incs, uphill, max_iter
    logical, intent(in), optional :: damp_max, nielsen, umnigh, use_ad
    logical, value, optional :: load_balancing
    type(gfh_fit_options_c) :: o
    type(gfh_fit_result_c) :: r
    integer(c_int32_t), allocatable :: act(:), glob(:)
    real(c_double), allocatable :: pars(:,:)
    integer :: i, j, n_act, np, stat_lb
    logical :: want_lb
    character(len=8) :: env_lb
    if (.not. allocated(fitfuncs)) call error(__FILE__, __LINE__, &
         & 'Number of datasets is undetermined. Call gadf_init first.')
    if (.not. allocated(x_data)) call read_data()
    if (.not. model_captured) then
       call capture_model()
       if (verify_capture()) call capture_model()      ! some literal followed x after all: rebuilt with it as a per-point column
    end if
    ! load_balancing (adaptive parallelism, gadfit.F90:672-673): the library re-cuts the ranges of the ranks / group
    ! members between iterations; it makes its host copy of the data when they are set
    want_lb = .false.
    if (present(load_balancing)) want_lb = load_balancing
    call get_environment_variable('GADFIT_HIP_LOAD_BALANCING', env_lb, status=stat_lb)      ! for unchanged programs
    if (stat_lb == 0) want_lb = trim(adjustl(env_lb)) /= '0'
    if (want_lb .neqv. lb_on) then
       call lib_check(gfh_set_load_balancing(ctx, merge(1_c_int, 0_c_int, want_lb)), __FILE__, __LINE__)
       lb_on = want_lb
       if (want_lb) data_uploaded = .false.
    end if
    if (.not. data_uploaded) then
       call lib_check(gfh_set_data(ctx, int(size(x_data), c_int64_t), x_data, y_data, weights, &
            & int(size(fitfuncs), c_int), data_positions), __FILE__, __LINE__)
       call lib_check(gfh_init_weights(ctx, int(data_error_type, c_int)), __FILE__, __LINE__)  ! gadfit.F90:445-470
       if (n_aux_cols > 0) call tabulate_aux()
       data_uploaded = .true.
    end if
    ! compact the active list (gadfit.F90:586-599), 0-based for the library
    np = size(fitfuncs(1)%pars)
    n_act = count(active_pars /= 0)
    if (n_act == 0) call error(__FILE__, __LINE__, 'There are no active parameters.')
    if (set_count < size(fitfuncs)*np) call warning(__FILE__, __LINE__, 'Some parameters might be uninitialized.')
    allocate(act(n_act), glob(np), pars(np, size(fitfuncs)))
    j = 0
    do i = 1, np
       if (active_pars(i) /= 0) then
          j = j + 1
          act(j) = i - 1
       end if
    end do
    glob = merge(1, 0, is_global)
    do i = 1, size(fitfuncs)
       pars(:, i) = fitfuncs(i)%pars%val
    end do
    ! marshal the options; present() -> has_*
    o%has_lambda = 0; o%has_lam_up = 0; o%has_lam_down = 0; o%has_accth = 0; o%has_grad_chi2 = 0
    o%has_cos_phi = 0; o%has_rel_error = 0; o%has_rel_error_global = 0; o%has_chi2_rel = 0; o%has_chi2_abs = 0
    o%has_lam_incs = 0; o%has_uphill = 0; o%has_max_iter = 0; o%has_damp_max = 0; o%has_nielsen = 0; o%has_umnigh = 0
    o%lambda = 0; o%lam_up = 0; o%lam_down = 0; o%accth = 0; o%grad_chi2 = 0; o%cos_phi = 0; o%rel_error = 0
    o%rel_error_global = 0; o%chi2_rel = 0; o%chi2_abs = 0
    o%lam_incs = 0; o%uphill = 0; o%max_iter = 0; o%damp_max = 0; o%nielsen = 0; o%umnigh = 0
    o%DTD_min = c_null_ptr
    if (present(lambda)) then; o%lambda = lambda; o%has_lambda = 1; end if
    if (present(lam_up)) then; o%lam_up = lam_up; o%has_lam_up = 1; end if
    if (present(lam_down)) then; o%lam_down = lam_down; o%has_lam_down = 1; end if
    if (present(accth)) then; o%accth = accth; o%has_accth = 1; end if
    if (present(grad_chi2)) then; o%grad_chi2 = grad_chi2; o%has_grad_chi2 = 1; end if
    if (present(cos_phi)) then; o%cos_phi = cos_phi; o%has_cos_phi = 1; end if
    if (present(rel_error)) then; o%rel_error = rel_error; o%has_rel_error = 1; end if
    if (present(rel_error_global)) then; o%rel_error_global = rel_error_global; o%has_rel_error_global = 1; end if
    if (present(chi2_rel)) then; o%chi2_rel = chi2_rel; o%has_chi2_rel = 1; end if
    if (present(chi2_abs)) then; o%chi2_abs = chi2_abs; o%has_chi2_abs = 1; end if
    if (present(DTD_min)) o%DTD_min = c_loc(DTD_min)
    if (present(lam_incs)) then; o%lam_incs = lam_incs; o%has_lam_incs = 1; end if
    if (present(uphill)) then; o%uphill = uphill; o%has_uphill = 1; end if
    if (present(max_iter)) then; o%max_iter = max_iter; o%has_max_iter = 1; end if
    if (present(damp_max)) then; o%damp_max = merge(1, 0, damp_max); o%has_damp_max = 1; end if
    if (present(nielsen)) then; o%nielsen = merge(1, 0, nielsen); o%has_nielsen = 1; end if
    if (present(umnigh)) then; o%umnigh = merge(1, 0, umnigh); o%has_umnigh = 1; end if
    o%verbosity = verbosity
    o%umnigh_a = umnigh_a
    call lib_check(gfh_set_loss(ctx, int(loss_type, c_int)), __FILE__, __LINE__)
    ! gadfit.F90:583-584, 684-687, 721-728: use_ad=.false. = the finite differences of fitfunction.F90:155-203 on the device
    i = 1
    if (present(use_ad)) then
       if (.not. use_ad) i = 0
    end if
    call lib_check(gfh_set_use_ad(ctx, int(i, c_int)), __FILE__, __LINE__)
    if (show_timings) call gfh_reset_timers(ctx)
    call lib_check(gfh_fit(ctx, pars, int(n_act, c_int), act, glob, o, r), __FILE__, __LINE__)
    gadf_iterations = r%iterations
    last_n_omega = r%n_omega; last_seconds = r%seconds
    if (show_timings) call print_device_timings(r)
    umnigh_a = o%umnigh_a
    do i = 1, size(fitfuncs)
       call set_vals(fitfuncs(i)%pars, pars(:, i))
    end do
    gadf_iterations = r%iterations
    gadf_chi2 = r%chi2
  end subroutine gadf_fit

  ! gadf_print (gadfit.F90:1255-1395): the fitted curves on a grid of `points` abscissas between begin and
  ! end (defaults: the data range, 200 points; logplot: logarithmic spacing) as "x y_1 .. y_n" lines in
  ! `output` (default 'out'), or one "x y" file per dataset `output<k>` with grouped=.false.; after a fit
  ! also `output_parameters` and `output_log`.  The curves are evaluated on the host (the recorder's
  ! elementals carry values); models that call integrate() exist only on the device and are refused here.
  subroutine gadf_print(begin, end, points, output, grouped, logplot, begin_kp, end_kp)
    real(real32), intent(in), optional :: begin, end
    integer, intent(in), optional :: points
    character(*), intent(in), optional :: output
    logical, intent(in), optional :: grouped, logplot
    real(kp), intent(in), optional :: begin_kp, end_kp
    real(kp) :: begin_loc, end_loc
    real(kp), allocatable :: buffer(:,:)
    character(:), allocatable :: output_loc
    character(32) :: num
    type(advar) :: y
    logical :: single, logp
    integer :: points_loc, u, i, j, k
    if (.not. allocated(fitfuncs)) call error(__FILE__, __LINE__, 'Call gadf_init first.')
    if (present(begin_kp)) then
       begin_loc = begin_kp
    else if (present(begin)) then
       begin_loc = begin
    else
       if (.not. allocated(x_data)) then
          if (n_added == 0) call error(__FILE__, __LINE__, 'Since no datasets are loaded, &
               &the lowest x-value must be explicitly given.')
          call read_data()
       end if
       begin_loc = x_data(1)
    end if
    if (present(end_kp)) then
       end_loc = end_kp
    else if (present(end)) then
       end_loc = end
    else
       if (.not. allocated(x_data)) then
          if (n_added == 0) call error(__FILE__, __LINE__, 'Since no datasets are loaded, &
               &the highest x-value must be explicitly given.')
          call read_data()
       end if
       end_loc = x_data(size(x_data))
    end if
    output_loc = 'out'
    if (present(output)) output_loc = output
    points_loc = 200
    if (present(points)) points_loc = max(points, 2)
    logp = .false.
    if (present(logplot)) logp = logplot
    allocate(buffer(size(fitfuncs) + 1, points_loc))
    do i = 1, points_loc
       if (logp) then
          buffer(1, i) = exp(log(begin_loc) + (i-1)*(log(end_loc) - log(begin_loc))/(points_loc - 1))
       else
          buffer(1, i) = begin_loc + (i-1)*(end_loc - begin_loc)/(points_loc - 1)
       end if
    end do
    do j = 1, size(fitfuncs)
       do i = 1, points_loc
          y = fitfuncs(j)%eval(buffer(1, i))
          buffer(j+1, i) = y%val
       end do
    end do
    single = size(fitfuncs) == 1 .or. .not. present(grouped)
    if (present(grouped)) single = single .or. grouped
    if (single) then
       open(newunit=u, file=output_loc, action='write', form='formatted')
       write(num, '(i0)') size(fitfuncs)
       do i = 1, points_loc
          write(u, '(g0, '//trim(num)//'(1x, g0))') buffer(:, i)
       end do
       close(u)
    else
       do k = 1, size(fitfuncs)
          write(num, '(i0)') k
          open(newunit=u, file=output_loc//trim(num), action='write', form='formatted')
          do i = 1, points_loc
             write(u, '(g0, 1x, g0)') buffer(1, i), buffer(1+k, i)
          end do
          close(u)
       end do
    end if
    if (gadf_iterations > 0) then
       open(newunit=u, file=output_loc//'_parameters', action='write', form='formatted')
       write(u, '(a)') 'gadfit (MI355X device path)'
       write(u, '(a, i0, a, es25.17)') 'iterations ', gadf_iterations, '   chi2 ', gadf_chi2
       do i = 1, size(fitfuncs)
          do j = 1, size(fitfuncs(i)%pars)
             write(u, '(i0, 1x, a, 1x, es25.17)') i, fitfuncs(i)%get_name(j), fitfuncs(i)%pars(j)%val
          end do
       end do
       close(u)
       open(newunit=u, file=output_loc//'_log', action='write', form='formatted')
       write(u, '(a)') 'gadfit (MI355X device path)'
       call write_device_timings(u)
       close(u)
    end if
  end subroutine gadf_print

  ! gadfit.F90:1399-1412
  subroutine gadf_close()
    if (c_associated(ctx)) call gfh_destroy(ctx)
    ctx = c_null_ptr
    if (allocated(fitfuncs)) deallocate(fitfuncs)
    if (allocated(active_pars)) deallocate(active_pars)
    if (allocated(is_global)) deallocate(is_global)
    if (allocated(x_data)) deallocate(x_data)
    if (allocated(y_data)) deallocate(y_data)
    if (allocated(weights)) deallocate(weights)
    if (allocated(data_positions)) deallocate(data_positions)
    if (allocated(data_pointers)) deallocate(data_pointers)
  end subroutine gadf_close
end module gadfit
