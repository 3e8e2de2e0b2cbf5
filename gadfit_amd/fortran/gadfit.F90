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
  use, intrinsic :: iso_fortran_env, only: real32, output_unit, error_unit
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
     type(c_ptr) :: cols = c_null_ptr           ! the parsed file between read_data's two passes (gfh_read_columns)
     logical :: owned = .false.                 ! the three arrays are copies made by gadf_add_dataset (freed by gadf_close)
  end type data_pointer
  ! gadf_add_dataset(x_data, y_data, weights) COPIES its arrays at the call, whatever their size (round 6: one semantic -- the fit is of
  ! the contents the arrays had when they were added; rounds 4-5 copied up to 2^20 points and borrowed above, so the same program
  ! fitted old or new contents depending on N).  The reference borrows by pointer until the first gadf_fit copies (gadfit.F90:241-245,
  ! 417-420) -- but its own programs hand over named constants (fortran/tests/1_gaussian_data.F90 ...), which flang passes as
  ! temporaries that are gone when gadf_fit would read them, and the user guide says of the call that it "reads a data set".  Arrays
  ! beyond COPY_THREADED_FROM points are copied on several threads (fresh pages: 3 x 80 MB at 1e7 points in ~20 ms instead of ~90).
  integer, parameter :: COPY_THREADED_FROM = 2**20

  ! There are as many instances of the fitting function as there are datasets.
  class(fitfunc), allocatable, protected :: fitfuncs(:)
  integer, allocatable :: active_pars(:)          ! 1-based index or 0 (gadfit.F90:68)
  logical, allocatable :: is_global(:)
  real(kp), allocatable, target :: x_data(:), y_data(:), weights(:)
  ! what gfh_set_data reads y and w from: y_data / weights, or -- one dataset handed over in memory -- the user's own arrays
  ! (read_data then copies the abscissas only; the reference copies all three, gadfit.F90:417-420)
  real(kp), pointer, contiguous :: up_y(:) => null(), up_w(:) => null()
  ! The abscissas as the recorder reads them: x_data, or -- during the first gadf_fit of that one in-memory dataset -- the
  ! user's array, while x_data is still being filled by the library's upload thread (gfh_queue_host_copy).
  real(kp), pointer, contiguous :: xs(:) => null()
  logical :: x_copy_pending = .false.
  logical :: copy_in_flight = .false.      ! the library's thread is filling x_data (gfh_queue_host_copy): joined when the fit is over
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
  logical :: model_captured, data_uploaded, lb_on = .false., compile_only = .false.
  ! the parameter values and the active set the model was captured with (a passive parameter's %val may be baked into it)
  real(kp), allocatable :: cap_vals(:,:)
  integer, allocatable :: cap_active(:)
  ! ---- model capture.  eval() is recorded under the recording advar (module ad); one recording follows one path through
  ! eval().  Each distinct path -- same operations, same outcomes of the comparisons of AD variables (guards) -- is kept as a
  ! path_t and handed to the library as one variant tape (include/gadfit_hip.h, gfh_set_model_variants).
  type path_t
     ! the raw recording the path was first met with
     integer :: n = 0, nsub = 0, nint = 0, nip = 0, res_node = -1
     type(gfh_node), allocatable :: raw(:)
     integer, allocatable :: psub(:), cnt(:), sub_result(:)
     type(gfh_integral), allocatable :: pints(:)
     integer, allocatable :: pint_sub(:), pipar(:)
     logical, allocatable :: script(:)             ! outcomes of the comparisons along the path, in the order they are met
     logical :: sub_guards = .false.               ! an integrand of this recording compares AD variables
     real(kp) :: theta = 0.5_kp                    ! ... and where its integration variable sat when the path was first recorded (ad_theta)
     integer :: n_guards = 0, dataset = 1
     ! what the real literals of eval() are as functions of x, learnt from every recording that took this path:
     ! lit_class 0 not a literal, 1 constant lit_c, 2 affine lit_alpha*x + lit_beta, 3 auxiliary per-point column
     integer :: n_seen = 0                         ! distinct abscissas seen: 0, 1, 2 (= two or more)
     real(kp) :: x1 = 0.0_kp, x2 = 0.0_kp
     real(kp), allocatable :: c1(:)
     integer, allocatable :: lit_class(:)
     real(kp), allocatable :: lit_c(:), lit_alpha(:), lit_beta(:)
     logical :: pars_probed = .false., pars_probed2 = .false.
     logical :: theta_probed = .false.
     ! lit_follow(j): a per-point column (lit_class 3) whose values ALSO follow the fitted parameters -- exp(-p%val*x) formed in plain
     ! real arithmetic.  The reference recomputes such a real at every point of every pass (gadfit.F90:679-690); here on_pars tabulates
     ! the column anew on the host whenever the parameters of a pass differ from those of the last tabulation (the slow way, and the
     ! reference's: one eval() per data point and pass) -- refused up to round 4.
     logical, allocatable :: lit_follow(:)
     ! A literal may follow the PARAMETERS (a real formed from a %val): with local parameters it then has one value per dataset.
     ! c_ds(j, d): literal j as first recorded in dataset d; ds_dep(j): its value differs between datasets at one and the same
     ! abscissa (observe); within a dataset, at the parameters of the capture, it must not move
     real(kp), allocatable :: c_ds(:,:)
     logical, allocatable :: ds_seen(:), ds_dep(:)
     ! the tape built from it (kept allocated: the library copies it during gfh_set_model_variants)
     integer :: n_aux = 0, aux0 = 0                ! its auxiliary columns: aux0 .. aux0 + n_aux - 1
     ! lit_class 4: a real that eval() forms from the %val of a FITTED parameter (constant over x, follows the parameters): read on the
     ! device from a passive pseudo-parameter that on_pars refreshes before every pass -- slots plit0 + 1 .. plit0 + n_plit behind the
     ! model's own parameters, plit_raw_k: the literals' raw nodes
     integer :: n_plit = 0, plit0 = 0
     integer, allocatable :: plit_raw_k(:)
     ! ... the slot (0-based, behind the model's own parameters) of the literal at raw node k, -1: none yet.  Outside a fit the slots
     ! are dealt afresh, path by path (upload_model); DURING a fit a slot once dealt stays and new ones are appended, so that the tapes
     ! another member of a device group still runs keep reading the right numbers from the block on_pars fills (ADVICE r4)
     integer, allocatable :: plit_slot(:)
     ! its call sites and sub-tapes in the form the threads' checks read them (load_check_ints; kept while threads run)
     integer(c_int32_t), allocatable :: ki_sub(:), ki_ipar(:), ki_res(:), ki_int(:,:)
     real(c_double), allocatable :: ki_rel(:), ki_abs(:)
     integer, allocatable :: aux_raw_k(:)
     type(gfh_node), allocatable :: final(:)
     type(gfh_subtape_c), allocatable :: sub(:)
     type(gfh_integral), allocatable :: ints(:)
     integer(c_int32_t), allocatable :: ipar(:)
     type(gfh_tape_c) :: tape
  end type path_t
  type(path_t), allocatable, target :: paths(:)
  integer :: n_paths = 0, last_match = 1
  logical :: at_capture_pars = .true.             ! recordings are being made at the parameters the capture began with (not in on_unseen)
  logical :: finite_differences = .false.         ! the gadf_fit in progress was asked for use_ad = .false.
  integer :: n_plit_total = 0                     ! pseudo-parameters of all paths (lit_class 4)
  ! ... and the slots the model's parameter block reserves for them behind the model's own parameters: the number in use when the
  ! model was captured, plus a few spare ones where eval() compares AD variables -- a path first met DURING a fit (on_unseen) may
  ! form such reals of its own, and the block of a fit in progress cannot grow
  integer :: n_plit_cap = 0
  ! columns that follow the parameters (lit_follow): how many, the table of the last tabulation, the parameters it was made at
  ! ([n_pars][n_datasets] as one array), a serial number per tabulation and which library handles hold it
  integer :: n_follow = 0
  real(c_double), allocatable :: tab_keep(:,:)
  real(kp), allocatable :: tab_keep_pars(:)
  integer :: tab_serial = 0, n_up = 0
  type(c_ptr) :: up_tgt(64)
  integer :: up_serial(64)
  integer :: n_retabulated = 0                    ! ... how often on_pars had to do so (GADFIT_HIP_SETUP_TIMES prints it)
  ! use_ad = .false. over such columns: the reference's forward differences call eval() at p + step e_j, where the reals have moved
  ! (fitfunction.F90:155-174) -- the table then holds 1 + n_active SETS of the columns, set 0 at the parameters of the pass, set j at
  ! p + step e_j (gfh_set_fd_column_sets; tabulate_all).  tab_hold: tabulate leaves its table in tab_keep instead of uploading it.
  logical :: tab_hold = .false., cap_fd = .false., accel_requested = .false.
  logical :: refreshing = .false.                 ! tabulate is called from on_pars: the columns at the parameters of a pass, nothing learnt
  ! a threaded tabulation found eval() to answer differently when called concurrently (saved / module state): every later tabulation
  ! of this capture -- the refreshes of on_pars before each pass too -- calls it from one thread only (cleared by the next capture)
  logical :: eval_serial_only = .false.
  ! gadf_init's keywords eval_is_thread_safe / force_outcomes (absent: .false. both = today's defaults)
  logical :: opt_eval_one_thread = .false., opt_no_forced_outcomes = .false., opt_sampled_capture = .false.
  integer, parameter :: PLIT_SPARE = 8
  logical :: fit_in_progress = .false.
  ! cross_check: the outcomes of comparisons (number, bits) every data point has been recorded along so far
  integer :: n_crossed = 0
  integer, allocatable :: crossed_n(:)
  integer(c_int64_t), allocatable :: crossed_bits(:)
  integer :: n_aux_total = 0, hint_col = -1       ! auxiliary literal columns of all paths; the per-point variant column (or -1)
  ! Round 5: one variant column per SET OF OUTCOMES behind the natural one (hint_col: the path a point takes by itself at the parameters
  ! of the tabulation).  cross_q(i, k): the path data point i follows when the outcomes of crossed set k are forced on eval() there --
  ! what cross_check finds anyway, for every point (cross_all); with the comparisons given only plain-real control flow is left, so
  ! the path is a function of the abscissa alone: tabulated once, and a point whose comparison flips during a fit finds its leaf
  ! behind a fork on the device without a report, a recording and a new tabulation (gfh_set_variant_hint_columns).
  ! set_of_col(c): the crossed set behind set column c (aux column hint_col + c); natural_col_used: some path has no comparisons.
  integer(c_int16_t), allocatable :: cross_q(:,:)
  logical :: cross_all = .false., natural_col_used = .true.
  integer :: n_set_cols = 0
  integer, allocatable :: set_of_col(:)
  logical :: need_tab = .false., tabulated = .false.
  integer, parameter :: VERIFY_ALL_UP_TO = 131072
  integer(c_int64_t), allocatable :: slow_i(:)    ! sample points that did not check out against a known path (discover)
  integer, allocatable :: slow_d(:)
  integer :: n_slow = 0

contains

  ! the name of parameter j of a fitting function as plain characters (fitfunc%get_name returns the reference's type(string))
  function par_name(f, j) result(y)
    class(fitfunc), intent(in) :: f
    integer, intent(in) :: j
    character(:), allocatable :: y
    y = ''
    if (allocated(f%parnames)) then
       if (allocated(f%parnames(j)%name)) y = f%parnames(j)%name
    end if
  end function par_name


  subroutine lib_check(rc, file, line)
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: file
    integer, intent(in) :: line
    if (rc /= 0) call error(file, line, c_message(gfh_last_error(ctx)))
  end subroutine lib_check

  ! gadfit.F90:133-184.  The AD / quadrature workspace sizes are accepted for source
  ! compatibility; the tape is recorded once per model, not per point.
  ! The three keywords behind the reference's own arguments (gadfit.F90:133-135) are this layer's (round 6; absent = the defaults; they
  ! state in the program's source what GADFIT_HIP_RECORD_THREADS=1 / GADFIT_HIP_CROSS_CHECK=0 state in its environment):
  !   eval_is_thread_safe = .false.: eval() keeps state in saved or module variables -- it is called from ONE thread only, whatever the
  !     size of the data, as the reference calls it (gadfit.F90:679-690); .true. (or absent): from 1e5 points on from several threads,
  !     every threaded result re-verified (tabulate);
  !   force_outcomes = .false.: eval() is never run along a branch whose own comparison of AD variables is false at that point (no
  !     cross_check: for an eval() that, say, indexes a table by the abscissa behind `if (x < p)`); a fork on the plain real x hidden
  !     behind such a comparison is then not seen before a fit meets it.
  !   record_every_abscissa = .false. (= GADFIT_HIP_VERIFY=sample): eval() is recorded over 2^17 evenly spaced abscissas instead of at
  !     every data point -- the first gadf_fit of 1e7 points 24 ms instead of 0.28 s; ONLY for an eval() that treats x through AD
  !     arithmetic alone: a plain-real feature of eval() between two samples (a window, a table) is then not seen.
  subroutine gadf_init(f, num_datasets, sweep_size, trace_size, const_size, ws_size, &
       & ws_size_inner, integration_rule, ad_memory, rel_error_inner, rel_error, eval_is_thread_safe, force_outcomes, &
       & record_every_abscissa)
    class(fitfunc), intent(in) :: f
    integer, intent(in), optional :: num_datasets, sweep_size, trace_size, const_size, &
         & ws_size, ws_size_inner, integration_rule
    character(*), intent(in), optional :: ad_memory
    real(kp), intent(in), optional :: rel_error_inner, rel_error
    logical, intent(in), optional :: eval_is_thread_safe, force_outcomes, record_every_abscissa
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
    if (allocated(cap_vals)) deallocate(cap_vals, cap_active)
    n_paths = 0; need_tab = .false.; tabulated = .false.; hint_col = -1; n_aux_total = 0
    n_follow = 0; n_up = 0; eval_serial_only = .false.
    opt_eval_one_thread = .false.; opt_no_forced_outcomes = .false.; opt_sampled_capture = .false.
    if (present(record_every_abscissa)) opt_sampled_capture = .not. record_every_abscissa
    if (present(eval_is_thread_safe)) opt_eval_one_thread = .not. eval_is_thread_safe
    if (present(force_outcomes)) opt_no_forced_outcomes = .not. force_outcomes
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
    compile_only = device < 0 .and. n_group == 0
    if (n_group /= 0) then
       call lib_check(gfh_create_group(int(max(n_group, 0), c_int), c_null_ptr, ctx), __FILE__, __LINE__)
    else
       ! (the HIP runtime starts up on a thread of the library -- 80 ms per process, 240 ms for the first one on a machine -- beside
       ! gadf_add_dataset / gadf_set and the recording of eval() in the first gadf_fit; whoever needs the device first waits for it)
       call lib_check(gfh_create_begin(int(device, c_int), ctx), __FILE__, __LINE__)
    end if
    ! The Jacobian has no reader behind this API (JacobianT is private in the reference, gadfit.F90:60-64): the fused
    ! STEP 1+2 kernel writes it only for the fits whose options read it back (gfh_set_keep_jacobian mode 2; same
    ! J^T J / J^T r / chi2 bit for bit).  GADFIT_HIP_KEEP_J overrides.
    call get_environment_variable('GADFIT_HIP_KEEP_J', env, status=stat)
    if (stat /= 0 .and. (device >= 0 .or. n_group /= 0)) call lib_check(gfh_set_keep_jacobian(ctx, 2_c_int), __FILE__, __LINE__)
    ! one process per GPU, started by a plain shell loop: GADFIT_HIP_NRANKS / _RANK / _IDFILE
    ! (replaces num_images()/this_image(); no-op when unset)
    if (device >= 0) call lib_check(gfh_comm_init_from_env(ctx), __FILE__, __LINE__)
    ! branches of eval() the recordings have not seen are reported by the device and recorded by on_unseen
    call lib_check(gfh_set_unseen_handler(ctx, c_funloc(on_unseen), c_null_ptr), __FILE__, __LINE__)
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

  ! gadfit.F90:226-246 (there: borrowed by pointer until the first gadf_fit; here: copied at the call, see COPY_THREADED_FROM)
  subroutine gadf_add_dataset_data(x_data, y_data, weights)
    !$ use omp_lib, only: omp_get_max_threads
    real(kp), intent(in), target :: x_data(:), y_data(:)
    real(kp), intent(in), target, optional :: weights(:)
    if (.not. allocated(fitfuncs)) call error(__FILE__, __LINE__, &
         & 'Number of datasets is undetermined. Call gadf_init first.')
    if (n_added >= size(fitfuncs)) call error(__FILE__, __LINE__, 'Too many calls to gadf_add_dataset.')
    n_added = n_added + 1
    call copy_in(data_pointers(n_added)%x_data, x_data)
    call copy_in(data_pointers(n_added)%y_data, y_data)
    if (present(weights)) call copy_in(data_pointers(n_added)%weights, weights)
    data_pointers(n_added)%owned = .true.
  contains
    subroutine copy_in(dst, src)
      real(kp), pointer, intent(out) :: dst(:)
      real(kp), intent(in) :: src(:)
      integer(c_int64_t) :: n, lo, k, nchunk
      integer(c_int64_t), parameter :: chunk = 262144
      integer :: nthreads
      n = size(src, kind=c_int64_t)
      if (n <= COPY_THREADED_FROM) then
         allocate(dst, source=src)
         return
      end if
      allocate(dst(n))
      nchunk = (n + chunk - 1)/chunk
      call omp_defaults()
      nthreads = 1
      !$ nthreads = max(1, min(16, recorder_threads_max(), omp_get_max_threads()))
      !$omp parallel do schedule(static) num_threads(nthreads) private(lo)
      do k = 1, nchunk
         lo = (k - 1)*chunk + 1
         dst(lo:min(n, lo + chunk - 1)) = src(lo:min(n, lo + chunk - 1))
      end do
      !$omp end parallel do
    end subroutine copy_in
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

  ! Before the first parallel region of the process.  The OpenMP runtime's affinity set-up walks the machine's topology when it
  ! starts -- 20 ms on a 256-thread host, twice the parallel loops it is started for; a few short-lived threads need no binding.
  ! Left alone if the user has asked for one.  (Only before the runtime starts: later calls are refused with a warning.)
  ! Threads that call eval() at once over a large data set: the host's CPU budget (affinity mask, cut down to the cgroup's quota:
  ! gfh_host_cpu_budget, ad_tls.c) less two -- the library's upload thread and the runtime's own are busy beside them, and a process
  ! over its quota is throttled as a whole --, between 1 and 32.
  ! eval() is called from several threads at once only for data sets of at least this many points (GADFIT_HIP_THREADS_FROM; default
  ! 100000).  The reference never calls eval() concurrently (gadfit.F90:679-690, 1023-1027: one image, one point at a time), so an
  ! eval() that keeps state in saved or module variables is legal there; below the threshold the serial recorder costs at most
  ! ~35 ms and such an eval() simply works.  Above it the capture would take seconds on one thread: eval() is then called
  ! concurrently, its per-point columns are made twice and re-verified serially at a sample (tabulate), and
  ! GADFIT_HIP_RECORD_THREADS=1 is the reference-faithful setting for an eval() that is not thread-safe (INTEGRATION.md).
  integer function threads_from() result(n)
    character(len=24) :: e
    integer :: stat
    n = 100000
    call get_environment_variable('GADFIT_HIP_THREADS_FROM', e, status=stat)
    if (stat == 0) then
       read(e, *, iostat=stat) n
       if (stat /= 0 .or. n < 1) n = 100000
    end if
  end function threads_from

  integer function recorder_threads_max() result(n)
    n = max(1, min(32, int(gfh_host_cpu_budget()) - 2))
  end function recorder_threads_max

  subroutine omp_defaults()
    !$ use omp_lib, only: kmp_set_defaults
    logical, save :: affinity_set = .false.
    character(len=16) :: envt
    integer :: stat
    if (affinity_set) return
    affinity_set = .true.
    stat = 1
    !$ call get_environment_variable('KMP_AFFINITY', envt, status=stat)
    !$ if (stat /= 0) call get_environment_variable('OMP_PROC_BIND', envt, status=stat)
    !$ if (stat /= 0) call get_environment_variable('OMP_PLACES', envt, status=stat)
    !$ if (stat /= 0) call kmp_set_defaults('KMP_AFFINITY=disabled')
  end subroutine omp_defaults

  ! read_data (gadfit.F90:401-443): concatenates all datasets; for USER the third column /
  ! weights argument holds the uncertainties, which init_weights inverts ON THE DEVICE.
  subroutine read_data()
    !$ use omp_lib, only: omp_get_max_threads
    integer :: i, n, k, nchunk, nthreads, n_files
    integer(c_int64_t) :: j, nfile
    integer(c_int64_t), parameter :: chunk = 262144
    integer, allocatable :: chunk_ds(:)
    integer(c_int64_t), allocatable :: chunk_lo(:), file_n(:)
    integer(c_int), allocatable :: file_rc(:)
    if (n_added /= size(fitfuncs)) call error(__FILE__, __LINE__, &
         & 'Some datasets are missing. gadf_add_dataset must be called for every dataset.')
    ! gadfit.F90:212-215, 422-437: the records that begin with a number, their first two (USER errors: three) numbers.  The
    ! library parses a file once, large ones on several threads (reader.cpp: flang's list-directed reads take 2.5 s per million
    ! lines, twice), and the files of a many-curve fit side by side; the columns are taken over below
    allocate(file_n(size(fitfuncs)), file_rc(size(fitfuncs)))
    file_n = 0; file_rc = 0
    n_files = count([(.not. associated(data_pointers(i)%x_data), i = 1, size(fitfuncs))])
    if (n_files > 1) then
       call omp_defaults()
       nthreads = 1
       !$ nthreads = max(1, min(16, omp_get_max_threads(), n_files))
       !$omp parallel do schedule(dynamic) num_threads(nthreads)
       do i = 1, size(fitfuncs)
          if (.not. associated(data_pointers(i)%x_data)) call read_file(i)
       end do
       !$omp end parallel do
    else if (n_files == 1) then         ! (no parallel region, which would start the OpenMP runtime for nothing)
       do i = 1, size(fitfuncs)
          if (.not. associated(data_pointers(i)%x_data)) call read_file(i)
       end do
    end if
    data_positions(1) = 0
    do i = 1, size(fitfuncs)
       if (associated(data_pointers(i)%x_data)) then
          n = size(data_pointers(i)%x_data)
       else
          ! (a file that failed is read once more, alone, for its message: the library keeps the last one of the process)
          if (file_rc(i) /= 0) then
             if (gfh_read_columns(data_pointers(i)%path//c_null_char, int(merge(3, 2, data_error_type == USER), c_int), &
                  & data_pointers(i)%cols, nfile) /= 0) call error(__FILE__, __LINE__, c_message(gfh_last_error(c_null_ptr)))
             file_n(i) = nfile
          end if
          n = int(file_n(i))
          if (n == 0) call error(__FILE__, __LINE__, data_pointers(i)%path//' contains no valid data points.')
       end if
       data_positions(i+1) = data_positions(i) + n
    end do
    n = int(data_positions(size(fitfuncs)+1))
    ! One dataset, in memory, contiguous: only x is copied (the recorder and gadf_print read it later); y and the
    ! uncertainties go to the device straight from the user's arrays during this first gadf_fit -- 160 MB less to copy at
    ! N = 1e7.  (Without USER errors the device computes the weights from y: the host array only has to exist.)
    if (size(fitfuncs) == 1 .and. associated(data_pointers(1)%x_data)) then
       if (is_contiguous(data_pointers(1)%y_data)) then
          if (data_error_type /= USER) then
             allocate(x_data(n))
             call borrow_x()
             up_y => data_pointers(1)%y_data; up_w => data_pointers(1)%y_data
             return
          else if (associated(data_pointers(1)%weights)) then
             if (is_contiguous(data_pointers(1)%weights)) then
                allocate(x_data(n))
                call borrow_x()
                up_y => data_pointers(1)%y_data; up_w => data_pointers(1)%weights
                return
             end if
          end if
       end if
    end if
    allocate(x_data(n), y_data(n), weights(n))
    up_y => y_data; up_w => weights; xs => x_data; x_copy_pending = .false.
    do i = 1, size(fitfuncs)
       if (associated(data_pointers(i)%x_data) .and. data_error_type == USER .and. .not. associated(data_pointers(i)%weights)) &
            & call error(__FILE__, __LINE__, 'USER errors requested but no weights were given.')
    end do
    ! the datasets side by side in one array each (gadfit.F90:417-420).  In pieces of 2^18 points on several threads: the arrays
    ! are fresh pages, and one thread faulting them in costs 27 ms for the 64 x 1e5 points of BASELINE config 3 (ten LM iterations
    ! of that fit take 1 ms)
    nchunk = 0
    do i = 1, size(fitfuncs)
       if (associated(data_pointers(i)%x_data)) nchunk = nchunk + int((data_positions(i+1) - data_positions(i) + chunk - 1)/chunk)
    end do
    allocate(chunk_ds(nchunk), chunk_lo(nchunk))
    nchunk = 0
    do i = 1, size(fitfuncs)
       if (.not. associated(data_pointers(i)%x_data)) cycle
       do j = 0, data_positions(i+1) - data_positions(i) - 1, chunk
          nchunk = nchunk + 1; chunk_ds(nchunk) = i; chunk_lo(nchunk) = j
       end do
    end do
    if (nchunk > 3) then
       call omp_defaults()
       nthreads = 1
       !$ nthreads = max(1, min(16, omp_get_max_threads(), nchunk))
       !$omp parallel do schedule(dynamic) num_threads(nthreads)
       do k = 1, nchunk
          call copy_chunk(k)
       end do
       !$omp end parallel do
    else                                 ! (small data: not worth starting the OpenMP runtime for)
       do k = 1, nchunk
          call copy_chunk(k)
       end do
    end if
    do i = 1, size(fitfuncs)
       if (associated(data_pointers(i)%x_data)) cycle
       j = data_positions(i)
       if (data_error_type /= USER) weights(j+1:data_positions(i+1)) = 1.0_kp
       if (gfh_take_columns(data_pointers(i)%cols, x_data(j+1:data_positions(i+1)), y_data(j+1:data_positions(i+1)), &
            & weights(j+1:data_positions(i+1))) /= 0) call error(__FILE__, __LINE__, c_message(gfh_last_error(c_null_ptr)))
       data_pointers(i)%cols = c_null_ptr                ! (taken over and freed)
    end do
  contains
    subroutine copy_chunk(k)
      integer, intent(in) :: k
      integer :: i
      integer(c_int64_t) :: j, lo, hi
      i = chunk_ds(k); lo = chunk_lo(k) + 1
      hi = min(chunk_lo(k) + chunk, data_positions(i+1) - data_positions(i))
      j = data_positions(i)
      x_data(j+lo:j+hi) = data_pointers(i)%x_data(lo:hi)
      y_data(j+lo:j+hi) = data_pointers(i)%y_data(lo:hi)
      if (data_error_type == USER) then
         weights(j+lo:j+hi) = data_pointers(i)%weights(lo:hi)
      else
         weights(j+lo:j+hi) = 1.0_kp
      end if
    end subroutine copy_chunk

    subroutine read_file(i)
      integer, intent(in) :: i
      if (c_associated(data_pointers(i)%cols)) call gfh_free_columns(data_pointers(i)%cols)
      data_pointers(i)%cols = c_null_ptr
      file_rc(i) = gfh_read_columns(data_pointers(i)%path//c_null_char, int(merge(3, 2, data_error_type == USER), c_int), &
           & data_pointers(i)%cols, file_n(i))
    end subroutine read_file

    ! x_data exists but is filled later (own_x): until then the user's abscissas are read in place
    subroutine borrow_x()
      if (is_contiguous(data_pointers(1)%x_data)) then
         xs => data_pointers(1)%x_data; x_copy_pending = .true.
      else
         x_data = data_pointers(1)%x_data; xs => x_data; x_copy_pending = .false.
      end if
    end subroutine borrow_x
  end subroutine read_data

  logical function x_on_device_or_borrowed() result(b)
    b = x_copy_pending .or. copy_in_flight
  end function x_on_device_or_borrowed

  ! x_data filled from the user's array now, if that is still outstanding
  subroutine own_x()
    if (copy_in_flight) then
       call lib_check(gfh_wait_host_copy(ctx), __FILE__, __LINE__)
    else if (x_copy_pending) then
       x_data = xs
    end if
    xs => x_data; x_copy_pending = .false.; copy_in_flight = .false.
  end subroutine own_x

  ! ---------------------------------------------------------------- model capture
  ! One recording of eval() for dataset d at abscissa x.  The first ns comparisons of AD variables take the outcomes of
  ! `script` (ns = 0: every comparison is decided by the values).  The recording is left in module ad (ad_tape ...).
  subroutine record(d, x, ns, script, res)
    integer, intent(in) :: d, ns
    real(kp), intent(in) :: x
    logical, intent(in) :: script(:)
    integer, intent(out) :: res
    type(advar) :: y
    integer :: k
    call ad_set_script(ns, script)
    call ad_capture_begin()
    do k = 1, size(fitfuncs(d)%pars)
       call set_node(fitfuncs(d)%pars(k), ad_emit(GFH_PARAM, k-1, -1, 0, 0.0_kp))
    end do
    y = fitfuncs(d)%eval(x)
    res = anode(y)
    call ad_capture_end()
    call ad_set_script(0)
    do k = 1, size(fitfuncs(d)%pars)
       call set_node(fitfuncs(d)%pars(k), -1)
    end do
    if (ad_capture_failed) call error(__FILE__, __LINE__, trim(ad_capture_msg))
  end subroutine record

  ! Is the recording in module ad the path p (same operations, operands, comparison outcomes, integrate() call sites)?
  logical function same_as(p, res) result(same)
    type(path_t), intent(in) :: p
    integer, intent(in) :: res
    integer :: j
    same = .false.
    if (ad_tape_n /= p%n .or. ad_nsub /= p%nsub .or. ad_n_integrals /= p%nint .or. ad_n_ipar /= p%nip .or. res /= p%res_node) return
    do j = 1, p%n
       if (ad_tape(j)%op /= p%raw(j)%op .or. ad_tape(j)%a /= p%raw(j)%a .or. ad_tape(j)%b /= p%raw(j)%b .or. &
            & ad_tape(j)%flags /= p%raw(j)%flags .or. ad_sub(j) /= p%psub(j)) return
    end do
    if (p%nip > 0) then
       if (any(ad_ipar_nodes(:p%nip) /= p%pipar(:p%nip))) return
    end if
    do j = 1, p%nint
       if (ad_integrals(j)%integrand /= p%pints(j)%integrand .or. ad_integrals(j)%lower /= p%pints(j)%lower .or. &
            & ad_integrals(j)%upper /= p%pints(j)%upper .or. ad_integrals(j)%lower_inf /= p%pints(j)%lower_inf .or. &
            & ad_integrals(j)%upper_inf /= p%pints(j)%upper_inf .or. ad_integrals(j)%n_ipars /= p%pints(j)%n_ipars .or. &
            & ad_integrals(j)%rel_error /= p%pints(j)%rel_error .or. ad_integrals(j)%abs_error /= p%pints(j)%abs_error) return
    end do
    same = .true.
  end function same_as

  ! one recording of eval() on the calling thread in the recorder's thread-checking mode (parameter nodes preset, values skipped)
  subroutine check_one(d, x, np_, cn, cdiv, clit, res)
    integer, intent(in) :: d, np_
    real(kp), intent(in) :: x
    integer(c_int), intent(out) :: cn, cdiv, clit
    integer, intent(out) :: res
    type(advar) :: y
    call gfh_adchk_begin(x, int(np_, c_int))
    y = fitfuncs(d)%eval(x)
    res = anode(y)
    call gfh_adchk_end(cn, cdiv, clit)
  end subroutine check_one

  ! a recording for dataset d whose parameter nodes were set beforehand
  subroutine record_preset(d, x, res)
    integer, intent(in) :: d
    real(kp), intent(in) :: x
    integer, intent(out) :: res
    type(advar) :: y
    call ad_set_script(0)
    call ad_capture_begin()
    call ad_emit_params(size(fitfuncs(d)%pars))
    y = fitfuncs(d)%eval(x)
    res = anode(y)
    call ad_capture_end()
  end subroutine record_preset

  ! are the literals of the recording in module ad (which follows path p) what p's classification says, at abscissa x?
  logical function literals_as_known(p, x, d) result(ok)
    type(path_t), intent(in) :: p
    real(kp), intent(in) :: x
    integer, intent(in) :: d
    integer :: j
    real(kp) :: c, want
    ok = .false.
    do j = 1, p%n
       if (p%raw(j)%op /= GFH_CONST) cycle
       c = ad_tape(j)%c
       if (p%lit_class(j) == 1) then
          if (c /= p%lit_c(j) .and. .not. (c /= c .and. p%lit_c(j) /= p%lit_c(j))) return
       else if (p%lit_class(j) == 2) then
          want = p%lit_alpha(j)*x + p%lit_beta(j)
          if (.not. (abs(want - c) <= 1e-11_kp*(abs(c) + abs(p%lit_alpha(j)*x) + abs(p%lit_beta(j))))) return
       else if (p%lit_class(j) == 4) then         ! (constant within a dataset at the parameters of the capture: observe)
          if (.not. p%ds_seen(d)) return
          if (c /= p%c_ds(j, d) .and. .not. (c /= c .and. p%c_ds(j, d) /= p%c_ds(j, d))) return
       end if
    end do
    ok = .true.
  end function literals_as_known

  subroutine push_slow(d, i)
    integer, intent(in) :: d
    integer(c_int64_t), intent(in) :: i
    integer(c_int64_t), allocatable :: ti(:)
    integer, allocatable :: td(:)
    if (n_slow == size(slow_i)) then
       allocate(ti(2*n_slow), td(2*n_slow))
       ti(:n_slow) = slow_i(:n_slow); td(:n_slow) = slow_d(:n_slow)
       call move_alloc(ti, slow_i); call move_alloc(td, slow_d)
    end if
    n_slow = n_slow + 1
    slow_i(n_slow) = i; slow_d(n_slow) = d
  end subroutine push_slow

  ! path p and what is known of its literals into module ad's checking arrays
  subroutine load_check(p, d)
    type(path_t), intent(in) :: p
    integer, intent(in) :: d                      ! the dataset whose recordings are about to be checked
    ad_chk_n = p%n
    ad_chk_op = p%raw%op; ad_chk_a = p%raw%a; ad_chk_b = p%raw%b; ad_chk_fl = p%raw%flags; ad_chk_sub = p%psub
    ad_chk_cls = p%lit_class; ad_chk_c = p%lit_c; ad_chk_alpha = p%lit_alpha; ad_chk_beta = p%lit_beta
    ! (a literal that follows the parameters is a constant of this dataset while the capture runs: checked as one, so that a
    ! recording where it moves with the abscissa does not check out and reaches observe)
    if (p%ds_seen(d) .and. at_capture_pars) then
       where (p%lit_class == 4)
          ad_chk_cls = 1; ad_chk_c = p%c_ds(:, d)
       end where
    end if
  end subroutine load_check

  ! the call sites and sub-tapes of path p for the threads' checks (known recording k of ad_tls.c; the arrays stay put while threads run)
  subroutine load_check_ints(k, p)
    integer, intent(in) :: k
    type(path_t), intent(in out) :: p
    integer :: i
    if (allocated(p%ki_sub)) deallocate(p%ki_sub, p%ki_ipar, p%ki_res, p%ki_int, p%ki_rel, p%ki_abs)
    allocate(p%ki_sub(p%n), p%ki_ipar(max(1, p%nip)), p%ki_res(0:p%nsub), p%ki_int(max(1, p%nint), 6), p%ki_rel(max(1, p%nint)), p%ki_abs(max(1, p%nint)))
    p%ki_sub = p%psub(:p%n)
    if (p%nip > 0) p%ki_ipar(:p%nip) = p%pipar(:p%nip)
    p%ki_res(0:) = p%sub_result(0:p%nsub)
    do i = 1, p%nint
       p%ki_int(i, :) = [p%pints(i)%integrand, p%pints(i)%lower, p%pints(i)%upper, p%pints(i)%lower_inf, p%pints(i)%upper_inf, p%pints(i)%n_ipars]
       p%ki_rel(i) = p%pints(i)%rel_error; p%ki_abs(i) = p%pints(i)%abs_error
    end do
    call gfh_adchk_load_ints(int(k, c_int), int(p%nsub, c_int), int(p%nint, c_int), int(p%nip, c_int), p%ki_sub, p%ki_ipar, p%ki_res, &
         & p%ki_int(:, 1), p%ki_int(:, 2), p%ki_int(:, 3), p%ki_int(:, 4), p%ki_int(:, 5), p%ki_int(:, 6), p%ki_rel, p%ki_abs)
  end subroutine load_check_ints

  ! after a recording made in checking mode that did not diverge: what the node-by-node comparison does not cover
  logical function checked_same(p, res) result(same)
    type(path_t), intent(in) :: p
    integer, intent(in) :: res
    integer :: j
    same = .false.
    if (ad_tape_n /= p%n .or. ad_nsub /= p%nsub .or. ad_n_integrals /= p%nint .or. ad_n_ipar /= p%nip .or. res /= p%res_node) return
    if (p%nip > 0) then
       if (any(ad_ipar_nodes(:p%nip) /= p%pipar(:p%nip))) return
    end if
    do j = 1, p%nint
       if (ad_integrals(j)%integrand /= p%pints(j)%integrand .or. ad_integrals(j)%lower /= p%pints(j)%lower .or. &
            & ad_integrals(j)%upper /= p%pints(j)%upper .or. ad_integrals(j)%lower_inf /= p%pints(j)%lower_inf .or. &
            & ad_integrals(j)%upper_inf /= p%pints(j)%upper_inf .or. ad_integrals(j)%n_ipars /= p%pints(j)%n_ipars .or. &
            & ad_integrals(j)%rel_error /= p%pints(j)%rel_error .or. ad_integrals(j)%abs_error /= p%pints(j)%abs_error) return
    end do
    same = .true.
  end function checked_same

  ! index of the known path the recording in module ad follows, or 0
  integer function find_path(res) result(q)
    integer, intent(in) :: res
    integer :: k
    q = 0
    if (n_paths == 0) return
    if (last_match >= 1 .and. last_match <= n_paths) then
       if (same_as(paths(last_match), res)) then
          q = last_match; return
       end if
    end if
    do k = 1, n_paths
       if (k == last_match) cycle
       if (same_as(paths(k), res)) then
          q = k; last_match = k; return
       end if
    end do
  end function find_path

  ! the recording in module ad becomes a new path (first met in dataset d)
  subroutine add_path(d, res)
    integer, intent(in) :: d, res
    type(path_t), allocatable :: tmp(:)
    type(path_t) :: blank
    integer :: n, j, g
    if (.not. allocated(paths)) allocate(paths(8))
    if (n_paths == size(paths)) then
       allocate(tmp(2*size(paths)))
       tmp(:n_paths) = paths(:n_paths)
       call move_alloc(tmp, paths)
    end if
    n_paths = n_paths + 1
    last_match = n_paths
    paths(n_paths) = blank                         ! (a slot left over from an earlier model: components released)
    associate(p => paths(n_paths))
      n = ad_tape_n
      p%n = n; p%nsub = ad_nsub; p%nint = ad_n_integrals; p%nip = ad_n_ipar; p%res_node = res; p%dataset = d
      p%raw = ad_tape(:n); p%psub = ad_sub(:n)
      allocate(p%cnt(0:ad_nsub), p%sub_result(0:ad_nsub))      ! (0-based like ad_sub_n: sub-tape 0 is eval())
      p%cnt(0:) = ad_sub_n(0:ad_nsub); p%sub_result(0:) = ad_sub_result(0:ad_nsub)
      allocate(p%pints(max(1, p%nint)), p%pint_sub(max(1, p%nint)), p%pipar(max(1, p%nip)))
      if (p%nint > 0) then
         p%pints(:p%nint) = ad_integrals(:p%nint); p%pint_sub(:p%nint) = ad_int_sub(:p%nint)
      end if
      if (p%nip > 0) p%pipar(:p%nip) = ad_ipar_nodes(:p%nip)
      ! (the comparisons of eval() itself: those inside integrands are decided per evaluation on the device and never forced)
      g = count((p%raw%op == GFH_GUARD_GT .or. p%raw%op == GFH_GUARD_LT) .and. p%psub == 0)
      p%n_guards = g
      p%sub_guards = any((p%raw%op == GFH_GUARD_GT .or. p%raw%op == GFH_GUARD_LT) .and. p%psub /= 0)
      p%theta = ad_theta
      allocate(p%script(max(1, g)))
      g = 0
      do j = 1, n
         if ((p%raw(j)%op == GFH_GUARD_GT .or. p%raw(j)%op == GFH_GUARD_LT) .and. p%psub(j) == 0) then
            g = g + 1
            p%script(g) = iand(p%raw(j)%flags, GFH_F_TAKEN) /= 0
         end if
      end do
      if (g > 64) call error(__FILE__, __LINE__, 'eval() makes more than 64 comparisons of AD variables on one path.')
      allocate(p%c1(n), p%lit_class(n), p%lit_c(n), p%lit_alpha(n), p%lit_beta(n), p%plit_slot(n), p%lit_follow(n))
      p%lit_follow = .false.; p%pars_probed2 = .false.
      p%lit_class = 0; p%lit_c = 0.0_kp; p%lit_alpha = 0.0_kp; p%lit_beta = 0.0_kp; p%c1 = 0.0_kp; p%plit_slot = -1
      p%n_seen = 0; p%pars_probed = .false.; p%theta_probed = .false.; p%n_aux = 0; p%aux0 = 0
      allocate(p%c_ds(n, size(fitfuncs)), p%ds_seen(size(fitfuncs)), p%ds_dep(n))
      p%c_ds = 0.0_kp; p%ds_seen = .false.; p%ds_dep = .false.
    end associate
  end subroutine add_path

  ! The recording in module ad took path p at abscissa x: learn what its real literals are.  A real(kp) value that eval()
  ! forms in plain real arithmetic reaches the recorder as a literal operand.  Over the recordings of one path a literal is
  ! either the same everywhere (a constant), or affine in x (x itself, -x, x - c, c*x: how a real abscissa enters advar
  ! arithmetic), or neither (x**2, exp(-x), a window ...): then it is an auxiliary per-point column, tabulated by the host.
  ! Literals inside integrands must not depend on x (x reaches an integrand through its pars(:), as in the reference's examples).
  subroutine observe(p, x, d)
    type(path_t), intent(in out) :: p
    real(kp), intent(in) :: x
    integer, intent(in) :: d                      ! the dataset the recording was made for
    integer :: j, res2
    real(kp) :: c, want, alpha, beta, scale
    real(kp), allocatable :: cv(:)
    logical :: refit, moved
    cv = ad_tape(1:p%n)%c                         ! (this recording's literals: the probes below record again)
    if (p%n_seen == 0) then
       p%x1 = x; p%n_seen = 1
       do j = 1, p%n
          if (p%raw(j)%op /= GFH_CONST) cycle
          p%c1(j) = cv(j); p%lit_class(j) = 1; p%lit_c(j) = cv(j)
       end do
       p%c_ds(:, d) = cv; p%ds_seen(d) = .true.
       return
    end if
    if (.not. p%ds_seen(d)) then
       ! The path's first recording in THIS dataset.  Constants that come out differently from the dataset it was first met in: is it
       ! the dataset -- a real formed from the %val of a local parameter, which has another value here -- or the abscissa?  eval() of
       ! this dataset is recorded once more at the path's FIRST abscissa (its comparisons forced): what differs there follows the dataset.
       p%c_ds(:, d) = cv
       moved = .false.
       do j = 1, p%n
          if (p%raw(j)%op /= GFH_CONST .or. (p%lit_class(j) /= 1 .and. p%lit_class(j) /= 4)) cycle
          if (cv(j) /= p%lit_c(j) .and. .not. (cv(j) /= cv(j) .and. p%lit_c(j) /= p%lit_c(j))) moved = .true.
       end do
       if (moved .and. x /= p%x1) then
          ad_theta = p%theta
          call record(d, p%x1, p%n_guards, p%script, res2)
          ad_theta = 0.5_kp
          if (same_as(p, res2)) then
             do j = 1, p%n
                if (p%raw(j)%op /= GFH_CONST .or. (p%lit_class(j) /= 1 .and. p%lit_class(j) /= 4)) cycle
                if (ad_tape(j)%c /= p%lit_c(j) .and. .not. (ad_tape(j)%c /= ad_tape(j)%c .and. p%lit_c(j) /= p%lit_c(j))) then
                   p%ds_dep(j) = .true.; p%c_ds(j, d) = ad_tape(j)%c
                end if
             end do
          end if
       else if (moved) then                       ! (recorded at the first abscissa itself: the comparison is at hand)
          do j = 1, p%n
             if (p%raw(j)%op /= GFH_CONST .or. (p%lit_class(j) /= 1 .and. p%lit_class(j) /= 4)) cycle
             if (cv(j) /= p%lit_c(j) .and. .not. (cv(j) /= cv(j) .and. p%lit_c(j) /= p%lit_c(j))) p%ds_dep(j) = .true.
          end do
       end if
       p%ds_seen(d) = .true.
       ! which of them follow FITTED parameters (lit_class 4: pseudo-parameters that on_pars refreshes); the others follow passive
       ! ones, which never move during a fit: a per-point column carries their values dataset by dataset
       if (any(p%ds_dep .and. p%lit_class == 1)) then
          call probe_pars(p, again=.true.)
          where (p%ds_dep .and. p%lit_class == 1) p%lit_class = 3
       end if
    end if
    if (p%n_seen == 1 .and. x == p%x1) return
    refit = .false.
    if (p%n_seen == 1) then
       p%x2 = x; p%n_seen = 2; refit = .true.
    else if (abs(x - p%x1) > 4.0_kp*abs(p%x2 - p%x1)) then
       refit = .true.            ! a longer baseline: slopes of affine literals are taken again from it (less rounding in them)
    end if
    do j = 1, p%n
       if (p%raw(j)%op /= GFH_CONST) cycle
       c = cv(j)
       select case (p%lit_class(j))
       case (4)
          ! follows the parameters: within one dataset, at the parameters of the capture, it must not move with the abscissa
          ! ... where it does, it becomes a per-point column that on_pars tabulates anew at the parameters of every pass
          if (at_capture_pars .and. c /= p%c_ds(j, d) .and. .not. (c /= c .and. p%c_ds(j, d) /= p%c_ds(j, d))) then
             if (p%psub(j) /= 0) call error(__FILE__, __LINE__, 'An integrand forms a real number from &
                  &parameter values (%val) AND the abscissa; such a literal cannot follow the parameters on the device. Keep it as advar.')
             p%lit_class(j) = 3; p%lit_follow(j) = .true.
          end if
       case (1)
          if (c == p%lit_c(j) .or. (c /= c .and. p%lit_c(j) /= p%lit_c(j))) cycle
          ! (inside an integrand too: a real that the integrand takes from the enclosing eval() -- a module variable carrying x past
          ! pars(:) -- is classified like eval()'s own literals; one that follows the %val of the integration variable is caught by
          ! probe_theta.  Recordings that differ only in an integrand's path are pooled on the device and share their columns: not
          ! with per-point literals inside those integrands.)
          if (p%psub(j) /= 0 .and. p%sub_guards) call error(__FILE__, __LINE__, 'A real literal inside an integrand that compares AD &
               &variables depends on x: pass x to such an integrand through its pars(:) array.')
          if (p%n_seen == 2 .and. refit .and. x == p%x2) then      ! the second abscissa: a first slope
             call fit_affine(p%x1, p%c1(j), x, c, alpha, beta)
             p%lit_class(j) = 2; p%lit_alpha(j) = alpha; p%lit_beta(j) = beta
          else
             p%lit_class(j) = 3                                     ! constant over the abscissas so far, not here
          end if
       ! (class 4 follows the PARAMETERS: recordings made at the parameters of a later pass -- on_unseen -- legitimately carry another
       ! value; that it does not move with x was established over the data at the parameters of the capture, before probe_pars)
       case (2)
          want = p%lit_alpha(j)*x + p%lit_beta(j)
          scale = abs(c) + abs(p%lit_alpha(j)*x) + abs(p%lit_beta(j))
          if (.not. (abs(want - c) <= 1e-11_kp*scale)) then
             p%lit_class(j) = 3
          else if (refit) then
             call fit_affine(p%x1, p%c1(j), x, c, alpha, beta)
             p%lit_alpha(j) = alpha; p%lit_beta(j) = beta
          end if
       end select
    end do
    if (refit) p%x2 = x
  contains
    subroutine fit_affine(xa, ca, xb, cb, al, be)
      real(kp), intent(in) :: xa, ca, xb, cb
      real(kp), intent(out) :: al, be
      real(kp) :: sc
      al = (cb - ca)/(xb - xa)
      if (abs(al - 1.0_kp) < 1e-13_kp) al = 1.0_kp
      if (abs(al + 1.0_kp) < 1e-13_kp) al = -1.0_kp
      be = ca - al*xa
      sc = abs(ca) + abs(al*xa)
      if (abs(be) <= 1e-13_kp*sc) be = 0.0_kp
    end subroutine fit_affine
  end subroutine observe

  ! A literal that follows the PARAMETERS (eval() reading %val into plain real arithmetic) cannot follow them on the device:
  ! the path is recorded once more at its first abscissa with perturbed parameter values (its comparisons forced to their
  ! recorded outcomes, so that only the values move) and every literal must come out the same.
  ! the sub-tape from which integrand sub-tape s of path p is integrated (0: eval() itself; -1: not an integrand of this path)
  integer function caller_of(p, s) result(c)
    type(path_t), intent(in) :: p
    integer, intent(in) :: s
    integer :: i
    c = -1
    do i = 1, p%nint
       if (p%pints(i)%integrand == s) then
          c = p%pint_sub(i); return
       end if
    end do
  end function caller_of

  subroutine probe_pars(p, again)
    type(path_t), intent(in out) :: p
    logical, intent(in), optional :: again        ! (observe, while the classes are still being learnt: the probe of the finished path follows)
    real(kp), allocatable :: saved(:), base(:)
    integer :: res
    logical :: first
    first = .true.
    if (present(again)) then
       if (.not. again .and. p%pars_probed) return
    else
       ! (once at the path's first abscissa and, as soon as a second one is known, once there: a real like p%val*x is zero at x = 0
       ! whatever the parameter)
       if (p%pars_probed .and. (p%pars_probed2 .or. p%n_seen < 2 .or. p%x2 == p%x1)) return
       first = .not. p%pars_probed
       p%pars_probed = .true.
    end if
    saved = fitfuncs(p%dataset)%pars%val
    if (first) then
       if (.not. probe_at(p%x1, p%c1)) return
    end if
    if (.not. present(again) .and. .not. p%pars_probed2 .and. p%n_seen >= 2 .and. p%x2 /= p%x1) then
       p%pars_probed2 = .true.
       ad_theta = p%theta
       call record(p%dataset, p%x2, p%n_guards, p%script, res)
       ad_theta = 0.5_kp
       if (.not. same_as(p, res)) return          ! (eval() does something else at that abscissa: plain-real control flow, another path)
       base = ad_tape(1:p%n)%c
       if (.not. probe_at(p%x2, base)) return
    end if
  contains
    ! eval() along the path at abscissa xp with the active parameters moved; base: its literals there at the parameters as they are.
    ! .false.: nothing could be learnt (an integrand's own comparison came out differently)
    logical function probe_at(xp, base) result(learnt)
      real(kp), intent(in) :: xp, base(:)
      integer :: j
      learnt = .false.
      ! (only the ACTIVE parameters: a passive one keeps its value for the whole fit, so what eval() makes of its %val in plain real
      ! arithmetic -- an integer exponent, a switch -- is a constant of this model; gadf_fit captures the model again when a passive
      ! value or the active set has changed since: cap_vals, cap_active)
      call set_vals(fitfuncs(p%dataset)%pars, merge(saved*(1.0_kp + 1.0e-3_kp) + 1.0e-3_kp, saved, active_pars /= 0))
      ad_theta = p%theta
      call record(p%dataset, xp, p%n_guards, p%script, res)
      ad_theta = 0.5_kp
      call set_vals(fitfuncs(p%dataset)%pars, saved)
      ! (a comparison inside an integrand may legitimately come out differently at the perturbed parameters -- it is not forced, the
      ! device decides it per evaluation: nothing can be learnt from such a recording)
      if (p%sub_guards .and. .not. same_as(p, res)) return
      if (.not. same_as(p, res)) call error(__FILE__, __LINE__, 'eval() executes a different operation sequence when only &
           &the parameter values change, and no comparison of AD variables accounts for it (control flow on %val): such &
           &branches cannot follow the parameters on the device. Compare the advar itself.')
      learnt = .true.
      do j = 1, p%n
         if (p%raw(j)%op /= GFH_CONST) cycle
         if (ad_tape(j)%c /= base(j) .and. .not. (base(j) /= base(j))) then
            ! a real formed from the %val of a fitted parameter.  The reference recomputes it whenever eval() runs.  Where it is a
            ! function of the parameters ALONE it becomes a passive pseudo-parameter that on_pars recomputes before every pass
            ! (GFH_VAL, gadfit_tape.h); where it moves with x as well (class 2 or 3 by now) a per-point column that on_pars
            ! tabulates anew before every pass (lit_follow)
            if (p%psub(j) /= 0) then
               if (p%lit_class(j) /= 1 .and. p%lit_class(j) /= 4) call error(__FILE__, __LINE__, 'An integrand forms a real number from &
                    &parameter values (%val) AND the abscissa; such a literal cannot follow the parameters on the device. Keep it as advar.')
               ! (inside an integrand that eval() itself integrates: the pseudo-parameter is handed to the integrand as one more entry
               ! of its pars(:), bound at the call site -- build_tape; deeper down, and in integrands whose recordings are pooled per
               ! call site because they compare AD variables, there is no such way)
               if (p%sub_guards .or. caller_of(p, p%psub(j)) /= 0) call error(__FILE__, __LINE__, 'An integrand forms a real number from &
                    &parameter values (%val); such a literal cannot follow the parameters on the device here (an integrand of an integrand, or &
                    &one that compares AD variables). Pass the parameter to the integrand and keep it as advar.')
               p%lit_class(j) = 4
            else if ((p%lit_class(j) == 1 .or. p%lit_class(j) == 4) .and. .not. finite_differences) then
               p%lit_class(j) = 4
            else        ! (under use_ad = .false. every such real is a column: the sets of the forward differences carry its values at p + step e_j)
               p%lit_class(j) = 3; p%lit_follow(j) = .true.
            end if
         end if
      end do
    end function probe_at
  end subroutine probe_pars

  ! Is the captured model still what eval() does?  (later fits: the reference calls eval() afresh at every point of every fit, so a
  ! module variable the user changed between two fits takes effect there; here it sits in the recordings as a literal.)  Every
  ! path is recorded again at its first abscissa, its comparisons forced, and compared with what it was -- a few recordings.
  logical function capture_is_current() result(ok)
    integer :: q, res, j
    ok = .true.
    do q = 1, n_paths
       ad_theta = paths(q)%theta
       call record(paths(q)%dataset, paths(q)%x1, paths(q)%n_guards, paths(q)%script, res)
       ad_theta = 0.5_kp
       if (.not. same_as(paths(q), res)) then
          if (paths(q)%sub_guards) cycle        ! (an integrand's comparison may come out differently at the parameters of today)
          ok = .false.; return
       end if
       do j = 1, paths(q)%n
          if (paths(q)%raw(j)%op /= GFH_CONST .or. paths(q)%lit_class(j) == 4 .or. paths(q)%lit_follow(j)) cycle      ! (follow the parameters by design)
          if (ad_tape(j)%c /= paths(q)%c1(j) .and. .not. (paths(q)%c1(j) /= paths(q)%c1(j))) then
             ok = .false.; return
          end if
       end do
    end do
  end function capture_is_current

  ! A literal inside an INTEGRAND that follows the INTEGRATION VARIABLE (the integrand reading t%val into plain real arithmetic: a
  ! weight function outside the operator set, say) cannot be captured: a recording holds the value it had at the one abscissa the
  ! integrand was recorded at, and the abscissas of the quadrature exist only on the device.  The path is recorded once more at
  ! its first abscissa with the integration variables elsewhere in their ranges; every literal of its integrands must come out
  ! the same.  (What depends on x or on a fitted parameter as well is caught by observe / probe_pars.)
  subroutine probe_theta(p)
    type(path_t), intent(in out) :: p
    integer :: res, j
    if (p%theta_probed .or. p%nint == 0) return
    p%theta_probed = .true.
    ad_theta = merge(0.6180339887498949_kp, 0.3819660112501051_kp, abs(p%theta - 0.3819660112501051_kp) < 0.05_kp)
    call record(p%dataset, p%x1, p%n_guards, p%script, res)
    ad_theta = 0.5_kp
    if (.not. same_as(p, res)) then
       ! (a comparison inside the integrand may come out differently over there: another path, nothing to compare)
       if (p%sub_guards) return
       call error(__FILE__, __LINE__, 'An integrand executes a different operation sequence when only the value of its &
            &integration variable changes, and no comparison of AD variables accounts for it (control flow on %val): such &
            &branches cannot be followed on the device. Compare the advar itself.')
    end if
    do j = 1, p%n
       if (p%raw(j)%op /= GFH_CONST .or. p%psub(j) == 0) cycle
       if (ad_tape(j)%c /= p%c1(j) .and. .not. (p%c1(j) /= p%c1(j))) call error(__FILE__, __LINE__, &
            & 'An integrand forms a real number from the value of its integration variable (%val); such literals cannot &
            &follow the abscissas of the quadrature on the device. Keep them as advar: for an integration variable f(t) and &
            &f(t%val) are the same function with the same Jacobian (no derivative with respect to a fitting parameter flows &
            &through the integration variable); only the second directional derivative of an integral with an ACTIVE bound &
            &(geodesic acceleration) sees the difference.')
    end do
  end subroutine probe_theta

  ! A path that was met at ONE abscissa only (a sampled data set, a branch the device reported): it is recorded at two more
  ! abscissas of its dataset with its comparisons forced, so that its literals can be told apart.  Where eval() then does
  ! something else altogether (control flow on the plain real x) nothing can be learnt: every literal of eval() becomes an
  ! auxiliary column.
  subroutine probe_abscissas(p)
    type(path_t), intent(in out) :: p
    integer :: res, k
    integer(c_int64_t) :: lo, hi
    real(kp) :: xq(6)
    if (p%n_seen >= 2) return
    lo = data_positions(p%dataset) + 1; hi = data_positions(p%dataset + 1)
    xq(1) = xs(lo); xq(2) = xs(hi)
    if (xq(1) == p%x1) xq(1) = xs(min(lo + 1, hi))
    if (xq(2) == p%x1 .or. xq(2) == xq(1)) xq(2) = 0.5_kp*(xq(1) + p%x1) + 1.0e-3_kp*(abs(p%x1) + 1.0_kp)
    if (xq(1) == p%x1) xq(1) = p%x1*(1.0_kp + 1.0e-3_kp) + 1.0e-3_kp
    ! (... and, where eval() also branches on the plain real x so that the far ends of the data lie on other paths, abscissas close by)
    xq(3) = p%x1 + 1.0e-3_kp*(abs(p%x1) + 1.0_kp); xq(4) = p%x1 - 1.0e-3_kp*(abs(p%x1) + 1.0_kp)
    xq(5) = p%x1 + 1.0e-6_kp*(abs(p%x1) + 1.0_kp); xq(6) = p%x1 - 1.0e-6_kp*(abs(p%x1) + 1.0_kp)
    ad_theta = p%theta
    do k = 1, size(xq)
       if (k > 2 .and. p%n_seen >= 2) exit
       call record(p%dataset, xq(k), p%n_guards, p%script, res)
       if (same_as(p, res)) call observe(p, xq(k), p%dataset)
    end do
    ad_theta = 0.5_kp
    if (p%n_seen < 2) then
       ! (a real that follows the fitted parameters stays what probe_pars found it to be: as a column it would be frozen at the
       ! parameters of the tabulation)
       do k = 1, p%n
          if (p%raw(k)%op == GFH_CONST .and. p%psub(k) == 0 .and. p%lit_class(k) /= 4) p%lit_class(k) = 3
       end do
    end if
  end subroutine probe_abscissas

  ! eval() recorded over the data: at EVERY abscissa of every dataset, as the reference evaluates eval() afresh at every point
  ! (gadfit.F90:679-690) -- a feature of eval() that is invisible to operator overloading (control flow on the plain real x, a real
  ! function of x) and narrower than the spacing of a sample is then found like any other: its points do not check out against the
  ! known paths, are recorded in full and learnt from.  All but a handful of recordings run in checking mode on the recorder threads
  ! (0.1-0.6 us each: 1e7 points of the 32-parameter headline model in 0.3-0.4 s on 16 threads, once per capture; later fits check one
  ! abscissa per path).  GADFIT_HIP_VERIFY=sample restores the sampled capture of rounds 1-3 -- VERIFY_ALL_UP_TO evenly spaced
  ! abscissas beyond that many points (first and last point of every dataset included) -- for programs whose eval() is known to
  ! treat x through AD arithmetic only: 24 ms instead of 0.4 s for the first gadf_fit at 1e7 points, and a plain-real feature
  ! between two samples is silently frozen (tests/fortran/fit_narrow_window.F90 shows both).  A path with comparisons of AD
  ! variables that the capture has not seen is met by the device either way, which reports it: on_unseen.
  ! Yields the paths and what their literals are.
  subroutine discover()
    !$ use omp_lib, only: omp_get_max_threads
    integer :: d, res, q, step, loaded, k, mine, nthreads, stat, np_, pn, pres
    integer(c_int) :: cn, cdiv, clit
    integer(c_int64_t) :: tc0, tc1, tcr, td(4)
    logical, allocatable :: todo(:)
    character(len=16) :: envt
    integer(c_int64_t) :: i, lo, hi, n, probe(3), is, ns
    logical :: none(1), fast, failed
    character(len=256) :: fail_msg

    none = .false.
    n_paths = 0; last_match = 1; n_crossed = 0; eval_serial_only = .false.
    if (allocated(cross_q)) deallocate(cross_q)
    cross_all = .true.; n_set_cols = 0
    n = size(xs, kind=c_int64_t)
    step = 1
    call get_environment_variable('GADFIT_HIP_VERIFY', envt, status=stat)
    if (((stat == 0 .and. trim(adjustl(envt)) == 'sample') .or. opt_sampled_capture) .and. n > VERIFY_ALL_UP_TO) &
         & step = int((n + VERIFY_ALL_UP_TO - 1)/VERIFY_ALL_UP_TO)
    ! first, last and middle point of every dataset: the slopes of affine literals come from the longest baseline there is
    do d = 1, size(fitfuncs)
       lo = data_positions(d) + 1; hi = data_positions(d + 1)
       if (hi < lo) cycle
       probe = [lo, hi, (lo + hi)/2]
       do k = 1, 3
          call record(d, xs(probe(k)), 0, none, res)
          q = find_path(res)
          if (q == 0) then
             call add_path(d, res); q = n_paths
          end if
          call observe(paths(q), xs(probe(k)), d)
       end do
    end do
    ! then the sample.  A point whose recording agrees, node by node as it is made, with a known path and with what that
    ! path's literals are known to be costs a comparison per node and stores nothing (module ad, checking mode).  Whatever
    ! does not check out -- another path, a literal that is not what it was taken for -- is collected and then recorded in
    ! full, one point at a time, and learnt from.  (The threaded form of the check keeps its per-node state in native
    ! thread-local storage, ad_tls.c: with the recorder's state OpenMP-threadprivate the loop was 30 x SLOWER on one thread --
    ! flang reaches a threadprivate variable through a call into the OpenMP runtime at every access.)
    n_slow = 0
    if (allocated(slow_i)) deallocate(slow_i, slow_d)
    allocate(slow_i(1024), slow_d(1024))
    failed = .false.
    call system_clock(td(1), tcr)
    nthreads = 1
    call omp_defaults()
    !$ nthreads = min(merge(recorder_threads_max(), 8, n > VERIFY_ALL_UP_TO .and. step == 1), omp_get_max_threads())      ! (every point of a large data set: more threads than a sample takes)
    call get_environment_variable('GADFIT_HIP_RECORD_THREADS', envt, status=stat)
    if (stat == 0) read(envt, *, iostat=stat) nthreads
    nthreads = max(1, nthreads)
    if (opt_eval_one_thread) nthreads = 1               ! (gadf_init(..., eval_is_thread_safe=.false.))
    do d = 1, size(fitfuncs)
       lo = data_positions(d) + 1; hi = data_positions(d + 1)
       if (hi < lo) cycle
       ns = (hi - lo + step - 1)/step + 1                   ! samples lo, lo+step, ..., and hi
       ! Large samples first go through the known straight-line paths on several threads (GADFIT_HIP_RECORD_THREADS, default
       ! min(8, the OpenMP maximum); 1 = never): one parallel loop per such path over the samples still unaccounted for, every
       ! recording compared node by node with the path (module ad, ad_thread_check).  eval() is then called concurrently -- as the
       ! images of the reference call it, but here within one process: it must not keep state in saved or module variables
       ! (set the variable to 1 if it does).  What passes needs nothing more; the rest takes the serial loop below.
       if (allocated(todo)) deallocate(todo)
       allocate(todo(ns)); todo = .true.
       if (nthreads > 1 .and. ns >= threads_from()) then
          ! (64 evenly spaced samples first, recorded in full and learnt from: a path that many points take is then known well enough
          ! -- seen twice -- for the threads to check the others against it, instead of all of them ending on the serial list)
          do is = 1, ns, max(1_c_int64_t, ns/64)
             i = min(lo + (is - 1)*step, hi)
             call record(d, xs(i), 0, none, res)
             q = find_path(res)
             if (q == 0) then
                call add_path(d, res); q = n_paths
             end if
             call observe(paths(q), xs(i), d)
          end do
          do k = 1, size(fitfuncs(d)%pars)
             call set_node(fitfuncs(d)%pars(k), k - 1)
          end do
          do q = 1, n_paths
             associate(p => paths(q))
               if (p%n_seen < 2) cycle
               if (count(todo) < 4096) exit
               call load_check(p, d)
               call gfh_adchk_load(int(p%n, c_int), ad_chk_op, ad_chk_a, ad_chk_b, ad_chk_fl, ad_chk_cls, ad_chk_c, ad_chk_alpha, ad_chk_beta)
               if (p%nsub > 0) call load_check_ints(0, p)       ! (a path that calls integrate(): its call sites and sub-tapes, ad_tls.c)
               np_ = size(fitfuncs(d)%pars); pn = p%n; pres = p%res_node
               ! (a path with comparisons of AD variables: the values are computed, the natural outcome of every comparison is checked
               ! against the path's)
               ! (... and where a literal of the path follows the parameters: it may be formed from the %val of an INTERMEDIATE AD variable,
               ! which a check without values leaves at 0 -- the literal then differs from the known one at every point and every point
               ! lands on the serial list: right, but seconds instead of milliseconds)
               ad_recording = .true.; ad_thread_check = .true.
               ad_need_vals = p%n_guards > 0 .or. p%sub_guards .or. p%n_plit > 0 .or. any(p%lit_follow(1:p%n)); ad_cur = 0; ad_fast_check = .not. ad_need_vals
               call system_clock(tc0, tcr)
               !$omp parallel do schedule(static) num_threads(nthreads) default(shared) private(is, i, cn, cdiv, clit, res)
               do is = 1, ns
                  if (.not. todo(is)) cycle
                  i = min(lo + (is - 1)*step, hi)
                  if (is == ns) i = hi
                  call check_one(d, xs(i), np_, cn, cdiv, clit, res)
                  if (cdiv == 0 .and. clit == 0 .and. cn == pn .and. res == pres) todo(is) = .false.
               end do
               !$omp end parallel do
               ad_recording = .false.; ad_thread_check = .false.; ad_fast_check = .false.; ad_need_vals = .true.
               call get_environment_variable('GADFIT_HIP_SETUP_TIMES', envt, status=stat)
               if (stat == 0) then
                  call system_clock(tc1)
                  if (trim(adjustl(envt)) == '3') write(error_unit, '(a, i0, a, i0, a, i0, a, f9.3, a, i0, a)') 'threaded check: path ', q, ', samples left ', &
                       & count(todo), ' of ', ns, '  ', 1e3*real(tc1 - tc0)/real(tcr), ' ms on ', nthreads, ' threads'
               end if
             end associate
          end do
          do k = 1, size(fitfuncs(d)%pars)
             call set_node(fitfuncs(d)%pars(k), -1)
          end do
       end if
       do k = 1, size(fitfuncs(d)%pars)                     ! the PARAM nodes are the first of every recording: set once
          call set_node(fitfuncs(d)%pars(k), k - 1)
       end do
       loaded = 0; mine = last_match
       do is = 1, ns
          if (.not. todo(is)) cycle
          i = min(lo + (is - 1)*step, hi)
          if (is == ns) i = hi
          fast = .false.
          q = mine
          if (q >= 1 .and. q <= n_paths) then
             if (paths(q)%n_seen >= 2) then
                if (loaded /= q) then
                   call load_check(paths(q), d); loaded = q
                end if
                call ad_check_begin(xs(i), paths(q)%n_guards > 0)      ! (a path without comparisons: no advar's value matters to the check)
                call record_preset(d, xs(i), res)
                call ad_check_end()
                fast = .not. ad_chk_diverged .and. .not. ad_chk_litfail .and. checked_same(paths(q), res)
             end if
          end if
          if (.not. fast .and. .not. ad_capture_failed) then
             ! not the path of the point before it: recorded in full and looked up among the known paths (read only)
             call record_preset(d, xs(i), res)
             do k = 1, n_paths
                if (k == q) cycle
                if (.not. same_as(paths(k), res)) cycle
                if (paths(k)%n_seen >= 2) then
                   if (literals_as_known(paths(k), xs(i), d)) then
                      fast = .true.; mine = k
                   end if
                end if
                exit
             end do
          end if
          if (ad_capture_failed) then
             failed = .true.; fail_msg = ad_capture_msg
          else if (.not. fast) then
             call push_slow(d, i)
          end if
       end do
       do k = 1, size(fitfuncs(d)%pars)
          call set_node(fitfuncs(d)%pars(k), -1)
       end do
    end do
    if (failed) call error(__FILE__, __LINE__, trim(fail_msg))
    call system_clock(td(2))
    do is = 1, n_slow
       call record(slow_d(is), xs(slow_i(is)), 0, none, res)
       q = find_path(res)
       if (q == 0) then
          call add_path(slow_d(is), res); q = n_paths
       end if
       call observe(paths(q), xs(slow_i(is)), slow_d(is))
    end do
    if (n_paths == 0) call error(__FILE__, __LINE__, 'There are no data points.')
    ! An integrand that compares AD variables takes its branch anew at every abscissa of the quadrature (AD:315-395), and a
    ! recording follows one path through it, at the one abscissa the integration variable is given (numerical_integration.F90,
    ! probe_at: ad_theta of the way through its range).  64 data points per dataset are recorded again with it at a dozen other
    ! places; every new path through an integrand is a recording of its own, and the library pools those that share eval()'s path
    ! into one call site (Model::alts).  What this misses the device reports as an error, not as a wrong integral.
    if (any(paths(1:n_paths)%sub_guards)) call explore_integrands()
    call cross_check()
    call system_clock(td(3))
    do q = 1, n_paths
       call probe_pars(paths(q)); call probe_theta(paths(q))
       call probe_abscissas(paths(q))
    end do
    call system_clock(td(4))
    call get_environment_variable('GADFIT_HIP_SETUP_TIMES', envt, status=stat)
    if (stat == 0) then
       if (trim(adjustl(envt)) == '3') write(error_unit, '(a, 3f9.3)') 'discover [ms]: sample loop, slow list, probes: ', &
            & 1e3*real(td(2) - td(1))/real(tcr), 1e3*real(td(3) - td(2))/real(tcr), 1e3*real(td(4) - td(3))/real(tcr)
    end if
  end subroutine discover

  ! A path through eval()'s comparisons of AD variables is recorded where the data first take it; the device then sends ANY point there
  ! whose comparisons come out that way at the parameters of some later pass.  Is the path the same for those points?  Not where
  ! eval() also branches on the plain real x (or forms reals from it that the recordings so far took for constants): the reference,
  ! which runs eval() afresh at every point (gadfit.F90:679-690), would take the other branch there, and operator overloading cannot
  ! see it.  So every set of outcomes that some path holds is FORCED on eval() at every data point (the sample, under
  ! GADFIT_HIP_VERIFY=sample) -- on the recorder threads, each recording checked node by node against the paths known with those
  ! outcomes -- and what follows none of them is recorded in full: a new path (parting from its siblings without a comparison: the
  ! per-point variant column then tells them apart) or a literal to be learnt.  Sets of outcomes met later (on_unseen) are crossed
  ! with the data when they appear.
  subroutine cross_check()
    !$ use omp_lib, only: omp_get_max_threads
    integer :: q, r, d, k, j, res, np_, nthreads, stat, step, ng, ncand, mine, tried, nmax
    integer, allocatable :: cand(:)
    integer(c_int64_t) :: bits, lo, hi, is, ns, i, n, n_bad
    integer(c_int) :: cn, cdiv, clit
    logical, allocatable :: bad(:)
    logical :: found, script(64), known
    character(len=16) :: envt
    integer(c_int32_t), allocatable, save :: k_op(:,:), k_a(:,:), k_b(:,:), k_fl(:,:), k_cls(:,:)
    real(c_double), allocatable, save :: k_c(:,:), k_al(:,:), k_be(:,:)
    integer, allocatable :: tn(:)
    integer(c_int64_t), allocatable :: tb(:)
    integer(c_int16_t), allocatable :: cq_tmp(:,:)
    integer :: ks
    if (x_copy_pending .and. .not. associated(xs)) then
       cross_all = .false.; return
    end if
    ! (GADFIT_HIP_CROSS_CHECK=0: not at all -- for an eval() whose branches must not be entered where their own comparison is false,
    ! e.g. one that indexes a table by the abscissa behind `if (x < p)`; a fork on the plain real x behind a comparison is then not seen)
    call get_environment_variable('GADFIT_HIP_CROSS_CHECK', envt, status=stat)
    if ((stat == 0 .and. trim(adjustl(envt)) == '0') .or. opt_no_forced_outcomes) then       ! (or gadf_init(..., force_outcomes=.false.))
       cross_all = .false.; return
    end if
    n = size(xs, kind=c_int64_t)
    step = 1
    call get_environment_variable('GADFIT_HIP_VERIFY', envt, status=stat)
    if (((stat == 0 .and. trim(adjustl(envt)) == 'sample') .or. opt_sampled_capture) .and. n > VERIFY_ALL_UP_TO) &
         & step = int((n + VERIFY_ALL_UP_TO - 1)/VERIFY_ALL_UP_TO)
    if (step > 1) cross_all = .false.            ! (a sample: the other points' paths under these outcomes stay unknown)
    nthreads = 1
    call omp_defaults()
    !$ nthreads = min(recorder_threads_max(), omp_get_max_threads())
    call get_environment_variable('GADFIT_HIP_RECORD_THREADS', envt, status=stat)
    if (stat == 0) read(envt, *, iostat=stat) nthreads
    nthreads = max(1, nthreads)
    if (opt_eval_one_thread) nthreads = 1               ! (gadf_init(..., eval_is_thread_safe=.false.))
    if (.not. allocated(crossed_n)) allocate(crossed_n(16), crossed_bits(16))
    q = 1
    do while (q <= n_paths)                     ! (paths found on the way come up in their turn)
       ng = paths(q)%n_guards
       if (ng == 0 .or. ng > 64) then
          q = q + 1; cycle
       end if
       bits = 0_c_int64_t
       do j = 1, ng
          if (paths(q)%script(j)) bits = ibset(bits, j - 1)
       end do
       known = .false.
       do k = 1, n_crossed
          if (crossed_n(k) == ng .and. crossed_bits(k) == bits) known = .true.
       end do
       if (known) then
          q = q + 1; cycle
       end if
       if (n_crossed == size(crossed_n)) then
          allocate(tn(2*n_crossed), tb(2*n_crossed))
          tn(:n_crossed) = crossed_n(:n_crossed); tb(:n_crossed) = crossed_bits(:n_crossed)
          call move_alloc(tn, crossed_n); call move_alloc(tb, crossed_bits)
       end if
       n_crossed = n_crossed + 1; crossed_n(n_crossed) = ng; crossed_bits(n_crossed) = bits
       script = .false.; script(:ng) = paths(q)%script(:ng)
       if (cross_all) then                        ! (a column of cross_q for this set: 0 = not known)
          if (.not. allocated(cross_q)) then
             allocate(cross_q(n, 4)); cross_q = 0_c_int16_t
          else if (size(cross_q, 1, kind=c_int64_t) /= n) then
             cross_all = .false.
          else if (n_crossed > size(cross_q, 2)) then
             allocate(cq_tmp(n, 2*size(cross_q, 2))); cq_tmp = 0_c_int16_t
             cq_tmp(:, :size(cross_q, 2)) = cross_q
             call move_alloc(cq_tmp, cross_q)
          end if
       end if
       do d = 1, size(fitfuncs)
          lo = data_positions(d) + 1; hi = data_positions(d + 1)
          if (hi < lo) cycle
          ! a few points recorded in full first: the literals of these outcomes' paths are then known over this dataset's range
          do is = 0, 8
             i = lo + (is*(hi - lo))/8
             call record(d, xs(i), ng, script, res)
             r = find_path(res)
             if (r == 0) then
                call add_path(d, res); r = n_paths
             end if
             call observe(paths(r), xs(i), d)
          end do
          ns = (hi - lo + step - 1)/step + 1
          if (allocated(bad)) deallocate(bad)
          allocate(bad(ns)); bad = .false.
          ! the paths known with these outcomes, side by side for the threads (ad_tls.c holds up to 16)
          if (allocated(cand)) deallocate(cand)
          allocate(cand(n_paths)); ncand = 0
          do r = 1, n_paths
             if (paths(r)%n_guards /= ng) cycle
             if (any(paths(r)%script(:ng) .neqv. script(:ng))) cycle
             if (paths(r)%n_seen < 2) cycle       ! (its literals are not told apart yet: the few points that take it go the serial way)
             ncand = ncand + 1; cand(ncand) = r
          end do
          if (nthreads > 1 .and. ns >= threads_from() .and. ncand >= 1 .and. ncand <= 16) then
             nmax = maxval(paths(cand(:ncand))%n)
             if (allocated(k_op)) deallocate(k_op, k_a, k_b, k_fl, k_cls, k_c, k_al, k_be)
             allocate(k_op(nmax, ncand), k_a(nmax, ncand), k_b(nmax, ncand), k_fl(nmax, ncand), k_cls(nmax, ncand), &
                  & k_c(nmax, ncand), k_al(nmax, ncand), k_be(nmax, ncand))
             do k = 1, ncand
                associate(p => paths(cand(k)))
                  call load_check(p, d)
                  k_op(1:p%n, k) = ad_chk_op(1:p%n); k_a(1:p%n, k) = ad_chk_a(1:p%n); k_b(1:p%n, k) = ad_chk_b(1:p%n); k_fl(1:p%n, k) = ad_chk_fl(1:p%n)
                  k_cls(1:p%n, k) = ad_chk_cls(1:p%n); k_c(1:p%n, k) = ad_chk_c(1:p%n); k_al(1:p%n, k) = ad_chk_alpha(1:p%n); k_be(1:p%n, k) = ad_chk_beta(1:p%n)
                  call gfh_adchk_load_path(int(k - 1, c_int), int(p%n, c_int), k_op(:, k), k_a(:, k), k_b(:, k), k_fl(:, k), k_cls(:, k), &
                       & k_c(:, k), k_al(:, k), k_be(:, k))
                  if (p%nsub > 0) call load_check_ints(k - 1, p)
                end associate
             end do
             np_ = size(fitfuncs(d)%pars)
             do k = 1, np_
                call set_node(fitfuncs(d)%pars(k), k - 1)
             end do
             ad_recording = .true.; ad_thread_check = .true.; ad_need_vals = .true.; ad_cur = 0
             ks = n_crossed
             !$omp parallel default(shared) num_threads(nthreads) private(is, i, cn, cdiv, clit, res, k, mine, tried, found)
             mine = 1
             !$omp do schedule(static)
             do is = 1, ns
                i = min(lo + (is - 1)*step, hi)
                if (is == ns) i = hi
                found = .false.
                do tried = 0, ncand - 1
                   k = mod(mine - 1 + tried, ncand) + 1
                   call gfh_adchk_script(int(ng, c_int), bits)
                   call gfh_adchk_use(int(k - 1, c_int))
                   call check_one(d, xs(i), np_, cn, cdiv, clit, res)
                   if (cdiv /= 0 .or. clit /= 0 .or. cn /= paths(cand(k))%n .or. res /= paths(cand(k))%res_node) cycle
                   found = .true.; mine = k
                   exit
                end do
                if (.not. found) then
                   bad(is) = .true.
                else if (cross_all) then
                   cross_q(i, ks) = int(cand(mine), c_int16_t)
                end if
             end do
             !$omp end do
             call gfh_adchk_script(0_c_int, 0_c_int64_t)
             call gfh_adchk_use(0_c_int)
             !$omp end parallel
             ad_recording = .false.; ad_thread_check = .false.; ad_fast_check = .false.; ad_need_vals = .true.
             do k = 1, np_
                call set_node(fitfuncs(d)%pars(k), -1)
             end do
          else
             bad = .true.
          end if
          n_bad = 0
          do is = 1, ns
             if (.not. bad(is)) cycle
             i = min(lo + (is - 1)*step, hi)
             if (is == ns) i = hi
             n_bad = n_bad + 1
             call record(d, xs(i), ng, script, res)
             r = find_path(res)
             if (r == 0) then
                call add_path(d, res); r = n_paths
             end if
             if (cross_all) cross_q(i, n_crossed) = int(r, c_int16_t)
             call observe(paths(r), xs(i), d)
          end do
          call get_environment_variable('GADFIT_HIP_SETUP_TIMES', envt, status=stat)
          if (stat == 0) then
             if (trim(adjustl(envt)) == '3') write(error_unit, '(a, i0, a, i0, a, i0, a, i0, a, i0, a)') 'cross_check: outcomes of path ', q, ' (', ng, &
                  & ' comparisons) over dataset ', d, ': ', ns, ' points, ', n_bad, ' recorded in full'
          end if
       end do
       q = q + 1
    end do
  end subroutine cross_check

  ! 64 data points per dataset recorded with the integration variable of their integrands at a dozen places of its range besides
  ! the middle (discover; on_unseen when the device reports an integrand path nobody recorded): new paths join the model
  subroutine explore_integrands()
    real(kp), parameter :: thetas(13) = [0.5_kp, 0.003_kp, 0.03_kp, 0.1_kp, 0.2_kp, 0.3_kp, 0.4_kp, 0.6_kp, 0.7_kp, 0.8_kp, 0.9_kp, &
         & 0.97_kp, 0.997_kp]
    integer :: d, k, res, q
    integer(c_int64_t) :: lo, hi, is, i
    logical :: none(1)
    none = .false.
    if (x_copy_pending .and. .not. associated(xs)) return
    do d = 1, size(fitfuncs)
       lo = data_positions(d) + 1; hi = data_positions(d + 1)
       if (hi < lo) cycle
       do is = 0, min(63_c_int64_t, hi - lo)
          i = lo + (is*(hi - lo))/max(1_c_int64_t, min(63_c_int64_t, hi - lo))
          do k = 1, size(thetas)
             ad_theta = thetas(k)
             call record(d, xs(i), 0, none, res)
             q = find_path(res)
             if (q == 0) then
                call add_path(d, res); q = n_paths
             end if
             call observe(paths(q), xs(i), d)
          end do
       end do
    end do
    ad_theta = 0.5_kp
  end subroutine explore_integrands

  ! The tape of path p from its raw recording: x-dependent literals expressed through the X node or an auxiliary column -- in
  ! eval()'s own tape and in the integrands' (a real an integrand takes from the enclosing eval() without passing it through
  ! pars(:): the reference evaluates the integrand afresh in that scope, numerical_integration.F90:195-201; the device reads the
  ! point's abscissa and columns back from a per-lane stash, codegen.cpp GFH_LANE_STASH); all sub-tapes contiguous in p%final.
  subroutine build_tape(p)
    type(path_t), intent(in out), target :: p
    integer, allocatable :: remap(:)
    integer :: k, n, nf, xnode, s, base, i, na, lc, npl, kk, jx, n_extra_total
    integer, allocatable :: n_extra(:), extra_node(:,:), new_off(:), extra_rank(:)
    real(kp) :: alpha, beta
    n = p%n
    ! Reals that an INTEGRAND forms from the %val of fitted parameters (lit_class 4 inside a sub-tape): the integrand has no parameter
    ! block to read, so each becomes one more entry of its pars(:) -- GFH_VAL(GFH_IPARAM(n_ipars + j)) inside, bound at the call site
    ! to GFH_VAL(GFH_PARAM(slot)) of eval()'s tape (passive both ways: no derivative flows through a %val, as in the reference).
    ! n_extra(i): how many call site i carries; extra_rank(k): which of them raw node k is.
    allocate(n_extra(max(1, p%nint)), extra_rank(max(1, n)), new_off(max(1, p%nint)))
    n_extra = 0; extra_rank = 0
    do k = 1, n
       if (p%raw(k)%op /= GFH_CONST .or. p%lit_class(k) /= 4 .or. p%psub(k) == 0) cycle
       do i = 1, p%nint
          if (p%pints(i)%integrand /= p%psub(k)) cycle
          if (p%pint_sub(i) /= 0) call error(__FILE__, __LINE__, 'internal: a %val literal inside a nested integrand reached build_tape')
          n_extra(i) = n_extra(i) + 1; extra_rank(k) = n_extra(i)
       end do
    end do
    n_extra_total = sum(n_extra(:max(1, p%nint)))
    allocate(extra_node(max(1, maxval(n_extra)), max(1, p%nint)))
    jx = 0
    do i = 1, p%nint
       new_off(i) = jx; jx = jx + p%pints(i)%n_ipars + n_extra(i)
    end do
    if (allocated(p%final)) deallocate(p%final, p%sub, p%ints, p%ipar, p%aux_raw_k, p%plit_raw_k)
    allocate(p%final(4*n + p%nsub + 8 + 2*n_extra_total), p%sub(p%nsub + 1), p%ints(max(1, p%nint)), p%ipar(max(1, p%nip + n_extra_total)), p%aux_raw_k(max(1, n)), p%plit_raw_k(max(1, n)))
    npl = 0
    allocate(remap(0:max(maxval(p%cnt(0:p%nsub)) - 1, 0)))
    nf = 0; na = 0
    do s = 0, p%nsub
       base = nf
       lc = 0; xnode = -1                               ! node index local to sub-tape s; its X node (made at first use)
       do k = 1, n
          if (p%psub(k) /= s) cycle
          associate(nd => p%raw(k))
            if (nd%op == GFH_CONST) then
               if (p%lit_class(k) == 4) then
                  npl = npl + 1
                  p%plit_raw_k(npl) = k
                  if (s == 0) then
                     call push(GFH_PARAM, size(fitfuncs(1)%pars) + p%plit_slot(k), -1, 0, 0.0_kp)
                  else        ! (the extra entry of this integrand's pars(:): see the head of this routine)
                     do i = 1, p%nint
                        if (p%pints(i)%integrand == s) call push(GFH_IPARAM, p%pints(i)%n_ipars + extra_rank(k) - 1, -1, 0, 0.0_kp)
                     end do
                  end if
                  call push(GFH_VAL, nf - 1 - base, -1, GFH_F_REAL, 0.0_kp)
                  remap(lc) = nf - 1 - base
               else if (p%lit_class(k) <= 1) then
                  call push(GFH_CONST, -1, -1, GFH_F_REAL, p%lit_c(k))
                  remap(lc) = nf - 1 - base
               else if (p%lit_class(k) == 3) then
                  na = na + 1
                  p%aux_raw_k(na) = k
                  call push(GFH_AUX, p%aux0 + na - 1, -1, GFH_F_REAL, 0.0_kp)
                  remap(lc) = nf - 1 - base
               else
                  alpha = p%lit_alpha(k); beta = p%lit_beta(k)
                  if (xnode < 0) then
                     call push(GFH_X, -1, -1, GFH_F_REAL, 0.0_kp)
                     xnode = nf - 1 - base
                  end if
                  remap(lc) = xnode
                  if (alpha == -1.0_kp) then
                     call push(GFH_NEG, remap(lc), -1, GFH_F_REAL, 0.0_kp)
                     remap(lc) = nf - 1 - base
                  else if (alpha /= 1.0_kp) then
                     call push(GFH_CONST, -1, -1, GFH_F_REAL, alpha)
                     call push(GFH_MUL, nf - 1 - base, remap(lc), GFH_F_REAL, 0.0_kp)
                     remap(lc) = nf - 1 - base
                  end if
                  if (beta /= 0.0_kp) then
                     call push(GFH_CONST, -1, -1, GFH_F_REAL, beta)
                     call push(GFH_ADD, remap(lc), nf - 1 - base, GFH_F_REAL, 0.0_kp)
                     remap(lc) = nf - 1 - base
                  end if
               end if
            else
               select case (nd%op)
               case (GFH_INTEGRATE)
                  ! (the pseudo-parameters its integrand reads: nodes of this tape, in front of the call site that binds them)
                  i = nd%a + 1
                  if (s == 0 .and. i >= 1 .and. i <= p%nint) then
                     if (n_extra(i) > 0) then
                        do kk = 1, n
                           if (extra_rank(kk) == 0 .or. p%psub(kk) /= p%pints(i)%integrand) cycle
                           call push(GFH_PARAM, size(fitfuncs(1)%pars) + p%plit_slot(kk), -1, 0, 0.0_kp)
                           call push(GFH_VAL, nf - 1 - base, -1, GFH_F_REAL, 0.0_kp)
                           extra_node(extra_rank(kk), i) = nf - 1 - base
                        end do
                     end if
                  end if
                  call push(nd%op, nd%a, -1, nd%flags, 0.0_kp)
               case (GFH_PARAM, GFH_IVAR, GFH_IPARAM)
                  call push(nd%op, nd%a, -1, nd%flags, 0.0_kp)
               case (GFH_POWI)
                  call push(nd%op, remap(nd%a), nd%b, nd%flags, 0.0_kp)
               case (GFH_ADD, GFH_SUB, GFH_MUL, GFH_DIV, GFH_POW, GFH_GUARD_GT, GFH_GUARD_LT)
                  call push(nd%op, remap(nd%a), remap(nd%b), nd%flags, 0.0_kp)
               case default
                  call push(nd%op, remap(nd%a), -1, nd%flags, 0.0_kp)
               end select
               remap(lc) = nf - 1 - base
            end if
          end associate
          lc = lc + 1
       end do
       p%sub(s+1)%n_nodes = nf - base
       p%sub(s+1)%nodes = c_loc(p%final(base+1))
       if (s == 0) then
          p%sub(s+1)%result = remap(p%res_node)
       else
          p%sub(s+1)%result = remap(p%sub_result(s))
       end if
       ! the integrate() call sites made from sub-tape s: their bounds and bindings are nodes of s
       do i = 1, p%nint
          if (p%pint_sub(i) /= s) cycle
          p%ints(i) = p%pints(i)
          if (p%ints(i)%lower_inf == 0) p%ints(i)%lower = remap(p%pints(i)%lower)
          if (p%ints(i)%upper_inf == 0) p%ints(i)%upper = remap(p%pints(i)%upper)
          do k = 1, p%pints(i)%n_ipars
             p%ipar(new_off(i) + k) = remap(p%pipar(p%pints(i)%ipar_off + k))
          end do
          do k = 1, n_extra(i)
             p%ipar(new_off(i) + p%pints(i)%n_ipars + k) = extra_node(k, i)
          end do
          p%ints(i)%ipar_off = new_off(i)
          p%ints(i)%n_ipars = p%pints(i)%n_ipars + n_extra(i)
       end do
    end do
    if (na /= p%n_aux) call error(__FILE__, __LINE__, 'internal: auxiliary column count changed while the tape was built')
    p%tape%n_pars = size(fitfuncs(1)%pars) + n_plit_cap; p%tape%n_subtapes = p%nsub + 1; p%tape%sub = c_loc(p%sub)
    p%tape%n_integrals = p%nint; p%tape%integrals = c_loc(p%ints); p%tape%ipar_nodes = c_loc(p%ipar)
    p%tape%gk_points = int_rule
    p%tape%rel_error_outer = int_rel_error_outer; p%tape%rel_error_inner = int_rel_error_inner
    p%tape%ws_size = int_ws_size; p%tape%ws_size_inner = int_ws_size_inner
  contains
    subroutine push(op, a, b, flags, c)
      integer, intent(in) :: op, a, b, flags
      real(kp), intent(in) :: c
      nf = nf + 1
      p%final(nf)%op = op; p%final(nf)%a = a; p%final(nf)%b = b; p%final(nf)%flags = flags; p%final(nf)%c = c
    end subroutine push
  end subroutine build_tape

  ! All paths as variant tapes to the context tgt (the user's context, or the member of a device group that reported a
  ! branch).  Columns: the auxiliary literals of path 1, of path 2, ... and last -- only where the library finds variants
  ! that part ways without a comparison -- the per-point variant column.
  subroutine upload_model(tgt)
    type(c_ptr), intent(in) :: tgt
    type(c_ptr), allocatable :: tapes(:)
    integer(c_int32_t), allocatable :: hcols(:)
    integer :: q, k, r, j, trace_stat
    character(len=8) :: trace_env
    n_aux_total = 0; n_follow = 0
    if (allocated(tab_keep_pars)) deallocate(tab_keep_pars)       ! (the columns are laid out anew: what is kept belongs to the old layout)
    if (.not. fit_in_progress) then
       n_plit_total = 0
       do q = 1, n_paths
          paths(q)%plit_slot = -1
       end do
    end if
    do q = 1, n_paths
       paths(q)%n_aux = count(paths(q)%raw%op == GFH_CONST .and. paths(q)%lit_class == 3)
       n_follow = n_follow + count(paths(q)%raw%op == GFH_CONST .and. paths(q)%lit_class == 3 .and. paths(q)%lit_follow)
       paths(q)%aux0 = n_aux_total
       n_aux_total = n_aux_total + paths(q)%n_aux
       paths(q)%n_plit = count(paths(q)%raw%op == GFH_CONST .and. paths(q)%lit_class == 4)
       paths(q)%plit0 = n_plit_total
       do k = 1, paths(q)%n
          if (paths(q)%raw(k)%op /= GFH_CONST .or. paths(q)%lit_class(k) /= 4 .or. paths(q)%plit_slot(k) >= 0) cycle
          paths(q)%plit_slot(k) = n_plit_total
          n_plit_total = n_plit_total + 1
       end do
    end do
    ! (use_ad = .false.: the reference's forward differences evaluate eval() at p + step, where a real formed from a fitted
    ! parameter's %val has moved too -- fitfunction.F90:155-174 -- while the pseudo-parameter that carries it here is refreshed once
    ! per pass: the two derivatives would differ, silently)
    if (fit_in_progress) then
       if (n_plit_total > n_plit_cap) call error(__FILE__, __LINE__, 'A branch of eval() first met during this fit forms more real &
            &numbers from the %val of fitted parameters than the model has room for. Call gadf_fit again: the model is captured &
            &anew with the branches known by now.')
    else
       n_plit_cap = n_plit_total
       if (any(paths(1:n_paths)%n_guards > 0) .or. any(paths(1:n_paths)%sub_guards)) n_plit_cap = n_plit_total + PLIT_SPARE
    end if
    if (finite_differences .and. n_plit_total > 0) call error(__FILE__, __LINE__, 'use_ad=.false. with a real number that an integrand &
         &forms from the %val of a fitted parameter: the finite differences of the device do not move such numbers with the &
         &parameter. Keep them as advar, or fit with automatic differentiation.')
    if (finite_differences .and. n_follow > 0 .and. accel_requested) call error(__FILE__, __LINE__, 'use_ad=.false. with geodesic acceleration &
         &(accth > 0) and a real number that eval() forms from the %val of a fitted parameter: the central difference of the second &
         &directional derivative (fitfunction.F90:188-203) would need such numbers at p +- h*delta. Fit without acceleration.')
    call lib_check(gfh_set_fd_column_sets(tgt, merge(1_c_int, 0_c_int, finite_differences .and. n_follow > 0)), __FILE__, __LINE__)
    call lib_check(gfh_set_pars_hook(tgt, merge(c_funloc(on_pars), c_null_funptr, n_plit_total > 0 .or. n_follow > 0), c_null_ptr), __FILE__, __LINE__)
    ! One source literal that is affine in x -- `x*c + d` before a comparison, say -- is met on several paths, and each path has fitted
    ! its slope and offset from its own abscissas: equal to a few units in the last place, not bit for bit.  The library takes recordings
    ! that differ in a node for different code (a fork without a comparison: the per-point variant column, a report and a new
    ! tabulation whenever a point changes sides), so what agrees to 1e-11 (the tolerance the checks of the capture use) is given the
    ! numbers of the path that came first.
    do q = 2, n_paths
       do k = 1, paths(q)%n
          if (paths(q)%raw(k)%op /= GFH_CONST .or. paths(q)%lit_class(k) /= 2) cycle
          canon: do r = 1, q - 1
             do j = 1, paths(r)%n
                if (paths(r)%raw(j)%op /= GFH_CONST .or. paths(r)%lit_class(j) /= 2) cycle
                if (abs(paths(q)%lit_alpha(k) - paths(r)%lit_alpha(j)) <= 1e-11_kp*abs(paths(r)%lit_alpha(j)) .and. &
                     & abs(paths(q)%lit_beta(k) - paths(r)%lit_beta(j)) <= 1e-11_kp*(abs(paths(r)%lit_beta(j)) + abs(paths(r)%lit_alpha(j)*paths(r)%x1))) then
                   paths(q)%lit_alpha(k) = paths(r)%lit_alpha(j); paths(q)%lit_beta(k) = paths(r)%lit_beta(j)
                   exit canon
                end if
             end do
          end do canon
       end do
    end do
    allocate(tapes(n_paths))
    do q = 1, n_paths
       call build_tape(paths(q))
       paths(q)%tape%n_aux = n_aux_total
       tapes(q) = c_loc(paths(q)%tape)
    end do
    hint_col = -1; n_set_cols = 0; natural_col_used = .true.
    call lib_check(gfh_set_model_variants(tgt, int(n_paths, c_int), tapes, -1_c_int), __FILE__, __LINE__)
    if (n_paths > 1) then
       if (gfh_model_needs_hint(tgt) == 1) then
          hint_col = n_aux_total
          allocate(hcols(n_paths))
          call plan_hint_columns(hcols)
          do q = 1, n_paths
             paths(q)%tape%n_aux = n_aux_total + 1 + n_set_cols
          end do
          if (n_set_cols > 0) call lib_check(gfh_set_variant_hint_columns(tgt, int(n_paths, c_int), hcols), __FILE__, __LINE__)
          call lib_check(gfh_set_model_variants(tgt, int(n_paths, c_int), tapes, int(hint_col, c_int)), __FILE__, __LINE__)
       end if
    end if
    need_tab = n_aux_total > 0 .or. hint_col >= 0
    tabulated = .false.
    call get_environment_variable('GADFIT_HIP_TRACE_PATHS', trace_env, status=trace_stat)      ! (debugging: what the capture holds)
    if (trace_stat == 0) then
       write(error_unit, '(a, i0, a, i0, a, i0, a, i0, a, l1)') 'model: ', n_paths, ' path(s), ', n_aux_total, ' column(s), hint column ', hint_col, &
            & ', pseudo-parameters ', n_plit_total, ', during a fit: ', fit_in_progress
       do q = 1, n_paths
          write(error_unit, '(a, i0, a, i0, a, i0, a, i0, a, es12.5, a, i0, a, 64l1)') '  path ', q, ': nodes ', paths(q)%n, ', dataset ', paths(q)%dataset, &
               & ', abscissas seen ', paths(q)%n_seen, ', first x ', paths(q)%x1, ', comparisons ', paths(q)%n_guards, ' outcomes ', paths(q)%script(:paths(q)%n_guards)
          do k = 1, paths(q)%n
             if (paths(q)%raw(k)%op /= GFH_CONST) cycle
             write(error_unit, '(a, i0, a, i0, a, es23.15, a, es12.4, a, es12.4, a, l1, a, l1)') '    literal at node ', k, ': class ', paths(q)%lit_class(k), ', value ', &
                  & paths(q)%lit_c(k), ', alpha ', paths(q)%lit_alpha(k), ', beta ', paths(q)%lit_beta(k), ', follows the dataset ', paths(q)%ds_dep(k), &
                  & ', follows the fitted parameters ', paths(q)%lit_follow(k)
          end do
       end do
    end if
  end subroutine upload_model

  ! Which per-point variant column does the device read for each path?  The natural one (hint_col: the path a point takes by itself
  ! at the parameters of the tabulation) for every path -- or, where cross_check has seen EVERY data point under the outcomes of every
  ! path that compares AD variables (cross_all, cross_q), the column of the path's own set of outcomes behind it (module header).
  ! GADFIT_HIP_HINT_SETS=0: the natural column only (round 4's scheme: a point that changes sides at a comparison in front of a fork
  ! is reported and the column tabulated anew).
  subroutine plan_hint_columns(cols)
    integer(c_int32_t), intent(out) :: cols(:)
    integer :: q, k, c, j, ng, stat
    integer(c_int64_t) :: bits
    logical :: multi
    character(len=8) :: envh
    cols = int(hint_col, c_int32_t)
    n_set_cols = 0; natural_col_used = .true.
    if (allocated(set_of_col)) deallocate(set_of_col)
    allocate(set_of_col(max(1, n_crossed)))
    call get_environment_variable('GADFIT_HIP_HINT_SETS', envh, status=stat)
    multi = cross_all .and. allocated(cross_q) .and. n_crossed > 0 .and. .not. (stat == 0 .and. trim(adjustl(envh)) == '0')
    if (multi) multi = size(cross_q, 1) == size(xs) .and. size(cross_q, 2) >= n_crossed
    if (.not. multi) return
    natural_col_used = .false.
    do q = 1, n_paths
       ng = paths(q)%n_guards
       if (ng == 0) then
          natural_col_used = .true.; cycle
       end if
       if (ng > 64) then
          multi = .false.; exit
       end if
       bits = 0_c_int64_t
       do j = 1, ng
          if (paths(q)%script(j)) bits = ibset(bits, j - 1)
       end do
       k = 0
       do j = 1, n_crossed
          if (crossed_n(j) == ng .and. crossed_bits(j) == bits) k = j
       end do
       if (k == 0) then                           ! (a path met after the cross-check, its outcomes not yet forced on the data)
          multi = .false.; exit
       end if
       c = 0
       do j = 1, n_set_cols
          if (set_of_col(j) == k) c = j
       end do
       if (c == 0) then
          if (any(cross_q(:, k) <= 0_c_int16_t)) then      ! (some point's path under these outcomes is not known)
             multi = .false.; exit
          end if
          n_set_cols = n_set_cols + 1; set_of_col(n_set_cols) = k; c = n_set_cols
       end if
       cols(q) = int(hint_col + c, c_int32_t)
    end do
    if (.not. multi) then
       cols = int(hint_col, c_int32_t); n_set_cols = 0; natural_col_used = .true.
    end if
  end subroutine plan_hint_columns

  ! Auxiliary per-point columns: eval() is recorded once per data point; the literals that are neither constant nor
  ! affine in x are read out of the recording (for every path: those a point is not on are reached by forcing the path's
  ! comparisons, so that a point that changes path during the fit finds its values), and the per-point variant column
  ! names the path the point takes at the current parameters.  A path first met here joins the model.
  subroutine tabulate(tgt)
    !$ use omp_lib, only: omp_get_max_threads
    type(c_ptr), intent(in) :: tgt
    real(c_double), allocatable :: tab(:,:)
    logical, allocatable :: done(:)
    integer :: d, res, q, r, j, ncol, round, nthreads, stat, np_, pn, k, nmax, npth, hc, mine, tried
    integer(c_int32_t), allocatable, save :: k_op(:,:), k_a(:,:), k_b(:,:), k_fl(:,:), k_cls(:,:)
    real(c_double), allocatable, save :: k_c(:,:), k_al(:,:), k_be(:,:)
    logical :: par_ok, racy, same, any_guards, found
    integer(c_int64_t) :: stride, nd_pts, n_spot, n_draw, k_spot
    integer(c_int64_t), save :: draw_state = 0
    integer(c_int64_t), allocatable, save :: sbits(:)
    real(c_double), allocatable :: row(:)
    real(kp), allocatable :: row_c(:)
    integer :: pass, n_racy
    integer(c_int64_t) :: tk0, tk1, tkr
    integer(c_int) :: cn, cdiv, clit, got
    real(c_double) :: vals(64)
    integer(c_int32_t) :: nodes(64)
    integer(c_int64_t) :: i
    logical :: grew, none(1)
    character(len=16) :: envt
    none = .false.
    nthreads = 1
    !$ nthreads = min(recorder_threads_max(), omp_get_max_threads())      ! (every data point is recorded here, not a sample: more threads than discover() takes)
    call get_environment_variable('GADFIT_HIP_RECORD_THREADS', envt, status=stat)
    if (stat == 0) read(envt, *, iostat=stat) nthreads
    nthreads = max(1, nthreads)
    if (opt_eval_one_thread) nthreads = 1               ! (gadf_init(..., eval_is_thread_safe=.false.))
    do round = 1, 16
       ncol = n_aux_total
       if (hint_col >= 0) ncol = ncol + 1 + n_set_cols
       if (ncol == 0) exit
       ! (every path compares AD variables, no literal needs a column: all the device reads are the set columns, which cross_check has
       ! filled -- nothing to record here)
       if (hint_col >= 0 .and. n_set_cols > 0 .and. n_aux_total == 0 .and. .not. natural_col_used) then
          if (allocated(tab)) deallocate(tab)
          allocate(tab(size(xs), ncol))
          tab(:, hint_col + 1) = -1.0_c_double
          call fill_set_columns()
          call lib_check(gfh_set_aux(tgt, int(ncol, c_int), tab), __FILE__, __LINE__)
          tabulated = .true.
          return
       end if
       if (allocated(tab)) deallocate(tab)
       allocate(tab(size(xs), ncol))       ! (every row is written below: by the threads, or zeroed and filled by the serial loop)
       if (allocated(done)) deallocate(done)
       allocate(done(size(xs))); done = .false.
       grew = .false.
       ! Straight-line paths only (the usual model with real(kp) arithmetic on x; an eval() that branches on the plain real x) and
       ! many points: every point's path and per-point inputs are read off recordings made in checking mode on several threads, as
       ! discover() checks its sample (module ad, ad_thread_check; the known paths side by side in ad_tls.c, the values of the class-3
       ! literals back through gfh_adchk_aux).  A point whose recording follows none of the paths is left to the serial loop below.
       ! eval() is called concurrently here (GADFIT_HIP_RECORD_THREADS=1: never).
       par_ok = nthreads > 1 .and. .not. eval_serial_only .and. n_paths >= 1 .and. n_paths <= 16 .and. size(xs) >= threads_from() .and. ncol <= 256
       any_guards = .false.
       if (par_ok) then
          do q = 1, n_paths
             associate(p => paths(q))
               par_ok = par_ok .and. p%n_seen >= 2 .and. p%n_guards <= 64 .and. p%n_aux <= 64
               any_guards = any_guards .or. p%sub_guards
               any_guards = any_guards .or. p%n_guards > 0
             end associate
          end do
       end if
       call get_environment_variable('GADFIT_HIP_SETUP_TIMES', envt, status=stat)
       if (stat == 0) then
          if (trim(adjustl(envt)) == '3' .and. .not. par_ok) write(error_unit, '(a, i0, a, i0, a, 16(1x, i0, "/", i0, "/", i0, "/", i0, "/", i0))') &
               & 'serial tabulation: ', nthreads, ' threads, ', n_paths, ' paths (n_seen/nsub/nint/n_guards/n_aux):', &
               & (paths(q)%n_seen, paths(q)%nsub, paths(q)%nint, paths(q)%n_guards, paths(q)%n_aux, q = 1, min(n_paths, 16))
       end if
       if (par_ok) then
          nmax = maxval(paths(1:n_paths)%n)
          if (allocated(k_op)) deallocate(k_op, k_a, k_b, k_fl, k_cls, k_c, k_al, k_be)
          allocate(k_op(nmax, n_paths), k_a(nmax, n_paths), k_b(nmax, n_paths), k_fl(nmax, n_paths), k_cls(nmax, n_paths), &
               & k_c(nmax, n_paths), k_al(nmax, n_paths), k_be(nmax, n_paths))
          if (allocated(sbits)) deallocate(sbits)
          allocate(sbits(n_paths)); sbits = 0_c_int64_t
          do q = 1, n_paths
             associate(p => paths(q))
               pn = p%n
               k_op(1:pn, q) = p%raw(1:pn)%op; k_a(1:pn, q) = p%raw(1:pn)%a; k_b(1:pn, q) = p%raw(1:pn)%b; k_fl(1:pn, q) = p%raw(1:pn)%flags
               k_cls(1:pn, q) = p%lit_class(1:pn); k_c(1:pn, q) = p%lit_c(1:pn); k_al(1:pn, q) = p%lit_alpha(1:pn); k_be(1:pn, q) = p%lit_beta(1:pn)
               call gfh_adchk_load_path(int(q - 1, c_int), int(pn, c_int), k_op(:, q), k_a(:, q), k_b(:, q), k_fl(:, q), k_cls(:, q), &
                    & k_c(:, q), k_al(:, q), k_be(:, q))
               if (p%nsub > 0) call load_check_ints(q - 1, p)
               do j = 1, p%n_guards
                  if (p%script(j)) sbits(q) = ibset(sbits(q), j - 1)
               end do
             end associate
          end do
          npth = n_paths; hc = hint_col
          n_racy = 0
          call system_clock(tk0, tkr)
          ! (two passes: the first writes, the second must find the same bits again -- an eval() that keeps state in saved or module
          ! variables gives itself away by answers that change from one concurrent call to the next)
          do pass = 1, 2      ! (also on the refreshes of on_pars, round 6: a race that the first tabulation happened not to show must not
                              ! slip into a column later -- one pass alone has nothing to be held against but 64 serial recordings)
          do d = 1, size(fitfuncs)
             if (data_positions(d + 1) <= data_positions(d)) cycle
             np_ = size(fitfuncs(d)%pars)
             do k = 1, np_
                call set_node(fitfuncs(d)%pars(k), k - 1)
             end do
             ! (comparisons of AD variables on some path: the values are computed, so that a recording can find its own way; and where
             ! a column follows the parameters: the real may be formed from the %val of an INTERMEDIATE AD variable -- t = p*x;
             ! s = cos(t%val) -- which a check that skips the values leaves at 0)
             ad_recording = .true.; ad_thread_check = .true.; ad_need_vals = any_guards .or. n_follow > 0; ad_cur = 0; ad_fast_check = .not. ad_need_vals
             !$omp parallel default(shared) num_threads(nthreads) private(i, cn, cdiv, clit, res, got, vals, nodes, j, q, r, mine, tried, same, row, found) &
             !$omp & reduction(+:n_racy)
             allocate(row(ncol))
             mine = 1
             !$omp do schedule(static)
             do i = data_positions(d) + 1, data_positions(d + 1)
                if (pass == 2 .and. .not. done(i)) cycle
                ! the point's own path: the recording decides its comparisons by the values and must agree with one of the known paths
                found = .false.
                call gfh_adchk_script(0_c_int, 0_c_int64_t)
                do tried = 0, npth - 1
                   q = mod(mine - 1 + tried, npth) + 1
                   call gfh_adchk_use(int(q - 1, c_int))
                   call check_one(d, xs(i), np_, cn, cdiv, clit, res)
                   if (cdiv /= 0 .or. clit /= 0 .or. cn /= paths(q)%n .or. res /= paths(q)%res_node) cycle
                   got = gfh_adchk_aux(64_c_int, vals, nodes)
                   if (got /= paths(q)%n_aux) cycle
                   if (got > 0) then
                      if (any(nodes(1:got) + 1 /= paths(q)%aux_raw_k(1:got))) cycle
                   end if
                   found = .true.
                   exit
                end do
                if (.not. found) then
                   if (pass == 2) n_racy = n_racy + 1              ! (followed a path in the first pass, none now)
                   cycle
                end if
                mine = q
                row = 0.0_c_double
                if (hc >= 0) row(hc + 1) = real(q - 1, c_double)
                do j = 1, got
                   row(paths(q)%aux0 + j) = vals(j)
                end do
                ! the per-point inputs of the OTHER paths the point may come to take while the parameters move: recorded along their
                ! comparisons (forced), as the serial tabulation does; a point that can never be on such a path diverges from it
                do r = 1, npth
                   if (r == q .or. paths(r)%n_aux == 0 .or. paths(r)%n_guards == 0) cycle
                   call gfh_adchk_script(int(paths(r)%n_guards, c_int), sbits(r))
                   call gfh_adchk_use(int(r - 1, c_int))
                   call check_one(d, xs(i), np_, cn, cdiv, clit, res)
                   if (cdiv /= 0 .or. cn /= paths(r)%n .or. res /= paths(r)%res_node) cycle
                   got = gfh_adchk_aux(64_c_int, vals, nodes)
                   if (got /= paths(r)%n_aux) cycle
                   if (any(nodes(1:got) + 1 /= paths(r)%aux_raw_k(1:got))) cycle
                   do j = 1, got
                      row(paths(r)%aux0 + j) = vals(j)
                   end do
                end do
                if (pass == 1) then
                   tab(i, :) = row
                   done(i) = .true.
                else
                   same = .true.
                   do j = 1, ncol
                      same = same .and. (tab(i, j) == row(j) .or. (row(j) /= row(j) .and. tab(i, j) /= tab(i, j)))
                   end do
                   if (.not. same) n_racy = n_racy + 1
                end if
             end do
             !$omp end do
             call gfh_adchk_script(0_c_int, 0_c_int64_t)
             call gfh_adchk_use(0_c_int)
             deallocate(row)
             !$omp end parallel
             ad_recording = .false.; ad_thread_check = .false.; ad_fast_check = .false.; ad_need_vals = .true.
             do k = 1, np_
                call set_node(fitfuncs(d)%pars(k), -1)
             end do
          end do
          end do
          call get_environment_variable('GADFIT_HIP_SETUP_TIMES', envt, status=stat)
          if (stat == 0) then
             call system_clock(tk1, tkr)
             if (trim(adjustl(envt)) == '3') write(error_unit, '(a, i0, a, i0, a, i0, a, i0, a, f10.3, a)') 'threaded tabulation: ', count(done), ' of ', size(done), &
                  & ' points on ', nthreads, ' threads (', n_racy, ' disagreements between its two passes)', 1e3*real(tk1 - tk0)/real(tkr), ' ms'
          end if
          ! what the threads read off is spot-checked against serial recordings (64 points per dataset): an eval() that keeps state
          ! in saved or module variables may have produced columns that follow nobody's path -- then everything is done again serially
          ! -- and, on top of those fixed points, 1 % of the points drawn afresh at every fit (a schedule no two fits share: a leak that
          ! the strided points happen to miss does not stay missed)
          racy = n_racy > 0
          call system_clock(tk0, tkr)
          do d = 1, size(fitfuncs)
             if (data_positions(d + 1) <= data_positions(d) .or. racy) cycle
             nd_pts = data_positions(d + 1) - data_positions(d)
             stride = max(1_c_int64_t, nd_pts/64)
             n_spot = (nd_pts + stride - 1)/stride
             n_draw = merge(0_c_int64_t, nd_pts/100, refreshing)      ! (a refresh before every pass: the 64 fixed points)
             do k_spot = 1, n_spot + n_draw
                if (k_spot <= n_spot) then
                   i = data_positions(d) + 1 + (k_spot - 1)*stride
                else       ! (a generator of the layer's own, seeded from the clock once per process: the program's random_number is left alone)
                   if (draw_state == 0) then
                      call system_clock(draw_state)
                      draw_state = ior(draw_state, 1_c_int64_t)
                   end if
                   draw_state = ieor(draw_state, ishft(draw_state, 13)); draw_state = ieor(draw_state, ishft(draw_state, -7))
                   draw_state = ieor(draw_state, ishft(draw_state, 17))
                   i = data_positions(d) + 1 + modulo(draw_state, nd_pts)
                end if
                if (.not. done(i)) cycle
                call record(d, xs(i), 0, none, res)
                q = find_path(res)
                if (q == 0) then
                   if (refreshing) cycle        ! (a turn no recording covers at the parameters of this pass: the device reports it)
                   racy = .true.; exit
                end if
                if (hint_col >= 0) then
                   if (tab(i, hint_col + 1) /= real(q - 1, c_double)) racy = .true.
                end if
                do j = 1, paths(q)%n_aux
                   if (tab(i, paths(q)%aux0 + j) /= ad_tape(paths(q)%aux_raw_k(j))%c .and. &
                        & .not. (tab(i, paths(q)%aux0 + j) /= tab(i, paths(q)%aux0 + j) .and. ad_tape(paths(q)%aux_raw_k(j))%c /= ad_tape(paths(q)%aux_raw_k(j))%c)) racy = .true.
                end do
                if (racy) exit
             end do
          end do
          call get_environment_variable('GADFIT_HIP_SETUP_TIMES', envt, status=stat)
          if (stat == 0) then
             call system_clock(tk1, tkr)
             if (trim(adjustl(envt)) == '3') write(error_unit, '(a, f10.3, a)') 'serial re-verification of the threaded columns (64 fixed points + 1 % drawn per dataset): ', &
                  & 1e3*real(tk1 - tk0)/real(tkr), ' ms'
          end if
          if (racy) then
             call warning(__FILE__, __LINE__, 'eval() gave other values when called from several threads than when called alone: &
                  &it seems to keep state in saved or module variables. Its per-point columns are tabulated serially; set &
                  &GADFIT_HIP_RECORD_THREADS=1 to skip the attempt.')
             done = .false.; nthreads = 1; eval_serial_only = .true.
          end if
       end if
       do d = 1, size(fitfuncs)
          do i = data_positions(d) + 1, data_positions(d + 1)
             if (done(i)) cycle
             tab(i, :) = 0.0_c_double
             call record(d, xs(i), 0, none, res)
             q = find_path(res)
             ! (on_pars -- the columns at the parameters of a pass: nothing is learnt and the model stays as it is; a point that takes
             ! a turn no recording covers at these parameters is reported by the device, on_unseen then tabulates at the same parameters)
             if (q == 0 .and. .not. refreshing) then
                call add_path(d, res); q = n_paths; grew = .true.
             end if
             row_c = ad_tape(1:ad_tape_n)%c                ! (this recording's literals: observe may record again)
             if (.not. refreshing) call observe(paths(q), xs(i), d)
             if (grew) cycle                               ! (the columns are laid out again once the new path is known)
             if (hint_col >= 0) tab(i, hint_col + 1) = real(q - 1, c_double)
             if (q > 0) then
             do j = 1, paths(q)%n_aux
                tab(i, paths(q)%aux0 + j) = row_c(paths(q)%aux_raw_k(j))
             end do
             ! (recordings that share eval()'s path and differ inside an integrand are one variant on the device, which reads the
             ! columns of whichever of them came first: all of them get the values)
             do r = 1, n_paths
                if (r == q .or. .not. eval_twins(paths(r), paths(q))) cycle
                do j = 1, min(paths(q)%n_aux, paths(r)%n_aux)
                   tab(i, paths(r)%aux0 + j) = row_c(paths(q)%aux_raw_k(j))
                end do
             end do
             end if
             do r = 1, n_paths
                if (r == q .or. paths(r)%n_aux == 0) cycle
                call record(d, xs(i), paths(r)%n_guards, paths(r)%script, res)
                if (.not. same_as(paths(r), res)) cycle     ! (this point can never be on that path)
                do j = 1, paths(r)%n_aux
                   tab(i, paths(r)%aux0 + j) = ad_tape(paths(r)%aux_raw_k(j))%c
                end do
             end do
          end do
       end do
       if (.not. grew) then
          ! a literal found here to be neither constant nor affine after all changes the columns too
          do q = 1, n_paths
             if (paths(q)%n_aux /= count(paths(q)%raw%op == GFH_CONST .and. paths(q)%lit_class == 3)) grew = .true.
          end do
       end if
       if (.not. grew) then
          call fill_set_columns()
          tabulated = .true.
          if (tab_hold) then                     ! (tabulate_all collects the sets of the forward differences: no upload from here)
             call move_alloc(tab, tab_keep)
             return
          end if
          call lib_check(gfh_set_aux(tgt, int(ncol, c_int), tab), __FILE__, __LINE__)
          if (n_follow > 0) then                 ! (kept: the hook of another handle at the same parameters uploads it as it is)
             call move_alloc(tab, tab_keep)
             if (allocated(tab_keep_pars)) deallocate(tab_keep_pars)
             allocate(tab_keep_pars(0))
             do d = 1, size(fitfuncs)
                tab_keep_pars = [tab_keep_pars, fitfuncs(d)%pars%val]
             end do
             tab_serial = tab_serial + 1
             call mark_uploaded(tgt)
          end if
          return
       end if
       do q = 1, n_paths
          call probe_pars(paths(q)); call probe_theta(paths(q))
          call probe_abscissas(paths(q))
       end do
       call upload_model(tgt)
    end do
    if (need_tab) call error(__FILE__, __LINE__, 'eval() keeps taking new paths while its per-point columns are tabulated.')
    tabulated = .true.
  contains
    ! the columns of the sets of outcomes (plan_hint_columns): the path cross_check saw every point take under them, 0-based
    subroutine fill_set_columns()
      integer :: c
      do c = 1, n_set_cols
         tab(:, hint_col + 1 + c) = real(int(cross_q(:, set_of_col(c))) - 1, c_double)
      end do
    end subroutine fill_set_columns
  end subroutine tabulate

  ! do two recordings follow the same path through eval() itself (they may differ inside their integrands)?
  logical function eval_twins(a, b) result(same)
    type(path_t), intent(in) :: a, b
    integer :: ja, jb
    same = .false.
    if (.not. (a%sub_guards .or. b%sub_guards)) return
    if (a%cnt(0) /= b%cnt(0) .or. a%n_guards /= b%n_guards .or. a%n_aux /= b%n_aux) return
    ja = 0; jb = 0
    do
       ja = ja + 1
       do while (ja <= a%n)
          if (a%psub(ja) == 0) exit
          ja = ja + 1
       end do
       jb = jb + 1
       do while (jb <= b%n)
          if (b%psub(jb) == 0) exit
          jb = jb + 1
       end do
       if (ja > a%n .or. jb > b%n) exit
       if (a%raw(ja)%op /= b%raw(jb)%op .or. a%raw(ja)%flags /= b%raw(jb)%flags) return
       if (a%raw(ja)%op /= GFH_INTEGRATE) then
          if (a%raw(ja)%a /= b%raw(jb)%a .or. a%raw(ja)%b /= b%raw(jb)%b) return
       end if
    end do
    same = ja > a%n .and. jb > b%n
  end function eval_twins

  ! gfh_unseen_handler (include/gadfit_hip.h): data points took a turn through eval() that no recorded path covers -- a
  ! comparison came out the other way for the first time at the parameters of the pass.  eval() is recorded at those points
  ! along the outcomes the device saw; new paths join the model, which is handed to `target` again (and the per-point
  ! columns with it); the library then repeats the pass.
  integer(c_int) function on_unseen(user, target, n, index, dataset, x, path, n_guards, pars) bind(c) result(rc)
    type(c_ptr), value :: user, target
    integer(c_int), value :: n
    integer(c_int64_t), intent(in) :: index(*), path(*)
    integer(c_int32_t), intent(in) :: dataset(*), n_guards(*)
    real(c_double), intent(in) :: x(*), pars(*)
    real(kp), allocatable :: saved(:,:)
    logical :: script(64), grew
    integer :: k, d, j, np, npl, res, q, ng, nbefore, trace_stat
    character(len=8) :: trace_env
    rc = 0
    np = size(fitfuncs(1)%pars)
    npl = np + n_plit_cap                        ! (the library's block carries the pseudo-parameters of lit_class 4 behind the model's own)
    allocate(saved(np, size(fitfuncs)))
    do d = 1, size(fitfuncs)
       saved(:, d) = fitfuncs(d)%pars%val
       call set_vals(fitfuncs(d)%pars, pars((d-1)*npl + 1 : (d-1)*npl + np))
    end do
    grew = .false.
    at_capture_pars = .false.                    ! (the recordings below are made at the parameters of this pass)
    call get_environment_variable('GADFIT_HIP_TRACE_PATHS', trace_env, status=trace_stat)
    if (trace_stat == 0) write(error_unit, '(a, i0, a, 8es14.6)') 'on_unseen: ', n, ' point(s); parameters of the pass (dataset 1): ', pars(1:min(8, np))
    if (n == 0) then
       ! an integrand met a path through its comparisons that no recording has (the parameters have moved since they were made):
       ! the integrands are recorded again over the sample, at the parameters of this pass
       nbefore = n_paths
       call explore_integrands()
       grew = n_paths > nbefore
       ! (a member of a device group may come here after another member has had the paths recorded: its own model still lacks them)
       if (.not. grew .and. gfh_model_n_tapes(target) >= n_paths) then
          rc = 1
          do d = 1, size(fitfuncs)
             call set_vals(fitfuncs(d)%pars, saved(:, d))
          end do
          at_capture_pars = .true.
          return
       end if
    end if
    do k = 1, n
       d = dataset(k) + 1
       ng = min(int(n_guards(k)), 64)
       do j = 1, ng
          script(j) = btest(path(k), j - 1)
       end do
       call record(d, x(k), ng, script, res)
       q = find_path(res)
       if (q == 0) then
          call add_path(d, res); q = n_paths; grew = .true.
       end if
       call observe(paths(q), x(k), d)
    end do
    if (grew) then                                ! (the outcomes first met in this pass: forced on eval() at every data point)
       nbefore = n_paths
       call cross_check()
    end if
    ! (a member of a device group may meet a path that another member has had recorded already: its own model still lacks it)
    if (grew .or. hint_col >= 0 .or. gfh_model_n_tapes(target) < n_paths) then
       do q = 1, n_paths
          call probe_pars(paths(q)); call probe_theta(paths(q))
          call probe_abscissas(paths(q))
       end do
       call upload_model(target)
       if (need_tab) call tabulate_all(target)
    else
       rc = 1                    ! the recordings follow paths the device already has: nothing to add
    end if
    do d = 1, size(fitfuncs)
       call set_vals(fitfuncs(d)%pars, saved(:, d))
    end do
    at_capture_pars = .true.
  end function on_unseen

  ! tabulate, and under use_ad = .false. with columns that follow the parameters the sets of the forward differences behind it: eval()
  ! over all data points once more per active parameter, at p + step e_j with the reference's step (fitfunction.F90:161-168:
  ! sqrt(epsilon)*p, taken as (p + step) - p -- the device forms the same number), every dataset at its own values.
  subroutine tabulate_all(tgt)
    type(c_ptr), intent(in) :: tgt
    real(c_double), allocatable :: big(:,:)
    real(kp), allocatable :: base(:,:)
    real(kp) :: v
    integer :: d, j, k, ncol, na, np
    logical :: was_refreshing, was_capture
    if (.not. (finite_differences .and. n_follow > 0)) then
       call tabulate(tgt); return
    end if
    tab_hold = .true.
    call tabulate(tgt)                           ! (may still learn and lay the columns out anew; what it leaves is set 0)
    if (.not. (finite_differences .and. n_follow > 0) .or. .not. allocated(tab_keep)) then
       tab_hold = .false.; tabulated = .false.
       call tabulate(tgt); return                ! (the model it ended with has no such column after all)
    end if
    ncol = size(tab_keep, 2); np = size(fitfuncs(1)%pars); na = count(active_pars /= 0)
    allocate(big(size(tab_keep, 1), ncol*(1 + na)), base(np, size(fitfuncs)))
    big(:, 1:ncol) = tab_keep
    do d = 1, size(fitfuncs)
       base(:, d) = fitfuncs(d)%pars%val
    end do
    was_refreshing = refreshing; was_capture = at_capture_pars
    refreshing = .true.; at_capture_pars = .false.
    k = 0
    do j = 1, np
       if (active_pars(j) == 0) cycle
       k = k + 1
       do d = 1, size(fitfuncs)
          v = base(j, d)
          call set_vals(fitfuncs(d)%pars, [base(:j-1, d), v + sqrt(epsilon(1.0_kp))*v, base(j+1:, d)])
       end do
       call tabulate(tgt)
       big(:, k*ncol + 1 : (k + 1)*ncol) = tab_keep
       do d = 1, size(fitfuncs)
          call set_vals(fitfuncs(d)%pars, base(:, d))
       end do
    end do
    refreshing = was_refreshing; at_capture_pars = was_capture
    tab_hold = .false.
    call lib_check(gfh_set_aux(tgt, int(size(big, 2), c_int), big), __FILE__, __LINE__)
    call move_alloc(big, tab_keep)
    if (allocated(tab_keep_pars)) deallocate(tab_keep_pars)
    allocate(tab_keep_pars(0))
    do d = 1, size(fitfuncs)
       tab_keep_pars = [tab_keep_pars, base(:, d)]
    end do
    tab_serial = tab_serial + 1
    call mark_uploaded(tgt)
  end subroutine tabulate_all

  ! which library handles hold the table of the latest tabulation (tab_serial)?
  logical function is_uploaded(tgt) result(yes)
    type(c_ptr), intent(in) :: tgt
    integer :: k
    yes = .false.
    do k = 1, n_up
       if (c_associated(up_tgt(k), tgt)) then
          yes = up_serial(k) == tab_serial; return
       end if
    end do
  end function is_uploaded

  subroutine mark_uploaded(tgt)
    type(c_ptr), intent(in) :: tgt
    integer :: k
    do k = 1, n_up
       if (c_associated(up_tgt(k), tgt)) then
          up_serial(k) = tab_serial; return
       end if
    end do
    if (n_up == size(up_tgt)) n_up = 0           ! (more handles than anybody has devices: start over, at worst an upload too many)
    n_up = n_up + 1
    up_tgt(n_up) = tgt; up_serial(n_up) = tab_serial
  end subroutine mark_uploaded

  ! gfh_pars_hook (include/gadfit_hip.h): before every pass the reals that eval() forms from the %val of fitted parameters
  ! (lit_class 4) are recomputed at the parameters of the pass, as the reference recomputes them whenever eval() runs
  ! (gadfit.F90:679-690): per dataset, every path that has such reals is recorded once at its first abscissa (its comparisons forced)
  ! and the values are written into the pseudo-parameters behind the model's own.
  integer(c_int) function on_pars(user, target, pars) bind(c) result(rc)
    type(c_ptr), value :: user, target
    real(c_double), intent(in out) :: pars(*)
    real(kp), allocatable :: saved(:), saved_all(:,:), blk(:)
    integer :: d, q, j, np, npl, res
    logical :: fresh
    rc = 0
    np = size(fitfuncs(1)%pars); npl = np + n_plit_cap
    ! Columns that follow the parameters AND the abscissa (lit_follow): tabulated anew -- eval() at every data point, as the reference
    ! does in every pass -- unless the table at hand was made at exactly these parameters (the sweep of an accepted step after the
    ! trial chi2() there, STEP 3 after the sweep, another member of a device group: then at most an upload)
    if (n_follow > 0) then
       allocate(blk(np*size(fitfuncs)))
       do d = 1, size(fitfuncs)
          blk((d-1)*np + 1 : d*np) = pars((d-1)*npl + 1 : (d-1)*npl + np)
       end do
       fresh = allocated(tab_keep_pars) .and. allocated(tab_keep)
       if (fresh) fresh = size(tab_keep_pars) == size(blk)
       if (fresh) fresh = all(tab_keep_pars == blk)
       if (.not. fresh) then
          allocate(saved_all(np, size(fitfuncs)))
          do d = 1, size(fitfuncs)
             saved_all(:, d) = fitfuncs(d)%pars%val
             call set_vals(fitfuncs(d)%pars, blk((d-1)*np + 1 : d*np))
          end do
          at_capture_pars = .false.; refreshing = .true.
          call tabulate_all(target)
          refreshing = .false.; at_capture_pars = .true.
          do d = 1, size(fitfuncs)
             call set_vals(fitfuncs(d)%pars, saved_all(:, d))
          end do
          n_retabulated = n_retabulated + 1
       else if (.not. is_uploaded(target)) then
          call lib_check(gfh_set_aux(target, int(size(tab_keep, 2), c_int), tab_keep), __FILE__, __LINE__)
          call mark_uploaded(target)
       end if
    end if
    do d = 1, size(fitfuncs)
       saved = fitfuncs(d)%pars%val
       call set_vals(fitfuncs(d)%pars, pars((d-1)*npl + 1 : (d-1)*npl + np))
       do q = 1, n_paths
          if (paths(q)%n_plit == 0) cycle
          ad_theta = paths(q)%theta
          call record(d, paths(q)%x1, paths(q)%n_guards, paths(q)%script, res)
          ad_theta = 0.5_kp
          if (.not. same_as(paths(q), res)) then
             if (paths(q)%sub_guards) cycle      ! (an integrand's own comparison came out differently: the values of before stay)
             rc = 1
          else
             do j = 1, paths(q)%n_plit
                pars((d-1)*npl + np + paths(q)%plit_slot(paths(q)%plit_raw_k(j)) + 1) = ad_tape(paths(q)%plit_raw_k(j))%c
             end do
          end if
       end do
       call set_vals(fitfuncs(d)%pars, saved)
    end do
  end function on_pars

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
    integer(c_int64_t) :: clk(7), clk_rate
    logical :: want_lb, uploading
    character(len=8) :: env_lb
    if (.not. allocated(fitfuncs)) call error(__FILE__, __LINE__, &
         & 'Number of datasets is undetermined. Call gadf_init first.')
    call system_clock(clk(1), clk_rate)
    finite_differences = .false.
    if (present(use_ad)) finite_differences = .not. use_ad
    accel_requested = .false.
    if (present(accth)) accel_requested = accth > 0
    if (.not. allocated(x_data)) call read_data()
    call system_clock(clk(2))
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
    ! The data start on their way to the device (the copies run on a thread of the library) while eval() is recorded over
    ! them here on the host; the next library call waits for the upload.
    uploading = .not. data_uploaded
    if (uploading .and. compile_only) then          ! (GADFIT_HIP_DEVICE=-1: capture and kernel generation without a GPU; the upload is what stops)
       call discover()
       call system_clock(clk(3))
       call get_environment_variable('GADFIT_HIP_SETUP_TIMES', env_lb, status=stat_lb)
       if (stat_lb == 0) write(error_unit, '(a, f9.3, a, i0, a)') 'record eval() over the data [ms]: ', ms(2, 3), '  (', n_paths, ' path(s))'
       call upload_model(ctx)
       model_captured = .true.
       ! ... and the kernels of this model and active set are compiled into the kernel cache (what build() runs the test
       ! programs for: a GPU box then loads them instead of compiling for half a second)
       n_act = count(active_pars /= 0)
       if (n_act > 0) then
          allocate(act(n_act))
          j = 0
          do i = 1, size(active_pars)
             if (active_pars(i) /= 0) then
                j = j + 1
                act(j) = i - 1
             end if
          end do
          call lib_check(gfh_model_prepare(ctx, int(n_act, c_int), act), __FILE__, __LINE__)
          deallocate(act)
       end if
    end if
    if (uploading) then
       if (.not. associated(up_y) .or. .not. associated(up_w)) call error(__FILE__, __LINE__, 'internal: no data to upload')
       if (x_copy_pending) call lib_check(gfh_queue_host_copy(ctx, c_loc(x_data), c_loc(xs), &
            & int(size(xs), c_int64_t)*int(storage_size(1.0_kp)/8, c_int64_t)), __FILE__, __LINE__)
       call lib_check(gfh_set_data_begin(ctx, int(size(xs), c_int64_t), xs, up_y, up_w, &
            & int(size(fitfuncs), c_int), data_positions), __FILE__, __LINE__)
    end if
    if (model_captured .and. allocated(cap_vals)) then
       ! a passive parameter has another value than at the capture, or the active set has changed: eval() may have read such a
       ! parameter's %val into plain real arithmetic (which the capture baked in), so it is recorded again
       if (any(cap_active /= active_pars)) model_captured = .false.
       ! (reals that follow the fitted parameters are carried differently under AD -- pseudo-parameters -- and under finite
       ! differences -- sets of columns: a model captured for the one is captured again for the other)
       if ((n_plit_total > 0 .or. n_follow > 0) .and. (finite_differences .neqv. cap_fd)) model_captured = .false.
       do i = 1, size(fitfuncs)
          if (any(active_pars == 0 .and. fitfuncs(i)%pars%val /= cap_vals(:, i))) model_captured = .false.
       end do
       ! ... or eval() reads something else that has changed since (a module variable of the user's): every path is recorded once
       ! more at its first abscissa and must come out as it did, literal for literal
       if (model_captured) then
          if (.not. capture_is_current()) model_captured = .false.
       end if
       if (.not. model_captured) then
          if (x_on_device_or_borrowed()) call own_x()
          tabulated = .false.
       end if
    end if
    if (.not. model_captured) then
       call discover()
       call system_clock(clk(3))
       call upload_model(ctx)
       model_captured = .true.; cap_fd = finite_differences
       if (allocated(cap_vals)) deallocate(cap_vals, cap_active)
       allocate(cap_vals(size(fitfuncs(1)%pars), size(fitfuncs)), cap_active(size(active_pars)))
       cap_active = active_pars
       do i = 1, size(fitfuncs)
          cap_vals(:, i) = fitfuncs(i)%pars%val
       end do
    else
       clk(3) = clk(2)
    end if
    call system_clock(clk(4))
    call get_environment_variable('GADFIT_HIP_SETUP_TIMES', env_lb, status=stat_lb)
    if (stat_lb == 0) then
       if (trim(adjustl(env_lb)) == '2') write(error_unit, '(a, f9.3, a, i0, a)') 'record eval() over the data [ms]: ', ms(2, 3), '  (', n_paths, ' path(s))'
    end if
    if (uploading) then
       call lib_check(gfh_init_weights(ctx, int(data_error_type, c_int)), __FILE__, __LINE__)  ! gadfit.F90:445-470 (waits for the upload)
       ! (x_data is being filled on a thread of the library; the user's abscissas are read in place until this fit is over)
       copy_in_flight = x_copy_pending
       data_uploaded = .true.
       tabulated = .false.
    end if
    call system_clock(clk(5))
    if (need_tab .and. .not. tabulated) call tabulate_all(ctx)
    call system_clock(clk(6))
    ! compact the active list (gadfit.F90:586-599), 0-based for the library
    np = size(fitfuncs(1)%pars)
    n_act = count(active_pars /= 0)
    if (n_act == 0) call error(__FILE__, __LINE__, 'There are no active parameters.')
    if (set_count < size(fitfuncs)*np) call warning(__FILE__, __LINE__, 'Some parameters might be uninitialized.')
    allocate(act(n_act), glob(np + n_plit_cap), pars(np + n_plit_cap, size(fitfuncs)))      ! (+ the passive pseudo-parameters of lit_class 4: on_pars)
    j = 0
    do i = 1, np
       if (active_pars(i) /= 0) then
          j = j + 1
          act(j) = i - 1
       end if
    end do
    glob = 0
    glob(:np) = merge(1, 0, is_global)
    pars = 0.0_c_double
    do i = 1, size(fitfuncs)
       pars(:np, i) = fitfuncs(i)%pars%val
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
    fit_in_progress = .true.
    call lib_check(gfh_fit(ctx, pars, int(n_act, c_int), act, glob, o, r), __FILE__, __LINE__)
    fit_in_progress = .false.
    if (copy_in_flight) then
       call lib_check(gfh_wait_host_copy(ctx), __FILE__, __LINE__)
       xs => x_data; x_copy_pending = .false.; copy_in_flight = .false.
    end if
    call system_clock(clk(7))
    ! GADFIT_HIP_SETUP_TIMES=1: where a gadf_fit call spends its time on the host clock (stderr)
    call get_environment_variable('GADFIT_HIP_SETUP_TIMES', env_lb, status=stat_lb)
    if (stat_lb == 0) write(error_unit, '(a, 6(a, f9.3), a, f9.3, a)') 'gadf_fit [ms]:', ' read_data', ms(1, 2), '  record eval() over the data', &
         & ms(2, 3), '  model to the library', ms(3, 4), '  wait for the upload + weights', ms(4, 5), '  per-point columns', ms(5, 6), &
         & '  options + gfh_fit', ms(6, 7), '  (of which the LM loop', 1e3*r%seconds, ')'
    if (stat_lb == 0 .and. n_follow > 0) write(error_unit, '(a, i0, a, i0, a)') 'gadf_fit: ', n_follow, ' per-point column(s) follow the fitted &
         &parameters (%val together with x); tabulated anew by eval() over all data points ', n_retabulated, ' time(s) so far'
    gadf_iterations = r%iterations
    last_n_omega = r%n_omega; last_seconds = r%seconds
    if (show_timings) call print_device_timings(r)
    umnigh_a = o%umnigh_a
    do i = 1, size(fitfuncs)
       call set_vals(fitfuncs(i)%pars, pars(:np, i))
    end do
    gadf_iterations = r%iterations
    gadf_chi2 = r%chi2
  contains
    real(kp) function ms(a, b)
      integer, intent(in) :: a, b
      ms = 1e3_kp*real(clk(b) - clk(a), kp)/real(clk_rate, kp)
    end function ms
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
       call own_x()
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
       call own_x()
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
             write(u, '(i0, 1x, i0, 1x, a, 1x, es25.17)') i, j, par_name(fitfuncs(i), j), fitfuncs(i)%pars(j)%val      ! dataset, parameter, name, value
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
    integer :: i
    if (c_associated(ctx)) call gfh_destroy(ctx)
    ctx = c_null_ptr
    if (allocated(fitfuncs)) deallocate(fitfuncs)
    if (allocated(active_pars)) deallocate(active_pars)
    if (allocated(is_global)) deallocate(is_global)
    if (allocated(x_data)) deallocate(x_data)
    nullify(up_y, up_w, xs)
    x_copy_pending = .false.; copy_in_flight = .false.
    if (allocated(y_data)) deallocate(y_data)
    if (allocated(weights)) deallocate(weights)
    if (allocated(data_positions)) deallocate(data_positions)
    if (allocated(data_pointers)) then
       do i = 1, size(data_pointers)
          if (.not. data_pointers(i)%owned) cycle
          if (associated(data_pointers(i)%x_data)) deallocate(data_pointers(i)%x_data)
          if (associated(data_pointers(i)%y_data)) deallocate(data_pointers(i)%y_data)
          if (associated(data_pointers(i)%weights)) deallocate(data_pointers(i)%weights)
       end do
       deallocate(data_pointers)
    end if
    if (allocated(paths)) deallocate(paths)
    if (allocated(cap_vals)) deallocate(cap_vals, cap_active)
    if (allocated(tab_keep)) deallocate(tab_keep)
    if (allocated(tab_keep_pars)) deallocate(tab_keep_pars)
    n_paths = 0; n_follow = 0; n_up = 0
  end subroutine gadf_close
end module gadfit
