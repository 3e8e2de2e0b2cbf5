! Small host-side helpers under the reference's names (fortran/gadfit/misc.F90:23, 28-71, 176-264), so that user code
! written against them compiles unchanged: type(string) -- a deferred-length name with assignment from and comparison
! with character, and len() -- which fitfunc%parnames is made of (fitfunction.F90:34); safe_deallocate, a generic that
! deallocates only what is allocated and reports a failure through messaging's check_err (modules ad and fitfunction
! extend it for advar and fitfunc arrays, AD:92-96, fitfunction.F90:66-70); safe_close; swap; the timer and
! data_pointer types.  The coarray sums of misc.F90:133-170 are what libgadfit_hip's all-reduce replaces: not here.
module misc
  use, intrinsic :: iso_fortran_env, only: int64
  use gadf_constants, only: dp, kp, qp
  use messaging, only: check_err, err_stat, err_msg
  implicit none
  private
  public :: data_pointer, string, len, swap, timer, safe_deallocate, safe_close

  type :: string
     character(:), allocatable :: name
   contains
     procedure :: string_from_character
     generic :: assignment(=) => string_from_character
     procedure :: string_is_character
     generic :: operator(==) => string_is_character
  end type string

  ! accumulated cpu and wall time of a code segment bracketed by two calls of time()
  type :: timer
     real(dp) :: cpu_time = 0
     integer(int64) :: wall_time = 0
     integer :: num_calls = 0
     logical, private :: running = .false.
     real(dp), private :: cpu_mark = 0
     integer(int64), private :: wall_mark = 0
   contains
     procedure :: reset => timer_reset
     procedure :: time => timer_toggle
  end type timer

  type :: data_pointer
     real(kp), pointer :: x_data(:) => null(), y_data(:) => null(), weights(:) => null()
  end type data_pointer

  interface len
     module procedure string_length
  end interface len

  interface safe_deallocate
     module procedure free_dp, free_qp, free_dp_2d, free_qp_2d, free_integer, free_string, free_logical, free_data_pointer
  end interface safe_deallocate

contains

  impure elemental subroutine string_from_character(this, x)
    class(string), intent(out) :: this
    character(*), intent(in) :: x
    this%name = x
  end subroutine string_from_character

  logical function string_is_character(this, x) result(same)
    class(string), intent(in) :: this
    character(*), intent(in) :: x
    same = allocated(this%name)
    if (same) same = this%name == x
  end function string_is_character

  elemental integer function string_length(x) result(n)
    type(string), intent(in) :: x
    n = 0
    if (allocated(x%name)) n = len(x%name)
  end function string_length

  elemental subroutine swap(a, b)
    integer, intent(in out) :: a, b
    integer :: keep
    keep = a
    a = b
    b = keep
  end subroutine swap

  subroutine timer_reset(this)
    class(timer), intent(out) :: this
    this%cpu_time = 0; this%wall_time = 0; this%num_calls = 0; this%running = .false.
  end subroutine timer_reset

  subroutine timer_toggle(this)
    class(timer), intent(in out) :: this
    real(dp) :: cpu_now
    integer(int64) :: wall_now
    call cpu_time(cpu_now)
    call system_clock(wall_now)
    if (this%running) then
       this%cpu_time = this%cpu_time + (cpu_now - this%cpu_mark)
       this%wall_time = this%wall_time + (wall_now - this%wall_mark)
       this%num_calls = this%num_calls + 1
    else
       this%cpu_mark = cpu_now; this%wall_mark = wall_now
    end if
    this%running = .not. this%running
  end subroutine timer_toggle

  ! One body for every type and rank: deallocate if allocated, report through check_err.
#define FREE_BODY \
    character(*), intent(in) :: file; \
    integer, intent(in) :: line; \
    if (.not. allocated(array)) return; \
    deallocate(array, stat=err_stat, errmsg=err_msg); \
    call check_err(file, line)

  subroutine free_dp(file, line, array)
    real(dp), allocatable, intent(in out) :: array(:)
    FREE_BODY
  end subroutine free_dp
  subroutine free_qp(file, line, array)
    real(qp), allocatable, intent(in out) :: array(:)
    FREE_BODY
  end subroutine free_qp
  subroutine free_dp_2d(file, line, array)
    real(dp), allocatable, intent(in out) :: array(:,:)
    FREE_BODY
  end subroutine free_dp_2d
  subroutine free_qp_2d(file, line, array)
    real(qp), allocatable, intent(in out) :: array(:,:)
    FREE_BODY
  end subroutine free_qp_2d
  subroutine free_integer(file, line, array)
    integer, allocatable, intent(in out) :: array(:)
    FREE_BODY
  end subroutine free_integer
  subroutine free_string(file, line, array)
    type(string), allocatable, intent(in out) :: array(:)
    FREE_BODY
  end subroutine free_string
  subroutine free_logical(file, line, array)
    logical, allocatable, intent(in out) :: array(:)
    FREE_BODY
  end subroutine free_logical
  subroutine free_data_pointer(file, line, array)
    type(data_pointer), allocatable, intent(in out) :: array(:)
    FREE_BODY
  end subroutine free_data_pointer

  subroutine safe_close(file, line, io_unit)
    character(*), intent(in) :: file
    integer, intent(in) :: line, io_unit
    close(io_unit, iostat=err_stat, iomsg=err_msg)
    call check_err(file, line)
  end subroutine safe_close
end module misc
