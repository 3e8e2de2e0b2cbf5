! Error convention of the reference (fortran/gadfit/messaging.f90:19-21, 32-62, 103-141): error / warning / comment print
! where-and-what and (error) stop; check_err turns the stat= / errmsg= pair err_stat / err_msg of an allocate, deallocate
! or close into an error; str and print_memory are the small formatting helpers its callers use.  Errors never return codes.
module messaging
  use, intrinsic :: iso_fortran_env, only: error_unit, output_unit
  implicit none
  private
  public :: error, warning, comment, check_err, err_stat, err_msg, str, print_memory
  integer :: err_stat = 0
  character(200) :: err_msg = ''
contains
  subroutine say(unit, kind, file, line, msg)
    integer, intent(in) :: unit, line
    character(*), intent(in) :: kind, file, msg
    write(unit, '(a, a, a, ":", i0)') kind, ' at ', file, line
    write(unit, '(2x, a)') msg
    flush(unit)
  end subroutine say

  subroutine error(file, line, msg)
    character(*), intent(in) :: file, msg
    integer, intent(in) :: line
    call say(error_unit, 'Error', file, line, msg)
    error stop
  end subroutine error

  subroutine warning(file, line, msg)
    character(*), intent(in) :: file, msg
    integer, intent(in) :: line
    call say(error_unit, 'Warning', file, line, msg)
  end subroutine warning

  subroutine comment(file, line, msg)
    character(*), intent(in) :: file, msg
    integer, intent(in) :: line
    call say(output_unit, 'Comment', file, line, msg)
  end subroutine comment

  ! after `allocate(..., stat=err_stat, errmsg=err_msg)` and the like
  subroutine check_err(file, line)
    character(*), intent(in) :: file
    integer, intent(in) :: line
    if (err_stat /= 0) call error(file, line, trim(err_msg))
  end subroutine check_err

  pure function str(x) result(y)
    integer, intent(in) :: x
    character(:), allocatable :: y
    character(24) :: digits
    write(digits, '(i0)') x
    y = trim(digits)
  end function str

  ! x bytes as "<n> B", "<n.n> kB", ... without a line end
  subroutine print_memory(io_unit, x)
    integer, intent(in) :: io_unit, x
    character(2), parameter :: unit(3) = ['kB', 'MB', 'GB']
    real :: v
    integer :: k
    if (x < 1000) then
       write(io_unit, '(i0, a)', advance='no') x, ' B'
       return
    end if
    v = real(x)/1e3; k = 1
    do while (v >= 1e3 .and. k < 3)
       v = v/1e3; k = k + 1
    end do
    write(io_unit, '(f0.1, 1x, a)', advance='no') v, unit(k)
  end subroutine print_memory
end module messaging
