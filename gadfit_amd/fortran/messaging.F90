! Error convention of the reference (fortran/gadfit/messaging.f90:32-41): print
! file:line + message to stderr and `error stop`.  Errors never return codes.
module messaging
  use, intrinsic :: iso_fortran_env, only: error_unit
  implicit none
  public
contains
  subroutine error(file, line, msg)
    character(*), intent(in) :: file, msg
    integer, intent(in) :: line
    write(error_unit, '(a, a, ":", i0)') 'Error at ', file, line
    write(error_unit, '(2x, a)') msg
    flush(error_unit)
    error stop
  end subroutine error

  subroutine warning(file, line, msg)
    character(*), intent(in) :: file, msg
    integer, intent(in) :: line
    write(error_unit, '(a, a, ":", i0)') 'Warning at ', file, line
    write(error_unit, '(2x, a)') msg
  end subroutine warning
end module messaging
