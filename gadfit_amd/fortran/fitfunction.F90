! The abstract fitting function (mirror of fortran/gadfit/fitfunction.F90:32-64): a user
! model extends fitfunc with init (allocates pars, optional names) and eval(this, x).
module fitfunction
  use ad
  use gadf_constants, only: kp
  use messaging
  implicit none
  private
  public :: fitfunc

  type, abstract :: fitfunc
     type(advar), allocatable :: pars(:)
     character(len=32), allocatable :: parnames(:)
   contains
     procedure(init), deferred :: init
     procedure(eval), deferred :: eval
     procedure :: set_value_int, set_value_char, set_name
     generic :: set => set_value_int, set_value_char, set_name
     procedure :: get_index
     procedure :: get_name
  end type fitfunc

  abstract interface
     subroutine init(this)
       import fitfunc
       class(fitfunc), intent(out) :: this
     end subroutine init
     type(advar) function eval(this, x) result(y)
       import fitfunc, advar, kp
       class(fitfunc), intent(in) :: this
       real(kp), intent(in) :: x
     end function eval
  end interface

contains

  ! fitfunction.F90:66-109
  subroutine set_value_int(this, par, val)
    class(fitfunc), intent(in out) :: this
    integer, intent(in) :: par
    real(kp), intent(in) :: val
    if (.not. allocated(this%pars)) call error(__FILE__, __LINE__, 'Parameter array is not allocated.')
    this%pars(par)%val = val
  end subroutine set_value_int

  subroutine set_value_char(this, par, val)
    class(fitfunc), intent(in out) :: this
    character(*), intent(in) :: par
    real(kp), intent(in) :: val
    call this%set_value_int(this%get_index(par), val)
  end subroutine set_value_char

  subroutine set_name(this, par, name)
    class(fitfunc), intent(in out) :: this
    integer, intent(in) :: par
    character(*), intent(in) :: name
    if (.not. allocated(this%pars)) call error(__FILE__, __LINE__, 'Parameter array is not allocated.')
    if (.not. allocated(this%parnames)) then
       allocate(this%parnames(size(this%pars)))
       this%parnames = ''
    end if
    this%parnames(par) = name
  end subroutine set_name

  ! fitfunction.F90:111-127
  integer function get_index(this, name) result(y)
    class(fitfunc), intent(in) :: this
    character(*), intent(in) :: name
    if (allocated(this%parnames)) then
       do y = 1, size(this%parnames)
          if (trim(this%parnames(y)) == name) return
       end do
    end if
    y = 0
    call error(__FILE__, __LINE__, 'Parameter with name '''//name//''' not found.')
  end function get_index

  function get_name(this, index) result(y)
    class(fitfunc), intent(in) :: this
    integer, intent(in) :: index
    character(:), allocatable :: y
    if (allocated(this%parnames)) then
       y = trim(this%parnames(index))
    else
       y = ''
    end if
  end function get_name
end module fitfunction
