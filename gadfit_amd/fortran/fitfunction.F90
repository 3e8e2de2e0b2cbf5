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
     procedure :: grad_finite            ! gradient by finite differences (host)
     procedure :: dir_deriv_2nd_finite   ! second directional derivative by finite differences (host)
     procedure :: info
     procedure :: destroy
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

  ! fitfunction.F90:155-174: forward differences with the step sqrt(eps)*value, for the parameters listed in
  ! active_pars; grad(:n).  Host-side debugging aid (the reference's use_ad=.false. path is built on it).
  subroutine grad_finite(this, x, active_pars, grad)
    class(fitfunc), intent(in out) :: this
    real(kp), intent(in) :: x
    integer, intent(in) :: active_pars(:)
    real(kp), intent(in out) :: grad(:)
    real(kp) :: saved_value, step
    type(advar) :: y
    character(16) :: num
    integer :: i
    do i = 1, size(active_pars)
       saved_value = this%pars(active_pars(i))%val
       step = sqrt(epsilon(1.0_kp))*saved_value
       if (.not. abs(step) > tiny(0.0_kp)) then
          write(num, '(i0)') active_pars(i)
          call error(__FILE__, __LINE__, 'Absolute value of parameter '//trim(num)//' is too small.')
       end if
       this%pars(active_pars(i))%val = this%pars(active_pars(i))%val + step
       step = this%pars(active_pars(i))%val - saved_value
       y = this%eval(x)
       grad(i) = y%val
       this%pars(active_pars(i))%val = saved_value
       y = this%eval(x)
       grad(i) = (grad(i) - y%val)/step
    end do
  end subroutine grad_finite

  ! fitfunction.F90:188-203: central second difference along dir with h = eps**(1/4)
  real(kp) function dir_deriv_2nd_finite(this, x, active_pars, dir) result(y)
    class(fitfunc), intent(in out) :: this
    real(kp), intent(in) :: x, dir(:)
    integer, intent(in) :: active_pars(:)
    real(kp) :: saved_values(size(active_pars)), h
    type(advar) :: f
    saved_values = this%pars(active_pars)%val
    h = sqrt(sqrt(epsilon(1.0_kp)))
    this%pars(active_pars)%val = this%pars(active_pars)%val + h*dir
    f = this%eval(x); y = f%val
    this%pars(active_pars)%val = saved_values - h*dir
    f = this%eval(x); y = y + f%val
    this%pars(active_pars)%val = saved_values
    f = this%eval(x); y = y - 2*f%val
    y = y/sqrt(epsilon(1.0_kp))
  end function dir_deriv_2nd_finite

  ! fitfunction.F90:207-225: names, values and activity (index /= 0) of the parameters
  subroutine info(this)
    use, intrinsic :: iso_fortran_env, only: output_unit
    class(fitfunc), intent(in out) :: this
    integer :: i
    if (.not. allocated(this%pars)) call this%init()
    do i = 1, size(this%pars)
       if (this%pars(i)%index == 0) then
          write(output_unit, '(a)', advance='no') 'Passive'
       else
          write(output_unit, '(1x, a)', advance='no') 'Active'
       end if
       write(output_unit, '(2x, a, 2x, g0)') this%get_name(i), this%pars(i)%val
    end do
  end subroutine info

  ! fitfunction.F90:227-231
  impure elemental subroutine destroy(this)
    class(fitfunc), intent(in out) :: this
    if (allocated(this%pars)) deallocate(this%pars)
    if (allocated(this%parnames)) deallocate(this%parnames)
  end subroutine destroy
end module fitfunction
