! The abstract fitting function (mirror of fortran/gadfit/fitfunction.F90:30-64): a user
! model extends fitfunc with init (allocates pars, optional names) and eval(this, x).  parnames
! is the reference's array of type(string) (misc.F90:28-35): user code may read parnames(i)%name,
! len(parnames), compare a name with a character, and names have any length.
module fitfunction
  use ad
  use gadf_constants, only: kp
  use messaging
  use misc, only: string, len, safe_deallocate
  implicit none
  private
  public :: fitfunc, safe_deallocate

  type, abstract :: fitfunc
     type(advar), allocatable :: pars(:)
     type(string), allocatable :: parnames(:)
   contains
     procedure(init), deferred :: init
     procedure(eval), deferred :: eval
     procedure, private :: set_value_int, set_value_char, set_name, blank_names
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

  ! the fitfunc specific of misc's generic (fitfunction.F90:66-70, 232-243)
  interface safe_deallocate
     module procedure safe_deallocate_fitfunc
  end interface safe_deallocate

contains

  ! every parameter gets an (empty) name as soon as anything is set: the print procedures rely on it (fitfunction.F90:74-79, 93-95)
  subroutine blank_names(this)
    class(fitfunc), intent(in out) :: this
    integer :: i
    if (allocated(this%parnames)) return
    allocate(this%parnames(size(this%pars)))
    do i = 1, size(this%parnames)
       this%parnames(i)%name = ''
    end do
  end subroutine blank_names

  ! fitfunction.F90:82-104
  subroutine set_value_int(this, par, val)
    class(fitfunc), intent(in out) :: this
    integer, intent(in) :: par
    real(kp), intent(in) :: val
    if (.not. allocated(this%pars)) call error(__FILE__, __LINE__, 'Parameter array is not allocated (init() allocates it).')
    if (par < 1 .or. par > size(this%pars)) call error(__FILE__, __LINE__, 'Index out of bounds.')
    this%pars(par)%val = val
    call this%blank_names()
  end subroutine set_value_int

  subroutine set_value_char(this, par, val)
    class(fitfunc), intent(in out) :: this
    character(*), intent(in) :: par
    real(kp), intent(in) :: val
    call this%set_value_int(this%get_index(par), val)
  end subroutine set_value_char

  ! fitfunction.F90:106-118 (a name given twice draws the reference's warning)
  subroutine set_name(this, par, name)
    class(fitfunc), intent(in out) :: this
    integer, intent(in) :: par
    character(*), intent(in) :: name
    if (.not. allocated(this%pars)) call error(__FILE__, __LINE__, 'Parameter array is not allocated (init() allocates it).')
    call this%blank_names()
    if (any_named(this%parnames, name)) call warning(__FILE__, __LINE__, 'The name "'//name//'" is already in use.')
    this%parnames(par) = name
  end subroutine set_name

  logical function any_named(names, name) result(y)
    type(string), intent(in) :: names(:)
    character(*), intent(in) :: name
    integer :: i
    y = .false.
    do i = 1, size(names)
       if (names(i) == name) y = .true.
    end do
  end function any_named

  ! fitfunction.F90:120-137: the index of the parameter called `name`; an unknown name is an error that lists the known ones
  integer function get_index(this, name) result(y)
    class(fitfunc), intent(in out) :: this
    character(*), intent(in) :: name
    character(:), allocatable :: known
    call this%blank_names()
    known = ''
    do y = 1, size(this%parnames)
       if (this%parnames(y) == name) return
       known = known//merge('  ', ', ', y == 1)//this%parnames(y)%name
    end do
    y = 0
    call error(__FILE__, __LINE__, 'There is no parameter called "'//name//'". Allowed names are'//known//'.')
  end function get_index

  ! fitfunction.F90:139-143
  elemental type(string) function get_name(this, par_i) result(y)
    class(fitfunc), intent(in) :: this
    integer, intent(in) :: par_i
    y%name = ''
    if (allocated(this%parnames)) then
       if (allocated(this%parnames(par_i)%name)) y%name = this%parnames(par_i)%name
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
       call this%blank_names()
       write(output_unit, '(2x, a, 2x, g0)') this%parnames(i)%name, this%pars(i)%val
    end do
  end subroutine info

  ! fitfunction.F90:227-231
  impure elemental subroutine destroy(this)
    class(fitfunc), intent(in out) :: this
    call safe_deallocate(__FILE__, __LINE__, this%pars)
    call safe_deallocate(__FILE__, __LINE__, this%parnames)
  end subroutine destroy

  ! destroys the elements, then the array (fitfunction.F90:232-243)
  subroutine safe_deallocate_fitfunc(file, line, array)
    character(*), intent(in) :: file
    integer, intent(in) :: line
    class(fitfunc), allocatable, intent(in out) :: array(:)
    if (.not. allocated(array)) return
    call array%destroy()
    deallocate(array, stat=err_stat, errmsg=err_msg)
    call check_err(file, line)
  end subroutine safe_deallocate_fitfunc
end module fitfunction
