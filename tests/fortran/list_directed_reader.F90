! The reference's way of reading a data file, verbatim in spirit (gadfit.F90:212-215 and 422-437): one list-directed read per record
! to count the records that begin with a number, then one list-directed read per record for their first two or three numbers.  No
! GPU, no library: the tests compare what libgadfit_hip's reader (reader.cpp) returns for the same file with what Fortran's own
! list-directed input makes of it.  usage: list_directed_reader <file> <2|3>; prints n and the values with 17 digits.
program list_directed_reader
  use, intrinsic :: iso_fortran_env, only: real64
  implicit none
  character(len=512) :: path, arg
  integer :: u, stat, n, j, ncol
  real(real64) :: a, b, c
  real(real64), allocatable :: x(:), y(:), w(:)
  call get_command_argument(1, path); call get_command_argument(2, arg); read(arg, *) ncol
  n = 0
  open(newunit=u, file=trim(path), status='old', action='read')
  do
     read(u, *, iostat=stat) a
     if (stat < 0) exit
     if (stat == 0) n = n + 1
  end do
  close(u)
  allocate(x(n), y(n), w(n)); w = 1.0_real64
  j = 0
  open(newunit=u, file=trim(path), status='old', action='read')
  do while (j < n)
     if (ncol == 3) then
        read(u, *, iostat=stat) a, b, c
     else
        read(u, *, iostat=stat) a, b
        c = 1.0_real64
     end if
     if (stat < 0) exit
     if (stat == 0) then
        j = j + 1
        x(j) = a; y(j) = b; w(j) = c
     end if
  end do
  close(u)
  print '(i0)', n
  do j = 1, n
     print '(3es25.17)', x(j), y(j), w(j)
  end do
end program list_directed_reader
