! What one recording of eval() costs in the recorder's thread-checking mode (module ad, ad_fast_check; ad_tls.c), single thread, headline model:
! amdflang -O2 -cpp -fopenmp -I gadfit_amd/fortran/build tools/probes/check_cost.F90 gadfit_amd/fortran/build/libgadfit_f.a -Lgadfit_amd/lib -lgadfit_hip -Wl,-rpath,$PWD/gadfit_amd/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib -o /tmp/check_cost
module gauss8_model
  use ad
  use gadfit_hip_c
  use fitfunction
  use gadf_constants
  implicit none
  type, extends(fitfunc) :: gauss8_t
   contains
     procedure :: init => g8_init
     procedure :: eval => g8_eval
  end type gauss8_t
contains
  subroutine g8_init(this)
    class(gauss8_t), intent(out) :: this
    allocate(this%pars(32))
  end subroutine g8_init
  type(advar) function g8_eval(this, x) result(y)
    class(gauss8_t), intent(in) :: this
    real(kp), intent(in) :: x
    integer :: k
    y = 0.0_kp
    do k = 0, 7
       y = y + this%pars(4*k+1)*exp(-(((x - this%pars(4*k+2))/this%pars(4*k+3))**2))*(1.0_kp + this%pars(4*k+4)*(x - this%pars(4*k+2)))
    end do
  end function g8_eval
end module gauss8_model
program t
  use gauss8_model
  use ad
  use gadfit_hip_c
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: int64
  implicit none
  type(gauss8_t) :: f
  type(advar) :: y
  integer :: i, k, n, rep, nn, res
  integer(int64) :: c0, c1, rate
  real(kp) :: x, best
  integer(c_int), allocatable, target :: op(:), a(:), b(:), fl(:), cls(:)
  real(c_double), allocatable, target :: c(:), al(:), be(:)
  integer(c_int) :: cn, cdiv, clit
  call f%init()
  do k = 0, 7
     f%pars(4*k+1) = 1.0_kp + 4.0_kp*k/7.0_kp; f%pars(4*k+2) = 6.0_kp + 12.0_kp*k
     f%pars(4*k+3) = 2.0_kp + 2.0_kp*k/7.0_kp; f%pars(4*k+4) = 0.01_kp*(1 + mod(k, 3))
  end do
  ! one full recording at x = 1 and one at x = 2: literals that differ are affine in x (the model's only literal kind besides constants)
  call ad_capture_begin(); call ad_emit_params(32)
  do k = 1, 32; f%pars(k)%node = k - 1; end do
  y = f%eval(1.0_kp); res = anode(y); call ad_capture_end()
  nn = ad_tape_n
  allocate(op(nn), a(nn), b(nn), fl(nn), cls(nn), c(nn), al(nn), be(nn))
  op = ad_tape(:nn)%op; a = ad_tape(:nn)%a; b = ad_tape(:nn)%b; fl = ad_tape(:nn)%flags; c = ad_tape(:nn)%c
  call ad_capture_begin(); call ad_emit_params(32)
  y = f%eval(2.0_kp); call ad_capture_end()
  cls = 0; al = 0; be = 0
  do k = 1, nn
     if (op(k) /= GFH_CONST) cycle
     if (ad_tape(k)%c == c(k)) then; cls(k) = 1
     else; cls(k) = 2; al(k) = ad_tape(k)%c - c(k); be(k) = c(k) - al(k); end if
  end do
  print *, 'nodes', nn, ' literals', count(op == GFH_CONST), ' affine', count(cls == 2)
  call gfh_adchk_load(int(nn, c_int), op, a, b, fl, cls, c, al, be)
  n = 200000
  ad_recording = .true.; ad_thread_check = .true.; ad_need_vals = .false.; ad_cur = 0; ad_fast_check = .true.
  best = huge(best)
  do rep = 1, 7
     call system_clock(c0, rate)
     do i = 1, n
        x = 100.0_kp*(real(i, kp) - 0.5_kp)/real(n, kp)
        call gfh_adchk_begin(x, 32_c_int)
        y = f%eval(x)
        res = anode(y)
        call gfh_adchk_end(cn, cdiv, clit)
        if (cdiv /= 0 .or. clit /= 0 .or. cn /= nn) stop 'check failed'
     end do
     call system_clock(c1)
     best = min(best, 1e9*real(c1 - c0)/real(rate)/n)
  end do
  print '(a, f8.1, a)', 'thread-check mode: ', best, ' ns per eval() (best of 7)'
end program t
