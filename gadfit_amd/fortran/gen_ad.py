#!/usr/bin/env python3
"""Generates ad.F90: ``module ad`` -- type(advar) with the reference's operator set
(fortran/gadfit/automatic_differentiation.F90:82-229) for real32 / real(dp) / real(qp) /
integer operands, written as a RECORDER: every elemental computes the value (same expression
shapes as the reference, e.g. a/r multiplies by the reciprocal, AD:843-866) and, while a
model is being captured, appends one node to the model tape (include/gadfit_tape.h).

The ~95 specific procedures differ only in operand types, so they are generated.
Run:  python gen_ad.py > ad.F90
"""

BIN = [  # name, operator, GFH op, value for (a,a), (a,r), (r,a)
    ('add', '+', 'GFH_ADD', 'x1%val + x2%val', 'x1%val + r2', 'r1 + x2%val'),
    ('subtract', '-', 'GFH_SUB', 'x1%val - x2%val', 'x1%val - r2', 'r1 - x2%val'),
    ('multiply', '*', 'GFH_MUL', 'x1%val*x2%val', 'x1%val*r2', 'r1*x2%val'),
    ('divide', '/', 'GFH_DIV', 'x1%val/x2%val', 'x1%val*(1/r2)', 'r1/x2%val'),
    ('power', '**', 'GFH_POW', 'x1%val**x2%val', 'x1%val**r2', 'r1**x2%val'),
]
# Forward mode (AD:454-1459, the "not reverse_mode" branches): an operand is active iff index /= 0; an active result
# carries the first and second directional derivatives (d, dd) and index = 1.  Y = y%val, V1/V2 = operand values.
FWD_AA = {
    'add': 'y%d = x1%d + x2%d; y%dd = x1%dd + x2%dd',
    'subtract': 'y%d = x1%d - x2%d; y%dd = x1%dd - x2%dd',
    'multiply': 'y%d = x1%val*x2%d + x1%d*x2%val; y%dd = x1%val*x2%dd + 2*x1%d*x2%d + x1%dd*x2%val',
    'divide': 't = 1/x2%val; y%d = (x1%d - y%val*x2%d)*t; y%dd = (x1%dd - y%val*x2%dd - 2*y%d*x2%d)*t',
    'power': 'y%d = y%val*x2%d*log(x1%val) + x1%d*x2%val*x1%val**(x2%val - 1); t = 1/x1%val; '
             'y%dd = y%d**2/y%val + y%val*(x2%dd*log(x1%val) + (2*x2%d*x1%d + x2%val*(x1%dd - x1%d**2*t))*t)',
}
FWD_AR = {   # advar o real (r2)
    'add': 'y%d = x1%d; y%dd = x1%dd',
    'subtract': 'y%d = x1%d; y%dd = x1%dd',
    'multiply': 'y%d = x1%d*r2; y%dd = x1%dd*r2',
    'divide': 't = 1/r2; y%d = x1%d*t; y%dd = x1%dd*t',
    'power': 'y%d = x1%d*r2*x1%val**(r2 - 1); t = 1/x1%val; y%dd = y%d**2/y%val + y%val*r2*(x1%dd - x1%d**2*t)*t',
}
FWD_RA = {   # real (r1) o advar
    'add': 'y%d = x2%d; y%dd = x2%dd',
    'subtract': 'y%d = -x2%d; y%dd = -x2%dd',
    'multiply': 'y%d = x2%d*r1; y%dd = x2%dd*r1',
    'divide': 'y%d = -y%val*x2%d/x2%val; y%dd = (-y%val*x2%dd - 2*y%d*x2%d)/x2%val',
    'power': 't = log(r1); y%d = y%val*x2%d*t; y%dd = y%d**2/y%val + y%val*x2%dd*t',
}
FWD_UN = {
    'abs': 't = sign(1.0_kp, x%val); y%d = x%d*t; y%dd = x%dd*t',
    'exp': 'y%d = x%d*y%val; y%dd = x%dd*y%val + x%d*y%d',
    'sqrt': 't = 1/y%val; y%d = x%d/2*t; y%dd = (x%dd*t - y%d*x%d/x%val)/2',
    'log': 't = 1/x%val; y%d = x%d*t; y%dd = (x%dd - x%d*y%d)*t',
    'sin': 't = cos(x%val); y%d = x%d*t; y%dd = x%dd*t - x%d**2*y%val',
    'cos': 't = -sin(x%val); y%d = x%d*t; y%dd = x%dd*t - x%d**2*y%val',
    'tan': 't = (1/cos(x%val))**2; y%d = x%d*t; y%dd = x%dd*t + 2*x%d*y%val*y%d',
    'asin': 't = 1/sqrt(1 - x%val**2); y%d = x%d*t; y%dd = t*(x%dd + x%val*y%d**2)',
    'acos': 't = -1/sqrt(1 - x%val**2); y%d = x%d*t; y%dd = t*(x%dd + x%val*y%d**2)',
    'atan': 't = 1/(1 + x%val**2); y%d = x%d*t; y%dd = x%dd*t - 2*x%val*y%d**2',
    'sinh': 't = cosh(x%val); y%d = x%d*t; y%dd = x%dd*t + y%val*x%d**2',
    'cosh': 't = sinh(x%val); y%d = x%d*t; y%dd = x%dd*t + y%val*x%d**2',
    'tanh': 't = 1/cosh(x%val)**2; y%d = x%d*t; y%dd = x%dd*t - 2*y%val*x%d*y%d',
    'asinh': 't = 1/sqrt(1 + x%val**2); y%d = x%d*t; y%dd = t*(x%dd - x%val*y%d**2)',
    'acosh': 't = 1/sqrt(x%val**2 - 1); y%d = x%d*t; y%dd = t*(x%dd - x%val*y%d**2)',
    'atanh': 't = 1/(1 - x%val**2); y%d = x%d*t; y%dd = (x%dd + 2*x%val*x%d*y%d)*t',
    'erf': 't = 1.1283791670955125738961589031215452_kp*exp(-x%val**2); y%d = x%d*t; y%dd = (x%dd - 2*x%d**2*x%val)*t',
}


# Reverse mode on the host (AD:454-1459 "reverse_mode" branches, ad_grad AD:1476-1659): an active result takes the next slot
# of forward_values and one fixed-width record [code, operand slot 1, operand slot 2 | integer payload, result slot] in
# `trace`; a real operand (or passive advar) goes to ad_constants.  code = GFH op + 100 for (advar o real), + 200 for
# (real o advar).  ad_grad walks the records backwards.  X1/X2 = operand adjoint slots, Y = result slot, C = the
# constant, V(.) = forward value, A = adjoint of the result.
REV_AA = {   # both operands active
    'add': 'adjoints(i1) = adjoints(i1) + a; adjoints(i2) = adjoints(i2) + a',
    'subtract': 'adjoints(i1) = adjoints(i1) + a; adjoints(i2) = adjoints(i2) - a',
    'multiply': 'adjoints(i1) = adjoints(i1) + a*forward_values(i2); adjoints(i2) = adjoints(i2) + a*forward_values(i1)',
    'divide': 'adjoints(i1) = adjoints(i1) + a/forward_values(i2); adjoints(i2) = adjoints(i2) - a*forward_values(iy)/forward_values(i2)',
    'power': 'adjoints(i1) = adjoints(i1) + a*forward_values(i2)*forward_values(i1)**(forward_values(i2) - 1); '
             'adjoints(i2) = adjoints(i2) + a*log(forward_values(i1))*forward_values(i1)**forward_values(i2)',
}
REV_AR = {   # advar o real: c = the real operand (for divide its reciprocal: a/r is a*(1/r), AD:843-866)
    'add': 'adjoints(i1) = adjoints(i1) + a',
    'subtract': 'adjoints(i1) = adjoints(i1) + a',
    'multiply': 'adjoints(i1) = adjoints(i1) + a*c',
    'divide': 'adjoints(i1) = adjoints(i1) + a*c',
    'power': 'adjoints(i1) = adjoints(i1) + a*c*forward_values(i1)**(c - 1)',
}
REV_RA = {   # real o advar: c = the real operand
    'add': 'adjoints(i1) = adjoints(i1) + a',
    'subtract': 'adjoints(i1) = adjoints(i1) - a',
    'multiply': 'adjoints(i1) = adjoints(i1) + a*c',
    'divide': 'adjoints(i1) = adjoints(i1) - a*c/forward_values(i1)/forward_values(i1)',
    'power': 'adjoints(i1) = adjoints(i1) + a*log(c)*c**forward_values(i1)',
}
REV_UN = {
    'abs': 'if (forward_values(i1) < 0) then; adjoints(i1) = adjoints(i1) - a; else; adjoints(i1) = adjoints(i1) + a; end if',
    'exp': 'adjoints(i1) = adjoints(i1) + a*forward_values(iy)',
    'sqrt': 'adjoints(i1) = adjoints(i1) + a/2/forward_values(iy)',
    'log': 'adjoints(i1) = adjoints(i1) + a/forward_values(i1)',
    'sin': 'adjoints(i1) = adjoints(i1) + a*cos(forward_values(i1))',
    'cos': 'adjoints(i1) = adjoints(i1) - a*sin(forward_values(i1))',
    'tan': 'adjoints(i1) = adjoints(i1) + a/cos(forward_values(i1))**2',
    'asin': 'adjoints(i1) = adjoints(i1) + a/sqrt(1 - forward_values(i1)**2)',
    'acos': 'adjoints(i1) = adjoints(i1) - a/sqrt(1 - forward_values(i1)**2)',
    'atan': 'adjoints(i1) = adjoints(i1) + a/(1 + forward_values(i1)**2)',
    'sinh': 'adjoints(i1) = adjoints(i1) + a*cosh(forward_values(i1))',
    'cosh': 'adjoints(i1) = adjoints(i1) + a*sinh(forward_values(i1))',
    'tanh': 'adjoints(i1) = adjoints(i1) + a/cosh(forward_values(i1))**2',
    'asinh': 'adjoints(i1) = adjoints(i1) + a/sqrt(forward_values(i1)**2 + 1)',
    'acosh': 'adjoints(i1) = adjoints(i1) + a/sqrt(forward_values(i1)**2 - 1)',
    'atanh': 'adjoints(i1) = adjoints(i1) + a/(1 - forward_values(i1)**2)',
    'erf': 'adjoints(i1) = adjoints(i1) + a*1.1283791670955125738961589031215452_kp*exp(-forward_values(i1)**2)',
}
GOP = {'add': 'GFH_ADD', 'subtract': 'GFH_SUB', 'multiply': 'GFH_MUL', 'divide': 'GFH_DIV', 'power': 'GFH_POW'}


def fwd(stmts, cond, rec=None):
    """Fortran lines for an active result: the reverse-mode record `rec` (a call) when reverse_mode is set, else the
    forward-mode statements (separated by '; ')"""
    body = '\n'.join('          ' + st.strip() for st in stmts.split(';'))
    return ('    if (%s) then\n       if (reverse_mode) then\n          %s\n       else\n%s\n          y%%index = 1\n'
            '       end if\n    end if' % (cond, rec, body))


RTYPES = [('real32', 'real(real32)'), ('dp', 'real(dp)'), ('qp', 'real(qp)'), ('integer', 'integer')]
UNARY = ['abs', 'exp', 'sqrt', 'log', 'sin', 'cos', 'tan', 'asin', 'acos', 'atan', 'sinh', 'cosh', 'tanh',
         'asinh', 'acosh', 'atanh', 'erf']

PLUMB = r'''
  ! ---------------------------------------------------------------- host-side reverse mode
  ! ad_init_reverse (AD:251-313): work arrays from a memory string ('<number> B|kB|MB|GB': forward_values(x),
  ! adjoints(x), trace(4x), ad_constants(x/2) with x = memory/(2.5*kp + 4*kind(1))) or from explicit sizes
  ! (defaults 10000, 4*10000, 10000/2); switches the default mode to reverse.
  subroutine ad_init_reverse(memory, sweep_size, trace_size, const_size)
    character(*), intent(in), optional :: memory
    integer, intent(in), optional :: sweep_size, trace_size, const_size
    integer :: ns, nt, nc, ios
    real(kp) :: amount, scale
    character(2) :: unit
    if (present(memory)) then
       read(memory, *, iostat=ios) amount, unit
       if (ios /= 0) error stop 'ad_init_reverse: cannot read the memory specification'
       select case (unit)
       case ('B', 'b'); scale = 1.0_kp
       case ('kB', 'kb'); scale = 1e3
       case ('MB', 'mb'); scale = 1e6
       case ('GB', 'gb'); scale = 1e9
       case default; error stop 'ad_init_reverse: unrecognized unit'
       end select
       ns = int(real(amount, kp)/(2.5*kp + 4*kind(1))*scale)
       nt = 4*ns
       nc = ns/2
    else
       ns = DEFAULT_SWEEP_SIZE; nt = 4*DEFAULT_SWEEP_SIZE; nc = DEFAULT_SWEEP_SIZE/2
       if (present(sweep_size)) ns = sweep_size
       if (present(trace_size)) nt = trace_size
       if (present(const_size)) nc = const_size
    end if
    call ad_close()
    allocate(forward_values(ns), adjoints(ns), trace(nt), ad_constants(nc))
    adjoints = 0.0_kp; trace = 0
    trace_count = 0; index_count = 0; const_count = 0
    max_trace_count = 0; max_index_count = 0; max_const_count = 0
    reverse_mode = .true.
  end subroutine ad_init_reverse

  subroutine safe_deallocate_advar(file, line, array)
    character(*), intent(in) :: file
    integer, intent(in) :: line
    type(advar), allocatable, intent(in out) :: array(:)
    if (.not. allocated(array)) return
    deallocate(array, stat=err_stat, errmsg=err_msg)
    call check_err(file, line)
  end subroutine safe_deallocate_advar

  ! frees the work arrays (AD: ad_close)
  subroutine ad_close()
    if (allocated(forward_values)) deallocate(forward_values)
    if (allocated(adjoints)) deallocate(adjoints)
    if (allocated(trace)) deallocate(trace)
    if (allocated(ad_constants)) deallocate(ad_constants)
  end subroutine ad_close

  ! one active result: next slot of forward_values, one record in trace, optionally one constant
  subroutine ad_push(code, i1, i2, y, c)
    integer, intent(in) :: code, i1, i2
    type(advar), intent(in out) :: y
    real(kp), intent(in), optional :: c
    if (.not. allocated(trace)) error stop 'reverse mode of module ad is not initialized (ad_init_reverse)'
    if (index_count + 1 > size(forward_values)) error stop 'module ad: forward_values is full (sweep_size)'
    if (trace_count + 4 > size(trace)) error stop 'module ad: trace is full (trace_size)'
    index_count = index_count + 1
    y%index = index_count
    forward_values(index_count) = y%val
    trace(trace_count+1) = code; trace(trace_count+2) = i1; trace(trace_count+3) = i2; trace(trace_count+4) = index_count
    trace_count = trace_count + 4
    if (present(c)) then
       if (const_count + 1 > size(ad_constants)) error stop 'module ad: ad_constants is full (const_size)'
       const_count = const_count + 1
       ad_constants(const_count) = c
    end if
  end subroutine ad_push

  ! ad_grad (AD:1476-1659): the return sweep.  The last value written is the function result (seed 1); on return
  ! adjoints(1:num_pars) hold the gradient with respect to the parameters in slots 1..num_pars and the counters are
  ! reset for the next evaluation (index_count = num_pars).
  subroutine ad_grad(num_pars)
    integer, intent(in) :: num_pars
    integer :: k, code, i1, i2, iy, ic
    real(kp) :: a, c
    max_trace_count = max(max_trace_count, trace_count)
    max_index_count = max(max_index_count, index_count)
    max_const_count = max(max_const_count, const_count)
    if (index_count > 0) then
       adjoints(:index_count) = 0.0_kp
       adjoints(index_count) = 1.0_kp
    end if
    ic = const_count
    do k = trace_count - 3, 1, -4
       code = trace(k); i1 = trace(k+1); i2 = trace(k+2); iy = trace(k+3)
       a = adjoints(iy)
       c = 0.0_kp
       if (code >= 100) then
          c = ad_constants(ic); ic = ic - 1
       end if
       select case (code)
%(cases)s
       case default
          error stop 'module ad: corrupt trace'
       end select
    end do
    trace_count = 0; const_count = 0; index_count = num_pars
  end subroutine ad_grad

  ! what was requested at ad_init_reverse and the largest use seen by ad_grad (elements and bytes)
  subroutine ad_memory_report(io_unit)
    use, intrinsic :: iso_fortran_env, only: output_unit
    integer, intent(in), optional :: io_unit
    integer :: u
    u = output_unit
    if (present(io_unit)) u = io_unit
    if (.not. allocated(trace)) then
       write(u, '(1x, a)') 'AD memory usage: reverse mode is not initialized'
       return
    end if
    write(u, '(1x, a)') 'AD memory usage'
    write(u, '(1x, a)') '==============='
    write(u, '(2x, a)') 'Requested:'
    write(u, '(2x, a, i0, a, i0, a)') 'forward+adjoints: ', 2*kp*size(adjoints), ' B (2x', size(adjoints), ')'
    write(u, '(13x, a, i0, a, i0, a)') 'trace: ', kind(1)*size(trace), ' B (', size(trace), ')'
    write(u, '(13x, a, i0, a, i0, a)') 'const: ', kp*size(ad_constants), ' B (', size(ad_constants), ')'
    write(u, '(13x, a, i0, a)') 'Total: ', kp*(2*size(adjoints) + size(ad_constants)) + kind(1)*size(trace), ' B'
    write(u, '(/, 2x, a)') 'Used:'
    write(u, '(2x, a, i0, a, i0, a)') 'forward+adjoints: ', 2*kp*max_index_count, ' B (2x', max_index_count, ')'
    write(u, '(13x, a, i0, a, i0, a)') 'trace: ', kind(1)*max_trace_count, ' B (', max_trace_count, ')'
    write(u, '(13x, a, i0, a, i0, a)') 'const: ', kp*max_const_count, ' B (', max_const_count, ')'
    write(u, '(13x, a, i0, a)') 'Total: ', kp*(2*max_index_count + max_const_count) + kind(1)*max_trace_count, ' B'
  end subroutine ad_memory_report
'''

def _case(code, stmts):
    body = '\n'.join('          ' + st.strip() for st in stmts.split(';'))
    return '       case (%s)\n%s' % (code, body)


CASES = []
for _n in ['add', 'subtract', 'multiply', 'divide', 'power']:
    CASES.append(_case(GOP[_n], REV_AA[_n]))
    CASES.append(_case(GOP[_n] + ' + 100', REV_AR[_n]))
    CASES.append(_case(GOP[_n] + ' + 200', REV_RA[_n]))
CASES.append(_case('GFH_POWI', 'adjoints(i1) = adjoints(i1) + a*i2*forward_values(i1)**(i2 - 1)'))
for _u, _st in REV_UN.items():
    CASES.append(_case('GFH_' + _u.upper(), _st))
REVPLUMB = PLUMB.replace('%(cases)s', '\n'.join(CASES))

out = []
w = out.append

w('''! GENERATED by gen_ad.py -- do not edit.
!
! module ad: the AD variable of the fitting-function API, as a model RECORDER.
!
! Drop-in for the reference's module ad (fortran/gadfit/automatic_differentiation.F90):
! same public type(advar) components (val, d, dd, index), same operators, assignments and
! elemental function names, so a user's fitfunc%eval compiles unchanged.  The reference
! differentiates on the host by recording a tape per data point; here eval() is run a few
! times on the host only to CAPTURE the operation sequence (model tape,
! include/gadfit_tape.h), which libgadfit_hip lowers to a HIP kernel that does the
! per-point arithmetic on the GPU.  Values (val) are always computed, and so are the forward-mode
! derivatives (d, dd) of active operands (index /= 0) with the reference's formulas, so eval() also
! works on the host as a plain evaluator and as a forward-mode differentiator.  The reference's host-side
! reverse mode is here too, with its public names (ad_init_reverse, ad_grad, forward_values, adjoints, trace,
! ad_constants, the counters, reverse_mode, ad_memory_report, ad_close; AD:233-313, 1476-1690): a stand-alone AD
! library for user code -- gadf_fit itself differentiates on the device.
module ad

  use, intrinsic :: iso_c_binding
  use gadf_constants, only: kp, dp, qp, real32
  use messaging
  use misc, only: safe_deallocate

  implicit none

  public      ! (wholesale, as the reference's module: messaging's procedures and the safe_deallocate generic travel with `use ad`, AD:26-31)

  ! the advar specific of misc's generic (AD:92-96, 1693-1702)
  interface safe_deallocate
     module procedure safe_deallocate_advar
  end interface safe_deallocate

  ! enum gfh_op (include/gadfit_tape.h)
  integer, parameter :: GFH_CONST = 0, GFH_X = 1, GFH_PARAM = 2, GFH_LIFT = 3, GFH_NEG = 4, GFH_VAL = 8, &
       & GFH_ADD = 10, GFH_SUB = 11, GFH_MUL = 12, GFH_DIV = 13, GFH_POW = 14, GFH_POWI = 15, &
       & GFH_ABS = 20, GFH_EXP = 21, GFH_SQRT = 22, GFH_LOG = 23, GFH_SIN = 24, GFH_COS = 25, &
       & GFH_TAN = 26, GFH_ASIN = 27, GFH_ACOS = 28, GFH_ATAN = 29, GFH_SINH = 30, &
       & GFH_COSH = 31, GFH_TANH = 32, GFH_ASINH = 33, GFH_ACOSH = 34, GFH_ATANH = 35, &
       & GFH_ERF = 36
  integer, parameter :: GFH_F_REAL = 1

  ! The AD variable (AD:65-80).  node: position in the model tape while capturing.
  type advar
     real(kp) :: val = 0.0_kp, d = 0.0_kp, dd = 0.0_kp
     integer :: index = 0
     integer :: node = -1
   contains
     procedure :: assign_advar_integer
     procedure, pass(this) :: assign_integer_advar
     procedure :: assign_advar_real32
     procedure, pass(this) :: assign_real32_advar
     procedure :: assign_advar_dp
     procedure, pass(this) :: assign_dp_advar
     procedure :: assign_advar_qp
     procedure, pass(this) :: assign_qp_advar
     generic :: assignment(=) => assign_advar_integer, assign_integer_advar, &
          & assign_advar_real32, assign_real32_advar, assign_advar_dp, &
          & assign_dp_advar, assign_advar_qp, assign_qp_advar
  end type advar

  ! One tape node (struct gfh_node)
  type, bind(c) :: gfh_node
     integer(c_int32_t) :: op, a, b, flags
     real(c_double) :: c
  end type gfh_node

  ! One integrate() call site (struct gfh_integral)
  type, bind(c) :: gfh_integral
     integer(c_int32_t) :: integrand, lower, upper, lower_inf, upper_inf, n_ipars, ipar_off, depth
     real(c_double) :: rel_error, abs_error
  end type gfh_integral

  integer, parameter :: GFH_IVAR = 5, GFH_IPARAM = 6, GFH_AUX = 7, GFH_INTEGRATE = 40
  integer, parameter :: GFH_GUARD_GT = 50, GFH_GUARD_LT = 51, GFH_F_TAKEN = 2
  integer, parameter :: AD_MAX_SUB = 16

  ! Comparisons met while recording (AD:315-395).  A recording follows ONE path through eval(): every comparison of AD
  ! variables becomes a guard node with the outcome it had.  The first ad_script_n outcomes can be FORCED (ad_script): the
  ! recording then follows a prescribed path wherever the values themselves would have led -- used to record the branch a
  ! data point took on the device, and to probe a path at other abscissas.  ad_guard_count: comparisons met so far.
  logical, allocatable :: ad_script(:)
  integer :: ad_script_n = 0, ad_guard_count = 0
  ! where in its range the integration variable sits while an integrand is recorded (numerical_integration.F90, probe_at): a
  ! comparison inside an integrand is decided there, and gadfit.F90 records such integrands at several places of the range
  real(kp) :: ad_theta = 0.5_kp

  ! Checking mode (ad_check_begin): the recording is compared, node by node as it is made, with a recording known already --
  ! same operation, operands, flags and sub-tape, and for real literals the value the known path's classification predicts
  ! (ad_chk_cls 1: the constant ad_chk_c; 2: ad_chk_alpha*x + ad_chk_beta at the abscissa ad_chk_x; other: anything) -- and
  ! nothing is stored.  This is what recording eval() over a large data set costs per point once its paths are known; the
  ! first disagreement sets ad_chk_diverged (another path: the caller records the point again, storing) or ad_chk_litfail
  ! (a literal is not what it was taken for).
  ! ad_need_vals = .false. (only ever inside checking mode, for a known path WITHOUT comparisons): the elementals skip the value
  ! arithmetic (exp, pow, divisions ...) -- nothing in such a recording depends on an advar's value; what the user's code computes in
  ! plain real arithmetic is untouched.  Should eval() read a %val after all, the literal it forms from it fails the check and the
  ! point is recorded again with values.
  logical :: ad_need_vals = .true.
  ! ad_thread_check (set by gadfit.F90 discover around its parallel loop): recordings are made on several threads at once, each
  ! compared with the known recording through per-thread state kept by libgadfit_hip (gfh_adchk_*, include/gadfit_hip.h: native
  ! thread-local storage -- module variables are shared, and flang's threadprivate costs a runtime call per access).  In this mode
  ! nothing of the capture state in this module is written.
  logical :: ad_thread_check = .false.
  ! ad_fast_check = ad_thread_check and no value is needed (ad_need_vals false: the known recording compares no AD variables and forms
  ! no real from a %val): an elemental then is ONE call that compares its node(s) with the known recording (ad_tls.c, gfh_adchk_op2 /
  ! _op_lit / _lift) and returns -- no value, no derivative, no index bookkeeping.  Half the cost of a check per point (round 5:
  ! the every-abscissa capture of the headline model 0.79 -> 0.4 us per point and thread).  Set and cleared with ad_thread_check.
  logical :: ad_fast_check = .false.
  interface
     integer(c_int) function gfh_adchk_op2(op, a, b) bind(c, name='gfh_adchk_op2')
       import c_int
       integer(c_int), value :: op, a, b
     end function gfh_adchk_op2
     integer(c_int) function gfh_adchk_op_lit(op, a, r, lit_first) bind(c, name='gfh_adchk_op_lit')
       import c_int, c_double
       integer(c_int), value :: op, a, lit_first
       real(c_double), value :: r
     end function gfh_adchk_op_lit
     integer(c_int) function gfh_adchk_lift(r) bind(c, name='gfh_adchk_lift')
       import c_int, c_double
       real(c_double), value :: r
     end function gfh_adchk_lift
     integer(c_int) function gfh_adchk_emit(op, a, b, flags, c) bind(c, name='gfh_adchk_emit')
       import c_int, c_double
       integer(c_int), value :: op, a, b, flags
       real(c_double), value :: c
     end function gfh_adchk_emit
     integer(c_int) function gfh_adchk_guard(natural) bind(c, name='gfh_adchk_guard')
       import c_int
       integer(c_int), value :: natural
     end function gfh_adchk_guard
     ! (integrate() recorded in thread-checking mode: module numerical_integration, record_integral)
     integer(c_int) function gfh_adchk_depth() bind(c, name='gfh_adchk_depth')
       import c_int
     end function gfh_adchk_depth
     subroutine gfh_adchk_ipar(n, nodes) bind(c, name='gfh_adchk_ipar')
       import c_int, c_int32_t
       integer(c_int), value :: n
       integer(c_int32_t), intent(in) :: nodes(*)
     end subroutine gfh_adchk_ipar
     integer(c_int) function gfh_adchk_sub_enter() bind(c, name='gfh_adchk_sub_enter')
       import c_int
     end function gfh_adchk_sub_enter
     subroutine gfh_adchk_sub_leave(result) bind(c, name='gfh_adchk_sub_leave')
       import c_int
       integer(c_int), value :: result
     end subroutine gfh_adchk_sub_leave
     integer(c_int) function gfh_adchk_integral(integrand, lower, upper, linf, uinf, nip, rel, abs_) bind(c, name='gfh_adchk_integral')
       import c_int, c_double
       integer(c_int), value :: integrand, lower, upper, linf, uinf, nip
       real(c_double), value :: rel, abs_
     end function gfh_adchk_integral
  end interface
  logical :: ad_checking = .false., ad_chk_diverged = .false., ad_chk_litfail = .false.
  integer :: ad_chk_n = 0
  integer, allocatable :: ad_chk_op(:), ad_chk_a(:), ad_chk_b(:), ad_chk_fl(:), ad_chk_sub(:), ad_chk_cls(:)
  real(kp), allocatable :: ad_chk_c(:), ad_chk_alpha(:), ad_chk_beta(:)
  real(kp) :: ad_chk_x = 0.0_kp


  ! Capture state.  All sub-tapes (0 = eval(), 1.. = integrands) share one flat node array;
  ! ad_sub(k) tags the sub-tape of node k and node indices are local to their sub-tape.
  logical :: ad_recording = .false.
  type(gfh_node), allocatable, target :: ad_tape(:)
  integer, allocatable :: ad_sub(:)
  integer :: ad_tape_n = 0
  integer :: ad_cur = 0, ad_nsub = 0, ad_depth = 0
  integer :: ad_sub_n(0:AD_MAX_SUB) = 0, ad_sub_result(0:AD_MAX_SUB) = -1
  type(gfh_integral), allocatable, target :: ad_integrals(:)
  integer, allocatable :: ad_int_sub(:)          ! enclosing sub-tape of each integral
  integer :: ad_n_integrals = 0
  integer(c_int32_t), allocatable, target :: ad_ipar_nodes(:)
  integer :: ad_n_ipar = 0
  logical :: ad_capture_failed = .false.
  character(len=256) :: ad_capture_msg = ''

  ! Host-side reverse mode (AD:233-250): intermediate values, adjoints, constants, execution trace; numbers of
  ! used elements; the mode switch (forward mode if .false.); high-water marks for ad_memory_report.
  integer, parameter :: DEFAULT_SWEEP_SIZE = 10000
  real(kp), allocatable :: forward_values(:), adjoints(:), ad_constants(:)
  integer, allocatable :: trace(:)
  integer :: trace_count = 0, index_count = 0, const_count = 0
  logical :: reverse_mode = .false.
  integer :: max_trace_count = 0, max_index_count = 0, max_const_count = 0

''')

for name, op, *_ in BIN:
    specs = ['%s_advar_advar' % name]
    for t, _ in RTYPES:
        specs += ['%s_advar_%s' % (name, t), '%s_%s_advar' % (name, t)]
    if name == 'subtract':
        specs.append('subtract_advar')
    w('  interface operator(%s)' % op)
    w('     module procedure ' + ', &\n          & '.join(specs))
    w('  end interface operator(%s)\n' % op)

for c, nm in [('>', 'gt'), ('<', 'lt')]:
    specs = ['advar_%s_advar' % nm]
    for t in ['real32', 'dp', 'qp']:
        specs += ['advar_%s_%s' % (nm, t), '%s_%s_advar' % (t, nm)]
    w('  interface operator(%s)' % c)
    w('     module procedure ' + ', &\n          & '.join(specs))
    w('  end interface operator(%s)\n' % c)

for u in UNARY:
    w('  interface %s\n     module procedure %s_advar\n  end interface %s\n' % (u, u, u))

w('''contains

  ! ---------------------------------------------------------------- capture plumbing
  subroutine ad_capture_begin()
    if (.not. allocated(ad_tape)) allocate(ad_tape(1024), ad_sub(1024))
    if (.not. allocated(ad_integrals)) allocate(ad_integrals(16), ad_int_sub(16), ad_ipar_nodes(256))
    ad_tape_n = 0
    ad_cur = 0; ad_nsub = 0; ad_depth = 0
    ad_sub_n = 0; ad_sub_result = -1
    ad_n_integrals = 0; ad_n_ipar = 0
    ad_guard_count = 0
    ad_recording = .true.
    ad_capture_failed = .false.
    ad_capture_msg = ''
  end subroutine ad_capture_begin

  ! the recordings that follow are compared with the known recording loaded into ad_chk_* instead of being stored (x: the
  ! abscissa they are made at); ad_check_end returns to storing
  subroutine ad_check_begin(x, need_vals)
    real(kp), intent(in) :: x
    logical, intent(in) :: need_vals
    ad_checking = .true.; ad_chk_diverged = .false.; ad_chk_litfail = .false.; ad_chk_x = x
    ad_need_vals = need_vals
  end subroutine ad_check_begin
  subroutine ad_check_end()
    ad_checking = .false.; ad_need_vals = .true.
  end subroutine ad_check_end

  ! outcomes to force on the first n comparisons of the recordings that follow (n = 0: none)
  subroutine ad_set_script(n, outcomes)
    integer, intent(in) :: n
    logical, intent(in), optional :: outcomes(:)
    if (n > 0) then
       if (.not. allocated(ad_script)) allocate(ad_script(max(64, n)))
       if (size(ad_script) < n) then
          deallocate(ad_script); allocate(ad_script(n))
       end if
       ad_script(:n) = outcomes(:n)
    end if
    ad_script_n = n
  end subroutine ad_set_script

  ! one comparison of values while recording: the guard node, and the outcome the recording continues with
  logical function ad_guard(op, na, nb, natural) result(y)
    integer, intent(in) :: op, na, nb
    logical, intent(in) :: natural
    integer :: k
    y = natural
    if (ad_thread_check) then
       if (gfh_adchk_depth() > 0) then               ! (inside an integrand: the natural outcome, as below; the depth is the thread's own)
          k = ad_emit(op, na, nb, merge(GFH_F_TAKEN, 0, y), 0.0_kp)
          return
       end if
    else if (ad_depth > 0) then
       ! inside an integrand: decided by the values at the abscissa the integrand is being recorded at (ad_theta of the way through
       ! its range); never forced, not counted among eval()'s comparisons.  The device decides it anew at every evaluation of the
       ! integrand (libgadfit_hip pools the recordings of an integrand that differ in their path: Model::alts)
       k = ad_emit(op, na, nb, merge(GFH_F_TAKEN, 0, y), 0.0_kp)
       return
    end if
    if (ad_thread_check) then                       ! (threads write nothing of this module: their forced outcomes live in ad_tls.c)
       y = gfh_adchk_guard(merge(1_c_int, 0_c_int, natural)) /= 0
    else
       ad_guard_count = ad_guard_count + 1
       if (ad_guard_count <= ad_script_n) y = ad_script(ad_guard_count)
    end if
    k = ad_emit(op, na, nb, merge(GFH_F_TAKEN, 0, y), 0.0_kp)
  end function ad_guard

  subroutine ad_capture_end()
    ad_recording = .false.
  end subroutine ad_capture_end

  integer function ad_emit(op, a, b, flags, c) result(k)
    integer, intent(in) :: op, a, b, flags
    real(kp), intent(in) :: c
    integer :: j
    real(kp) :: want
    if (ad_thread_check) then
       k = gfh_adchk_emit(op, a, b, flags, c)
       return
    end if
    if (ad_checking) then
       j = ad_tape_n + 1
       ad_tape_n = j
       k = ad_sub_n(ad_cur)
       ad_sub_n(ad_cur) = k + 1
       if (ad_chk_diverged) return
       if (j > ad_chk_n) then
          ad_chk_diverged = .true.; return
       end if
       if (op /= ad_chk_op(j) .or. a /= ad_chk_a(j) .or. b /= ad_chk_b(j) .or. flags /= ad_chk_fl(j) .or. ad_cur /= ad_chk_sub(j)) then
          ad_chk_diverged = .true.; return
       end if
       if (op == GFH_CONST) then
          if (ad_chk_cls(j) == 1) then
             if (c /= ad_chk_c(j) .and. .not. (c /= c .and. ad_chk_c(j) /= ad_chk_c(j))) ad_chk_litfail = .true.
          else if (ad_chk_cls(j) == 2) then
             want = ad_chk_alpha(j)*ad_chk_x + ad_chk_beta(j)
             if (.not. (abs(want - c) <= 1e-11_kp*(abs(c) + abs(ad_chk_alpha(j)*ad_chk_x) + abs(ad_chk_beta(j))))) ad_chk_litfail = .true.
          end if
       end if
       return
    end if
    if (ad_tape_n == size(ad_tape)) call ad_grow()      ! (kept out of line: allocatable locals here would be set up on every call)
    ad_tape_n = ad_tape_n + 1
    ad_tape(ad_tape_n)%op = op
    ad_tape(ad_tape_n)%a = a
    ad_tape(ad_tape_n)%b = b
    ad_tape(ad_tape_n)%flags = flags
    ad_tape(ad_tape_n)%c = c
    ad_sub(ad_tape_n) = ad_cur
    k = ad_sub_n(ad_cur)            ! 0-based node index inside the current sub-tape
    ad_sub_n(ad_cur) = k + 1
  end function ad_emit

  ! the parameter nodes 0 .. np-1 with which every recording of eval() begins, in one go (checking mode: they are what the
  ! known recording begins with, nothing to compare)
  subroutine ad_emit_params(np)
    integer, intent(in) :: np
    integer :: k
    do while (size(ad_tape) < np)
       call ad_grow()
    end do
    if (.not. ad_checking) then
       do k = 1, np
          ad_tape(k)%op = GFH_PARAM; ad_tape(k)%a = k - 1; ad_tape(k)%b = -1; ad_tape(k)%flags = 0; ad_tape(k)%c = 0.0_kp
          ad_sub(k) = 0
       end do
    else if (np > ad_chk_n) then
       ad_chk_diverged = .true.
    end if
    ad_tape_n = np
    ad_sub_n(0) = np
  end subroutine ad_emit_params

  subroutine ad_grow()
    type(gfh_node), allocatable :: tmp(:)
    integer, allocatable :: itmp(:)
    allocate(tmp(2*size(ad_tape)), itmp(2*size(ad_tape)))
    tmp(:ad_tape_n) = ad_tape(:ad_tape_n)
    itmp(:ad_tape_n) = ad_sub(:ad_tape_n)
    call move_alloc(tmp, ad_tape)
    call move_alloc(itmp, ad_sub)
  end subroutine ad_grow

  ! node of a real operand: a literal slot (x-dependence is detected by probing, see
  ! gadfit.F90 capture_model)
  integer function rnode(r) result(k)
    real(kp), intent(in) :: r
    k = ad_emit(GFH_CONST, -1, -1, GFH_F_REAL, r)
  end function rnode

  ! node of an advar operand; an advar that was never produced by a recorded operation
  ! (e.g. assigned before capture) enters as a lifted literal
  integer function anode(x) result(k)
    type(advar), intent(in) :: x
    if (x%node >= 0) then
       k = x%node
    else
       k = ad_emit(GFH_LIFT, rnode(x%val), -1, 0, 0.0_kp)
    end if
  end function anode

  subroutine ad_fail(msg)
    character(*), intent(in) :: msg
    ad_capture_failed = .true.
    ad_capture_msg = msg
  end subroutine ad_fail
''')
w(REVPLUMB)
w('''  ! ---------------------------------------------------------------- comparisons (AD:315-395)
  ! They compare val only.  While a model is being captured the comparison is recorded as a guard node (ad_guard) and the
  ! recording continues along its outcome, so eval() may branch as it does under the reference.''')

for c, nm in [('>', 'gt'), ('<', 'lt')]:
    G = 'GFH_GUARD_' + nm.upper()
    w('''  impure elemental logical function advar_%(nm)s_advar(x1, x2) result(y)
    type(advar), intent(in) :: x1, x2
    integer :: n1
    y = x1%%val %(c)s x2%%val
    if (ad_recording) then
       n1 = anode(x1)
       y = ad_guard(%(G)s, n1, anode(x2), y)
    end if
  end function advar_%(nm)s_advar
''' % dict(nm=nm, c=c, G=G))
    for t, decl in RTYPES[:3]:
        w('''  impure elemental logical function advar_%(nm)s_%(t)s(x1, x2) result(y)
    type(advar), intent(in) :: x1
    %(decl)s, intent(in) :: x2
    integer :: n1
    y = x1%%val %(c)s x2
    if (ad_recording) then
       n1 = anode(x1)
       y = ad_guard(%(G)s, n1, rnode(real(x2, kp)), y)
    end if
  end function advar_%(nm)s_%(t)s

  impure elemental logical function %(t)s_%(nm)s_advar(x1, x2) result(y)
    %(decl)s, intent(in) :: x1
    type(advar), intent(in) :: x2
    integer :: n1
    y = x1 %(c)s x2%%val
    if (ad_recording) then
       n1 = rnode(real(x1, kp))
       y = ad_guard(%(G)s, n1, anode(x2), y)
    end if
  end function %(t)s_%(nm)s_advar
''' % dict(nm=nm, c=c, t=t, decl=decl, G=G))

w('  ! ---------------------------------------------------------------- assignments (AD:401-447)')
for t, decl in RTYPES:
    conv = 'real(x, kp)'
    back = {'integer': 'int(this%val)', 'real32': 'real(this%val, real32)', 'dp': 'real(this%val, dp)',
            'qp': 'real(this%val, qp)'}[t]
    w('''  impure elemental subroutine assign_advar_%(t)s(this, x)
    class(advar), intent(out) :: this
    %(decl)s, intent(in) :: x
    this%%val = %(conv)s
    if (ad_fast_check) then
       this%%node = gfh_adchk_lift(this%%val)
       return
    end if
    if (ad_recording) this%%node = ad_emit(GFH_LIFT, rnode(this%%val), -1, 0, 0.0_kp)
  end subroutine assign_advar_%(t)s

  impure elemental subroutine assign_%(t)s_advar(x, this)
    %(decl)s, intent(out) :: x
    class(advar), intent(in) :: this
    x = %(back)s
  end subroutine assign_%(t)s_advar
''' % dict(t=t, decl=decl, conv=conv, back=back))

w('  ! ---------------------------------------------------------------- binary elementals (AD:454-1108)')
for name, op, gop, vaa, var, vra in BIN:
    cr = '1/x2%val' if name == 'divide' else 'x2%val'
    aa = fwd(FWD_AA[name], 'x1%index /= 0 .and. x2%index /= 0', 'call ad_push(%s, x1%%index, x2%%index, y)' % gop)
    ar = fwd(FWD_AR[name].replace('r2', 'x2%val'), 'x1%index /= 0 .and. x2%index == 0',
             'call ad_push(%s + 100, x1%%index, 0, y, %s)' % (gop, cr))
    ra = fwd(FWD_RA[name].replace('r1', 'x1%val'), 'x1%index == 0 .and. x2%index /= 0',
             'call ad_push(%s + 200, x2%%index, 0, y, x1%%val)' % gop)
    val = '    if (ad_need_vals) y%%val = %s' % vaa
    if name == 'divide':
        # an ACTIVE numerator is multiplied by the reciprocal (AD:817-819 both active, 826-827 -> divide_advar_real 845-847), a passive
        # one is divided (AD:828-831 -> divide_real_advar 895): the reference's known answers hold to the last bit only with both forms
        val = ('    if (ad_need_vals) then\n       if (x1%index /= 0) then\n          y%val = x1%val*(1/x2%val)\n       else\n'
               '          y%val = x1%val/x2%val\n       end if\n    end if')
    w('''  type(advar) function %(name)s_advar_advar(x1, x2) result(y)
    type(advar), intent(in) :: x1, x2
    real(kp) :: t
    if (ad_fast_check) then                       ! (checking mode without values: one call, nothing else -- see ad_fast_check)
       if (x1%%node >= 0 .and. x2%%node >= 0) then
          y%%node = gfh_adchk_op2(%(gop)s, x1%%node, x2%%node)
          return
       end if
    end if
%(val)s
%(aa)s
%(ar)s
%(ra)s
    if (ad_recording) y%%node = ad_emit(%(gop)s, anode(x1), anode(x2), 0, 0.0_kp)
  end function %(name)s_advar_advar
''' % dict(name=name, val=val, gop=gop, aa=aa, ar=ar, ra=ra))
    for t, decl in RTYPES:
        if name == 'power' and t == 'integer':
            # power_advar_integer has its own op (AD:1033-1059)
            w('''  type(advar) function power_advar_integer(x1, x2) result(y)
    type(advar), intent(in) :: x1
    integer, intent(in) :: x2
    real(kp) :: t
    if (ad_fast_check) then
       if (x1%node >= 0) then
          y%node = gfh_adchk_op2(GFH_POWI, x1%node, x2)
          return
       end if
    end if
    if (ad_need_vals) y%val = x1%val**x2
    if (x1%index /= 0) then                       ! AD:1044-1054
       if (reverse_mode) then
          call ad_push(GFH_POWI, x1%index, x2, y)
       else
          t = 1/x1%val
          y%d = y%val*x2*x1%d*t
          y%dd = y%d**2/y%val + y%val*x2*(x1%dd - x1%d**2*t)*t
          y%index = 1
       end if
    end if
    if (ad_recording) y%node = ad_emit(GFH_POWI, anode(x1), x2, 0, 0.0_kp)
  end function power_advar_integer
''')
        else:
            w('''  type(advar) function %(name)s_advar_%(t)s(x1, x2) result(y)
    type(advar), intent(in) :: x1
    %(decl)s, intent(in) :: x2
    real(kp) :: r2, t
    integer :: n1
    r2 = real(x2, kp)
    if (ad_fast_check) then
       if (x1%%node >= 0) then
          y%%node = gfh_adchk_op_lit(%(gop)s, x1%%node, r2, 0)
          return
       end if
    end if
    if (ad_need_vals) y%%val = %(var)s
%(f)s
    if (ad_recording) then
       n1 = anode(x1)
       y%%node = ad_emit(%(gop)s, n1, rnode(r2), 0, 0.0_kp)
    end if
  end function %(name)s_advar_%(t)s
''' % dict(name=name, t=t, decl=decl, var=var, gop=gop,
           f=fwd(FWD_AR[name], 'x1%index /= 0', 'call ad_push(%s + 100, x1%%index, 0, y, %s)' % (gop, '1/r2' if name == 'divide' else 'r2'))))
        w('''  type(advar) function %(name)s_%(t)s_advar(x1, x2) result(y)
    %(decl)s, intent(in) :: x1
    type(advar), intent(in) :: x2
    real(kp) :: r1, t
    integer :: n1
    r1 = real(x1, kp)
    if (ad_fast_check) then
       if (x2%%node >= 0) then
          y%%node = gfh_adchk_op_lit(%(gop)s, x2%%node, r1, 1)
          return
       end if
    end if
    if (ad_need_vals) y%%val = %(vra)s
%(f)s
    if (ad_recording) then
       n1 = rnode(r1)
       y%%node = ad_emit(%(gop)s, n1, anode(x2), 0, 0.0_kp)
    end if
  end function %(name)s_%(t)s_advar
''' % dict(name=name, t=t, decl=decl, vra=vra, gop=gop,
           f=fwd(FWD_RA[name], 'x2%index /= 0', 'call ad_push(%s + 200, x2%%index, 0, y, r1)' % gop)))

w('''  ! unary minus: -a = 0 - a (AD:598-601)
  type(advar) function subtract_advar(x) result(y)
    type(advar), intent(in) :: x
    y = subtract_dp_advar(0.0_dp, x)
  end function subtract_advar

  ! ---------------------------------------------------------------- unary elementals (AD:939-957, 1110-1459)''')
for u in UNARY:
    w('''  type(advar) function %(u)s_advar(x) result(y)
    type(advar), intent(in) :: x
    real(kp) :: t
    if (ad_fast_check) then
       if (x%%node >= 0) then
          y%%node = gfh_adchk_op2(GFH_%(U)s, x%%node, -1)
          return
       end if
    end if
    if (ad_need_vals) y%%val = %(u)s(x%%val)
%(f)s
    if (ad_recording) y%%node = ad_emit(GFH_%(U)s, anode(x), -1, 0, 0.0_kp)
  end function %(u)s_advar
''' % dict(u=u, U=u.upper(), f=fwd(FWD_UN[u], 'x%index /= 0', 'call ad_push(GFH_%s, x%%index, 0, y)' % u.upper())))

w('end module ad')
print('\n'.join(out))
