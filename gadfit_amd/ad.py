"""advar -- the automatic-differentiation variable of the fitting-function API.

Host-side mirror of the reference's ``module ad``
(fortran/gadfit/automatic_differentiation.F90:65-229): the same operator set
(``+ - * / **``, unary minus, ``abs exp sqrt log sin cos tan asin acos atan sinh cosh tanh
asinh acosh atanh erf``) with the same (advar,advar) / (advar,real) / (real,advar) /
``**integer`` overload resolution.  Unlike the reference, evaluating a model does not
compute anything: it RECORDS the operation sequence once into a model tape
(include/gadfit_tape.h).  The device code generator lowers that tape to registers; the
per-point arithmetic then happens in the HIP sweep kernel, one lane per data point.

``x`` reaches ``eval`` as a ``Real`` (a symbolic real(kp)); plain-real arithmetic on it
(``-x``, ``x-1.0``) is recorded as real-typed nodes so the (real,advar) variants are
selected exactly as the Fortran compiler would select them.

Comparisons (``>``, ``<``: AD:315-395, values only) make eval() branch.  A recording at a CONCRETE point
(``trace_model(fn, n, x=..., pars=...)``, ``tape.Variants``) carries values along, decides every comparison
from them -- or from a forced ``script`` of outcomes -- and records it as a guard node; one recording is one path
through eval().  A purely symbolic recording (no ``x``) cannot decide a comparison and refuses it.
"""
import math
import numbers

import numpy as np

from . import tape as T

__all__ = ['advar', 'Real', 'INFINITY', 'integrate', 'trace_model',
           'exp', 'sqrt', 'log', 'sin', 'cos', 'tan', 'asin', 'acos', 'atan', 'sinh', 'cosh',
           'tanh', 'asinh', 'acosh', 'atanh', 'erf']


class _Inf:
    """INFINITY sentinel (numerical_integration.F90:36); ``-INFINITY`` is its negation."""

    def __init__(self, sign=1):
        self.sign = sign

    def __neg__(self):
        return _Inf(-self.sign)

    def __repr__(self):
        return 'INFINITY' if self.sign > 0 else '-INFINITY'


INFINITY = _Inf(1)


class _Rec:
    """Recording context: a stack of sub-tapes being built."""

    def __init__(self, n_pars, concrete=False, script=None, theta=0.5):
        self.tape = T.Tape(n_pars)
        self.stack = []          # indices into self.tape.subtapes of open sub-tapes
        self.depth = 0           # integrate nesting depth of the code being recorded
        self.concrete = concrete # values travel with the nodes: comparisons can be decided
        self.script = list(script) if script is not None else []   # forced outcomes of the first comparisons OF eval()
        self.n_guards = 0        # comparisons met in eval() itself so far (those inside integrands always take their natural outcome)
        self.theta = float(theta)   # where in its range the integration variable sits while an integrand is recorded at a point
        self.leaves = {}         # (sub-tape, leaf) -> node: the abscissa / auxiliary inputs re-emitted inside integrands (_Sym._n)

    def open(self):
        self.tape.subtapes.append(([], -1))
        self.stack.append(len(self.tape.subtapes) - 1)
        return self.stack[-1]

    def close(self, result):
        idx = self.stack.pop()
        nodes, _ = self.tape.subtapes[idx]
        self.tape.subtapes[idx] = (nodes, result)
        return idx

    @property
    def cur(self):
        return self.stack[-1]

    def emit(self, op, a=-1, b=-1, flags=0, c=0.0):
        nodes, _ = self.tape.subtapes[self.cur]
        nodes.append((op, int(a), int(b), int(flags), float(c)))
        return len(nodes) - 1


_rec = None  # the active recorder


def _need_rec():
    if _rec is None:
        raise RuntimeError('advar arithmetic outside of model tracing '
                           '(use gadfit_amd.ad.trace_model / gadf_init)')
    return _rec


class _Sym:
    __slots__ = ('sub', 'node', 'val', 'leaf')

    def __init__(self, node, val=None):
        self.sub = _need_rec().cur
        self.node = node
        self.val = val           # the value at the recording point (concrete recordings), else None
        self.leaf = None         # ('x',) / ('aux', k): the abscissa / an auxiliary input itself (may enter integrands, _n)

    def _n(self):
        r = _need_rec()
        if self.sub != r.cur and getattr(self, 'leaf', None) is not None and r.depth > 0:
            # the data point's abscissa (or an auxiliary per-point input) taken into an integrand from the enclosing eval() without
            # passing it through pars(:): the reference evaluates the integrand afresh in that scope (NI:195-201); here it is a leaf of
            # the integrand's own sub-tape
            key = (r.cur, self.leaf)
            if key not in r.leaves:
                r.leaves[key] = r.emit(T.X, flags=T.F_REAL) if self.leaf[0] == 'x' else r.emit(T.AUX, self.leaf[1], -1, T.F_REAL)
            return r.leaves[key]
        if self.sub != r.cur:
            raise RuntimeError('a value from an enclosing scope was used inside an integrand; '
                               'pass it through the integrand\'s pars(:) array like the '
                               'reference API requires')
        return self.node

    def __bool__(self):
        raise TypeError('data-dependent control flow on a traced value is not representable '
                        'on the device path')

    # comparisons compare %val in the reference (AD:315-395): advar/advar, advar/real, real/advar, `>` and `<` only
    def _guard(self, op, other, swap=False):
        r = _need_rec()
        if isinstance(other, _Sym):
            nb, vb = other._n(), other.val
        elif _is_num(other):
            nb, vb = _real_node(other), float(other)
        else:
            return NotImplemented
        na, va = self._n(), self.val
        if swap:
            na, nb, va, vb = nb, na, vb, va
        if r.depth > 0:
            # inside an integrand: the comparison is decided by the values at the abscissa the integrand is being recorded at
            # (trace_model(..., theta=)); the device decides it anew at every evaluation of the integrand
            if not r.concrete or va is None or vb is None:
                raise TypeError('comparison inside an integrand: record eval() at concrete points (gadfit_amd.tape.Variants), '
                                'which samples the integration variable over its range')
            out = (va > vb) if op == T.GUARD_GT else (va < vb)
            r.emit(op, na, nb, T.F_TAKEN if out else 0)
            return out
        k = r.n_guards
        r.n_guards += 1
        if k < len(r.script):
            out = bool(r.script[k])
        elif not r.concrete or va is None or vb is None:
            raise TypeError('comparison of traced values: eval() branches, which a symbolic recording cannot decide -- record '
                            'it at concrete points (gadfit_amd.tape.Variants / trace_model(..., x=, pars=))')
        else:
            out = (va > vb) if op == T.GUARD_GT else (va < vb)
        r.emit(op, na, nb, T.F_TAKEN if out else 0)
        return out

    def __gt__(self, o): return self._guard(T.GUARD_GT, o)
    def __lt__(self, o): return self._guard(T.GUARD_LT, o)
    # (the reference has no >=, <=, ==: AD:82-90)
    def __le__(self, o): self.__bool__()
    def __ge__(self, o): self.__bool__()


def _is_int(v):
    return isinstance(v, (int, np.integer)) and not isinstance(v, bool)


def _is_num(v):
    return isinstance(v, (numbers.Real, np.floating, np.integer)) and not isinstance(v, bool)


def _real_node(v):
    """node index of a real-typed operand (Real or Python/numpy number -> real(kp))."""
    if isinstance(v, Real):
        return v._n()
    return _need_rec().emit(T.CONST, flags=T.F_REAL, c=float(v))


_PYF = {T.ABS: abs, T.EXP: math.exp, T.SQRT: math.sqrt, T.LOG: math.log, T.SIN: math.sin,
        T.COS: math.cos, T.TAN: math.tan, T.ASIN: math.asin, T.ACOS: math.acos,
        T.ATAN: math.atan, T.SINH: math.sinh, T.COSH: math.cosh, T.TANH: math.tanh,
        T.ASINH: math.asinh, T.ACOSH: math.acosh, T.ATANH: math.atanh, T.ERF: math.erf}


def _vof(v):
    return v.val if isinstance(v, _Sym) else float(v)


def _cval(op, va, vb=None):
    """value of an operation at the recording point (only used to decide comparisons: the device evaluates the
    guards itself, gfh_select); None when an operand has no value, NaN where Python arithmetic raises"""
    if va is None or (vb is None and op in (T.ADD, T.SUB, T.MUL, T.DIV, T.POW)):
        return None
    try:
        if op == T.ADD: return va + vb
        if op == T.SUB: return va - vb
        if op == T.MUL: return va * vb
        if op == T.DIV: return va / vb
        if op == T.POW: return float(va) ** vb
        if op == T.POWI: return float(va) ** int(vb)
        if op == T.NEG: return -va
        return _PYF[op](va)
    except (ZeroDivisionError, OverflowError, ValueError, TypeError):
        return float('nan')


class Real(_Sym):
    """A symbolic real(kp): x and plain-real arithmetic on it."""
    __slots__ = ()

    def _bin(self, op, other, swap=False):
        if isinstance(other, advar):
            return NotImplemented
        if op == T.POW and _is_int(other) and not swap:
            return Real(_need_rec().emit(T.POWI, self._n(), int(other), T.F_REAL), _cval(T.POWI, self.val, int(other)))
        if not (isinstance(other, Real) or _is_num(other)):
            return NotImplemented
        a, b = self._n(), _real_node(other)
        va, vb = self.val, _vof(other)
        if swap:
            a, b, va, vb = b, a, vb, va
        return Real(_need_rec().emit(op, a, b, T.F_REAL), _cval(op, va, vb))

    def __add__(self, o): return self._bin(T.ADD, o)
    def __radd__(self, o): return self._bin(T.ADD, o, True)
    def __sub__(self, o): return self._bin(T.SUB, o)
    def __rsub__(self, o): return self._bin(T.SUB, o, True)
    def __mul__(self, o): return self._bin(T.MUL, o)
    def __rmul__(self, o): return self._bin(T.MUL, o, True)
    def __truediv__(self, o): return self._bin(T.DIV, o)
    def __rtruediv__(self, o): return self._bin(T.DIV, o, True)
    def __pow__(self, o): return self._bin(T.POW, o)
    def __rpow__(self, o): return self._bin(T.POW, o, True)
    def __neg__(self): return Real(_need_rec().emit(T.NEG, self._n(), -1, T.F_REAL), _cval(T.NEG, self.val))
    def __pos__(self): return self
    def __abs__(self): return _unary(T.ABS, self)


class advar(_Sym):
    """type(advar) (AD:65-80).  ``advar(v)`` of a real / Real is the assignment
    ``advar = real`` (AD:401-447): a passive AD variable."""
    __slots__ = ()

    def __init__(self, v=0.0):
        if isinstance(v, advar):
            _Sym.__init__(self, v._n(), v.val)
        elif isinstance(v, (Real,)) or _is_num(v):
            r = _need_rec()
            _Sym.__init__(self, r.emit(T.LIFT, _real_node(v), -1, 0), _vof(v))
        else:
            raise TypeError('cannot make an advar from %r' % (v,))

    @classmethod
    def _from_node(cls, node, val=None):
        o = cls.__new__(cls)
        _Sym.__init__(o, node, val)
        return o

    def _bin(self, op, other, swap=False):
        r = _need_rec()
        if isinstance(other, advar):
            a, b, va, vb = self._n(), other._n(), self.val, other.val
            if swap:
                a, b, va, vb = b, a, vb, va
            return advar._from_node(r.emit(op, a, b, 0), _cval(op, va, vb))
        if op == T.POW and _is_int(other) and not swap:
            # power_advar_integer, AD:1033-1059
            return advar._from_node(r.emit(T.POWI, self._n(), int(other), 0), _cval(T.POWI, self.val, int(other)))
        if isinstance(other, Real) or _is_num(other):
            # (advar, T) / (T, advar) for T in real32/dp/qp/integer convert to real(kp)
            a, b, va, vb = self._n(), _real_node(other), self.val, _vof(other)
            if swap:
                a, b, va, vb = b, a, vb, va
            return advar._from_node(r.emit(op, a, b, 0), _cval(op, va, vb))
        return NotImplemented

    def __add__(self, o): return self._bin(T.ADD, o)
    def __radd__(self, o): return self._bin(T.ADD, o, True)
    def __sub__(self, o): return self._bin(T.SUB, o)
    def __rsub__(self, o): return self._bin(T.SUB, o, True)
    def __mul__(self, o): return self._bin(T.MUL, o)
    def __rmul__(self, o): return self._bin(T.MUL, o, True)
    def __truediv__(self, o): return self._bin(T.DIV, o)
    def __rtruediv__(self, o): return self._bin(T.DIV, o, True)
    def __pow__(self, o): return self._bin(T.POW, o)
    def __rpow__(self, o): return self._bin(T.POW, o, True)

    def __neg__(self):
        # subtract_advar, AD:598-601: -a = 0.0 - a
        return self._bin(T.SUB, 0.0, True)

    def __pos__(self): return self
    def __abs__(self): return _unary(T.ABS, self)


def _unary(op, v):
    if isinstance(v, advar):
        return advar._from_node(_need_rec().emit(op, v._n(), -1, 0), _cval(op, v.val))
    if isinstance(v, Real):
        return Real(_need_rec().emit(op, v._n(), -1, T.F_REAL), _cval(op, v.val))
    if _is_num(v):
        return _PYF[op](float(v))
    raise TypeError(v)


def value(a):
    """The VALUE of an AD variable as a real (GFH_VAL): what `p%val` is to plain real arithmetic in a Fortran eval() -- a number that
    follows the parameter, through which no derivative flows (the reference's AD never sees what real arithmetic does with it)."""
    if not isinstance(a, advar):
        raise TypeError('value() takes an AD variable')
    return Real(_need_rec().emit(T.VAL, a._n(), -1, T.F_REAL), a.val)


def exp(v): return _unary(T.EXP, v)
def sqrt(v): return _unary(T.SQRT, v)
def log(v): return _unary(T.LOG, v)
def sin(v): return _unary(T.SIN, v)
def cos(v): return _unary(T.COS, v)
def tan(v): return _unary(T.TAN, v)
def asin(v): return _unary(T.ASIN, v)
def acos(v): return _unary(T.ACOS, v)
def atan(v): return _unary(T.ATAN, v)
def sinh(v): return _unary(T.SINH, v)
def cosh(v): return _unary(T.COSH, v)
def tanh(v): return _unary(T.TANH, v)
def asinh(v): return _unary(T.ASINH, v)
def acosh(v): return _unary(T.ACOSH, v)
def atanh(v): return _unary(T.ATANH, v)
def erf(v): return _unary(T.ERF, v)


def _as_node_any(v):
    """node of an advar / Real / number in the current sub-tape (for bounds & bindings)."""
    if isinstance(v, advar):
        return v._n()
    return _real_node(v)


def integrate(f, pars, lower, upper, rel_error=None, abs_error=None):
    """``integrate(f, pars, lower, upper [, rel_error, abs_error])``
    (numerical_integration.F90:53-58, 193-630).  ``f(x, pars) -> advar`` takes the
    integration variable as an advar and ``pars`` as a list of advar; bounds may be real,
    advar or (+-)INFINITY.  Nesting depth is limited to 2 like the reference (NI:70)."""
    r = _need_rec()
    if r.depth >= 2:
        raise RuntimeError('integrals can be nested at most twice (ws(2), NI:70)')
    pars = list(pars)
    binds = [_as_node_any(p) for p in pars]
    d = dict(lower=-1, upper=-1, lower_inf=0, upper_inf=0)
    if isinstance(lower, _Inf):
        d['lower_inf'] = lower.sign
    else:
        d['lower'] = _as_node_any(lower)
    if isinstance(upper, _Inf):
        d['upper_inf'] = upper.sign
    else:
        d['upper'] = _as_node_any(upper)
    # record the integrand into its own sub-tape
    r.depth += 1
    r.open()
    tval = None
    if r.concrete:
        # a recording at a point: the integration variable takes ONE value of its range (theta of the way through, in the
        # variable the quadrature runs over: NI:314-318, 347-351 for the infinite ranges), so that comparisons inside the
        # integrand can be decided; tape.Variants records at several theta
        lo = None if isinstance(lower, _Inf) else _value_of(lower)
        hi = None if isinstance(upper, _Inf) else _value_of(upper)
        th = min(max(r.theta, 1e-9), 1.0 - 1e-9)
        if lo is not None and hi is not None:
            tval = lo + th * (hi - lo)
        elif lo is not None and isinstance(upper, _Inf):
            tval = (lo - 1.0 + 1.0 / (1.0 - th)) if upper.sign > 0 else (lo + 1.0 - 1.0 / (1.0 - th))
        elif hi is not None and isinstance(lower, _Inf):
            tval = (hi + 1.0 - 1.0 / (1.0 - th)) if lower.sign < 0 else (hi - 1.0 + 1.0 / (1.0 - th))
        elif lo is None and hi is None and isinstance(lower, _Inf) and isinstance(upper, _Inf):
            import math
            tval = math.tan(math.pi * (th - 0.5))
    xi = advar._from_node(r.emit(T.IVAR), tval)
    ip = [advar._from_node(r.emit(T.IPARAM, k), _value_of(pars[k]) if r.concrete else None) for k in range(len(pars))]
    y = f(xi, ip)
    if not isinstance(y, advar):
        y = advar(y)
    sub = r.close(y._n())
    r.depth -= 1
    ipar_off = len(r.tape.ipar_nodes)
    r.tape.ipar_nodes.extend(binds)
    d.update(integrand=sub, n_ipars=len(pars), ipar_off=ipar_off, depth=r.depth + 1,
             rel_error=-1.0 if rel_error is None else float(rel_error),
             abs_error=-1.0 if abs_error is None else float(abs_error))
    r.tape.integrals.append(d)
    return advar._from_node(r.emit(T.INTEGRATE, len(r.tape.integrals) - 1, -1, 0))


def _value_of(v):
    """value at the recording point of something that may be a number or a traced value (None when unknown)"""
    if isinstance(v, _Sym):
        return v.val
    if _is_num(v):
        return float(v)
    return None


def aux(k):
    """The k-th auxiliary per-point real input (GFH_AUX): a real(kp) function of x that the host tabulates
    (Context.set_aux).  This is how a Fortran eval() that does plain real arithmetic on x reaches the device;
    in Python the same arithmetic on the symbolic x is recorded directly, so aux() is for tests."""
    r = _need_rec()
    r.tape.n_aux = max(r.tape.n_aux, int(k) + 1)
    a = Real(r.emit(T.AUX, int(k), -1, T.F_REAL))
    a.leaf = ('aux', int(k))
    return a


def trace_model(fn, n_pars, x=None, pars=None, script=None, theta=0.5):
    """Record ``fn(pars, x)`` (pars: list of advar, x: Real) into a Tape.

    Without ``x`` the recording is symbolic: ``fn`` must be straight-line code.  With ``x`` (and ``pars``, the parameter values)
    it is a recording AT that point: values travel with the nodes, every comparison of AD variables is decided from them --
    the first ``len(script)`` ones take the outcomes of ``script`` instead -- and is recorded as a guard node."""
    global _rec
    if _rec is not None:
        raise RuntimeError('nested model tracing')
    concrete = x is not None
    if concrete and (pars is None or len(pars) != n_pars):
        raise ValueError('a recording at a point needs the n_pars parameter values')
    _rec = _Rec(n_pars, concrete=concrete, script=script, theta=theta)
    try:
        _rec.open()
        ps = [advar._from_node(_rec.emit(T.PARAM, k), float(pars[k]) if concrete else None) for k in range(n_pars)]
        xr = Real(_rec.emit(T.X, flags=T.F_REAL), float(x) if concrete else None)
        xr.leaf = ('x',)
        y = fn(ps, xr)
        if isinstance(y, advar):
            res = y._n()
        else:
            res = _real_node(y)
        _rec.close(res)
        # sub[0] must be eval(); integrands were appended after it was opened
        return _rec.tape
    finally:
        _rec = None
