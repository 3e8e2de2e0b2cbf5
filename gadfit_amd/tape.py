"""Model tape (include/gadfit_tape.h) as ctypes structures + a small builder.

The tape is what crosses the C ABI in ``gfh_set_model``: an SSA restatement of the user's
``fitfunc.eval`` written with advar operators (reference:
fortran/gadfit/fitfunction.F90:59-63, automatic_differentiation.F90:82-229).
"""
import ctypes as C

# enum gfh_op
CONST, X, PARAM, LIFT, NEG, IVAR, IPARAM, AUX, VAL = 0, 1, 2, 3, 4, 5, 6, 7, 8
ADD, SUB, MUL, DIV, POW, POWI = 10, 11, 12, 13, 14, 15
ABS, EXP, SQRT, LOG, SIN, COS, TAN, ASIN, ACOS, ATAN = range(20, 30)
SINH, COSH, TANH, ASINH, ACOSH, ATANH, ERF = range(30, 37)
INTEGRATE = 40
GUARD_GT, GUARD_LT = 50, 51     # comparisons of values (AD:315-395) met while recording: a, b operands, F_TAKEN = outcome
F_REAL = 1
F_TAKEN = 2

UNARY_NAMES = {ABS: 'abs', EXP: 'exp', SQRT: 'sqrt', LOG: 'log', SIN: 'sin', COS: 'cos',
               TAN: 'tan', ASIN: 'asin', ACOS: 'acos', ATAN: 'atan', SINH: 'sinh',
               COSH: 'cosh', TANH: 'tanh', ASINH: 'asinh', ACOSH: 'acosh', ATANH: 'atanh',
               ERF: 'erf'}


class gfh_node(C.Structure):
    _fields_ = [('op', C.c_int32), ('a', C.c_int32), ('b', C.c_int32), ('flags', C.c_int32),
                ('c', C.c_double)]


class gfh_subtape(C.Structure):
    _fields_ = [('n_nodes', C.c_int32), ('result', C.c_int32), ('nodes', C.POINTER(gfh_node))]


class gfh_integral(C.Structure):
    _fields_ = [('integrand', C.c_int32), ('lower', C.c_int32), ('upper', C.c_int32),
                ('lower_inf', C.c_int32), ('upper_inf', C.c_int32), ('n_ipars', C.c_int32),
                ('ipar_off', C.c_int32), ('depth', C.c_int32),
                ('rel_error', C.c_double), ('abs_error', C.c_double)]


class gfh_tape(C.Structure):
    _fields_ = [('n_pars', C.c_int32), ('n_subtapes', C.c_int32),
                ('sub', C.POINTER(gfh_subtape)),
                ('n_integrals', C.c_int32), ('integrals', C.POINTER(gfh_integral)),
                ('ipar_nodes', C.POINTER(C.c_int32)),
                ('gk_points', C.c_int32), ('n_aux', C.c_int32),
                ('rel_error_outer', C.c_double), ('rel_error_inner', C.c_double),
                ('ws_size', C.c_int32), ('ws_size_inner', C.c_int32)]


class Tape:
    """Python-side owner of a tape; ``.c`` is the ctypes ``gfh_tape`` (kept alive by self)."""

    def __init__(self, n_pars):
        self.n_pars = n_pars
        self.subtapes = []      # list of (nodes:list[tuple(op,a,b,flags,c)], result)
        self.integrals = []     # list of dict
        self.ipar_nodes = []
        self.gk_points = 0
        self.n_aux = 0          # auxiliary per-point columns (GFH_AUX nodes)
        # numerical_integration.F90:61-62 defaults; init_integration without an inner
        # workspace sets outer := inner (NI:117-119)
        eps = 2.220446049250313e-16
        self.rel_error_inner = 1e2 * eps
        self.rel_error_outer = 1e2 * eps
        self.ws_size = 0          # 0 = the reference's default of 1000 intervals (NI:40)
        self.ws_size_inner = 0
        self._c = None

    def set_integration(self, rel_error=None, rel_error_inner=None, rule=None, dbl=False, ws_size=None, ws_size_inner=None):
        """gadf_init's integration arguments (gadfit.F90:166-172; NI:114-135)."""
        eps = 2.220446049250313e-16
        if ws_size is not None:
            self.ws_size = int(ws_size)
        if ws_size_inner is not None:
            self.ws_size_inner = int(ws_size_inner)
        if dbl:
            if rel_error_inner is not None:
                self.rel_error_inner = float(rel_error_inner)
            # ws(2) allocated => outer default stays 1e3*eps unless given
            self.rel_error_outer = 1e3 * eps if rel_error is None else float(rel_error)
        else:
            self.rel_error_outer = self.rel_error_inner if rel_error is None else float(rel_error)
        if rule is not None:
            self.gk_points = int(rule)
        self._c = None

    @property
    def c(self):
        if self._c is None:
            self._build()
        return self._c

    def _build(self):
        keep = []
        subs = (gfh_subtape * len(self.subtapes))()
        for i, (nodes, result) in enumerate(self.subtapes):
            arr = (gfh_node * max(1, len(nodes)))()
            for k, (op, a, b, fl, c) in enumerate(nodes):
                arr[k] = gfh_node(op, a, b, fl, c)
            keep.append(arr)
            subs[i] = gfh_subtape(len(nodes), result, arr)
        ints = (gfh_integral * max(1, len(self.integrals)))()
        for i, d in enumerate(self.integrals):
            ints[i] = gfh_integral(d['integrand'], d['lower'], d['upper'], d['lower_inf'],
                                   d['upper_inf'], d['n_ipars'], d['ipar_off'], d['depth'],
                                   d['rel_error'], d['abs_error'])
        ip = (C.c_int32 * max(1, len(self.ipar_nodes)))(*self.ipar_nodes)
        t = gfh_tape(self.n_pars, len(self.subtapes), subs, len(self.integrals), ints, ip,
                     self.gk_points, self.n_aux, self.rel_error_outer, self.rel_error_inner, self.ws_size, self.ws_size_inner)
        self._keep = (keep, subs, ints, ip)
        self._c = t

    def n_ops(self):
        return sum(len(n) for n, _ in self.subtapes)

    def signature(self):
        """everything that makes two recordings the same path through eval(): operations, operands, literal values, the
        outcomes of the comparisons, the integrate() call sites"""
        return (tuple((tuple(n), r) for n, r in self.subtapes),
                tuple(tuple(sorted(d.items())) for d in self.integrals), tuple(self.ipar_nodes))

    def has_integrand_guards(self):
        """does an integrand of this recording compare AD variables (its path then depends on the integration variable)?"""
        return any(op in (GUARD_GT, GUARD_LT) for nodes, _ in self.subtapes[1:] for (op, a, b, fl, c) in nodes)

    def guard_outcomes(self):
        """outcomes of the comparisons along eval(), in the order they were met"""
        return [bool(fl & F_TAKEN) for (op, a, b, fl, c) in self.subtapes[0][0] if op in (GUARD_GT, GUARD_LT)]


class Variants:
    """The recorded paths of ONE branching eval() (include/gadfit_tape.h, guard nodes; gfh_set_model_variants).

    ``fn(pars, x)`` is recorded at concrete points: the comparisons of AD variables it makes (``>``, ``<``, also through
    ``max`` / ``min``) are decided from the values there and become guard nodes; recordings that took the same path are one
    variant.  ``add_point`` is what the library's unseen-branch handler calls during a fit."""

    def __init__(self, fn, n_pars, configure=None):
        self.fn = fn
        self.n_pars = n_pars
        self.configure = configure        # called on every new Tape (e.g. lambda t: t.set_integration(...))
        self.tapes = []
        self._index = {}
        self._origin = []                 # per variant: (x, theta) of its first recording (is_current)
        self._c = None

    # where the integration variable sits in its range when an integrand that compares AD variables is recorded (besides 0.5)
    THETAS = (0.003, 0.03, 0.1, 0.2, 0.3, 0.4, 0.6, 0.7, 0.8, 0.9, 0.97, 0.997)

    def add_point(self, x, pars, script=None):
        """records fn at (x, pars); the first len(script) comparisons of eval() are forced to the given outcomes.  Returns the
        variant's index.  An integrand that compares AD variables is recorded with its integration variable at several places of
        its range (every distinct path through it is a tape of its own; the library pools them into one call site)."""
        from . import ad
        first = None
        for theta in (0.5,) + self.THETAS:
            t = ad.trace_model(self.fn, self.n_pars, x=float(x), pars=[float(v) for v in pars], script=script, theta=theta)
            if self.configure is not None:
                self.configure(t)
            key = t.signature()
            if key not in self._index:
                if first is None and t.integrals and not t.has_integrand_guards():
                    # a number formed from the VALUE of an integration variable (reading .val inside an integrand) would be a
                    # literal of the recording, frozen where the variable sat: recorded elsewhere the path must be the same
                    t2 = ad.trace_model(self.fn, self.n_pars, x=float(x), pars=[float(v) for v in pars], script=script, theta=0.3819660112501051)
                    if self.configure is not None:
                        self.configure(t2)
                    if t2.signature() != key:
                        raise TypeError('an integrand forms a number from the value of its integration variable (.val): such literals '
                                        'cannot follow the abscissas of the quadrature on the device; keep them as advar')
                self._index[key] = len(self.tapes)
                self.tapes.append(t)
                self._origin.append((float(x), theta))
                self._c = None
            if first is None:
                first = self._index[key]
                if not t.has_integrand_guards():
                    break
        return first

    def is_current(self, pars):
        """Does fn still do what the recordings hold?  (a later fit: the reference calls eval() afresh at every point of every
        fit, so a global that fn reads and the program changed between two fits takes effect there.)  Every variant is recorded
        again where it was first met, its comparisons forced, and must come out the same, literal for literal."""
        from . import ad
        for t, (x, theta) in zip(self.tapes, self._origin):
            t2 = ad.trace_model(self.fn, self.n_pars, x=x, pars=[float(v) for v in pars], script=t.guard_outcomes(), theta=theta)
            if self.configure is not None:
                self.configure(t2)
            if t2.signature() != t.signature() and not t.has_integrand_guards():
                return False
        return True

    def explore(self, xs, pars):
        """records fn at every abscissa of xs with one parameter set; returns the variant index per point"""
        return [self.add_point(x, pars) for x in xs]

    def __len__(self):
        return len(self.tapes)

    @property
    def n_aux(self):
        return max([t.n_aux for t in self.tapes] + [0])

    @property
    def c_array(self):
        """(count, array of pointers to gfh_tape) for gfh_set_model_variants / the oracle"""
        if self._c is None:
            arr = (C.POINTER(gfh_tape) * max(1, len(self.tapes)))()
            for i, t in enumerate(self.tapes):
                arr[i] = C.pointer(t.c)
            self._c = arr
        return len(self.tapes), self._c
