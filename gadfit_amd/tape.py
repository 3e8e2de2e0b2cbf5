"""Model tape (include/gadfit_tape.h) as ctypes structures + a small builder.

The tape is what crosses the C ABI in ``gfh_set_model``: an SSA restatement of the user's
``fitfunc.eval`` written with advar operators (reference:
fortran/gadfit/fitfunction.F90:59-63, automatic_differentiation.F90:82-229).
"""
import ctypes as C

# enum gfh_op
CONST, X, PARAM, LIFT, NEG, IVAR, IPARAM, AUX = 0, 1, 2, 3, 4, 5, 6, 7
ADD, SUB, MUL, DIV, POW, POWI = 10, 11, 12, 13, 14, 15
ABS, EXP, SQRT, LOG, SIN, COS, TAN, ASIN, ACOS, ATAN = range(20, 30)
SINH, COSH, TANH, ASINH, ACOSH, ATANH, ERF = range(30, 37)
INTEGRATE = 40
F_REAL = 1

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
                ('rel_error_outer', C.c_double), ('rel_error_inner', C.c_double)]


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
        self._c = None

    def set_integration(self, rel_error=None, rel_error_inner=None, rule=None, dbl=False):
        """gadf_init's integration arguments (gadfit.F90:166-172; NI:114-135)."""
        eps = 2.220446049250313e-16
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
                     self.gk_points, self.n_aux, self.rel_error_outer, self.rel_error_inner)
        self._keep = (keep, subs, ints, ip)
        self._c = t

    def n_ops(self):
        return sum(len(n) for n, _ in self.subtapes)
