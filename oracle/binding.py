"""ctypes binding of the CPU oracle (oracle/libgadfit_oracle.so).  TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NONE, SQRT_Y, PROPTO_Y, INVERSE_Y, USER = 0, 1, 2, 3, 4


def build():
    subprocess.check_call(['make', '-s', '-C', HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, 'libgadfit_oracle.so')
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_last_error.restype = C.c_char_p
    return _LIB


class FitOptions(C.Structure):
    _fields_ = ([(n, C.c_double) for n in ('lambda_', 'lam_up', 'lam_down', 'accth', 'grad_chi2', 'cos_phi',
                                            'rel_error', 'rel_error_global', 'chi2_rel', 'chi2_abs')] +
                [('has_' + n, C.c_int) for n in ('lambda', 'lam_up', 'lam_down', 'accth', 'grad_chi2', 'cos_phi',
                                                 'rel_error', 'rel_error_global', 'chi2_rel', 'chi2_abs')] +
                [('DTD_min', C.POINTER(C.c_double)),
                 ('lam_incs', C.c_int), ('has_lam_incs', C.c_int),
                 ('uphill', C.c_int), ('has_uphill', C.c_int),
                 ('max_iter', C.c_int), ('has_max_iter', C.c_int),
                 ('damp_max', C.c_int), ('has_damp_max', C.c_int),
                 ('nielsen', C.c_int), ('has_nielsen', C.c_int),
                 ('umnigh', C.c_int), ('has_umnigh', C.c_int),
                 ('n_images', C.c_int), ('umnigh_a', C.c_double)])


class Problem(C.Structure):
    _fields_ = [('tape', C.c_void_p), ('n_datasets', C.c_int), ('data_positions', C.POINTER(C.c_int64)),
                ('x', C.POINTER(C.c_double)), ('y', C.POINTER(C.c_double)), ('w', C.POINTER(C.c_double)),
                ('n_pars', C.c_int), ('pars', C.POINTER(C.c_double)), ('n_active', C.c_int),
                ('active_pars', C.POINTER(C.c_int32)), ('is_global', C.POINTER(C.c_int32)), ('loss', C.c_int),
                ('aux', C.POINTER(C.c_double)), ('finite_diff', C.c_int),
                ('n_variants', C.c_int), ('variants', C.c_void_p), ('hint', C.POINTER(C.c_double))]


class FitResult(C.Structure):
    _fields_ = [('iterations', C.c_int), ('dim', C.c_int), ('lambda_', C.c_double), ('chi2', C.c_double),
                ('dof', C.c_int), ('exit_reason', C.c_int), ('n_sweeps', C.c_int), ('n_chi2', C.c_int),
                ('n_omega', C.c_int), ('JTJ0', C.POINTER(C.c_double)), ('JTres0', C.POINTER(C.c_double)),
                ('delta1_0', C.POINTER(C.c_double)), ('delta2_0', C.POINTER(C.c_double)), ('chi2_0', C.c_double)]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _chk(rc):
    if rc != 0:
        raise RuntimeError('oracle: ' + lib().orc_last_error().decode())


def init_weights(error_type, y, sigma=None):
    y = np.ascontiguousarray(y, dtype=np.float64)
    w = np.empty_like(y)
    s = np.ascontiguousarray(sigma if sigma is not None else y, dtype=np.float64)
    lib().orc_init_weights(error_type, C.c_int64(y.size), _dp(y), _dp(s), _dp(w))
    return w


def eval_reverse(tape, x, pars, active):
    pars = np.ascontiguousarray(pars, dtype=np.float64)
    act = np.ascontiguousarray(active, dtype=np.int32)
    val = C.c_double()
    grad = np.zeros(max(1, int(act.astype(bool).sum())))
    _chk(lib().orc_eval_reverse(C.byref(tape.c), C.c_double(x), _dp(pars), _ip(act), C.byref(val), _dp(grad)))
    return val.value, grad


def eval_forward(tape, x, pars, active, d_seed, dd_seed=None):
    pars = np.ascontiguousarray(pars, dtype=np.float64)
    act = np.ascontiguousarray(active, dtype=np.int32)
    d = np.ascontiguousarray(d_seed, dtype=np.float64)
    dd = np.ascontiguousarray(dd_seed if dd_seed is not None else np.zeros_like(d), dtype=np.float64)
    out = np.zeros(3)
    _chk(lib().orc_eval_forward(C.byref(tape.c), C.c_double(x), _dp(pars), _ip(act), _dp(d), _dp(dd), _dp(out)))
    return out


class OracleProblem:
    """Holds the arrays of one fitting problem for the oracle."""

    def __init__(self, tape, x_list, y_list, w_list, pars, active_pars, is_global, loss=0, aux=None, use_ad=True, hint=None):
        """tape: a Tape, or a gadfit_amd.tape.Variants (branching eval(): the oracle takes, per point, the recorded path whose
        comparisons hold there); hint: per point, the variant to prefer among those (paths that differ without a comparison)"""
        self.variants = None
        if hasattr(tape, 'tapes'):
            self.variants = tape
            tape = tape.tapes[0]
        self.tape = tape
        self.nd = len(x_list)
        self.dp = np.zeros(self.nd + 1, dtype=np.int64)
        for i, xs in enumerate(x_list):
            self.dp[i + 1] = self.dp[i] + len(xs)
        self.x = np.ascontiguousarray(np.concatenate(x_list), dtype=np.float64)
        self.y = np.ascontiguousarray(np.concatenate(y_list), dtype=np.float64)
        self.w = np.ascontiguousarray(np.concatenate(w_list), dtype=np.float64)
        self.pars = np.ascontiguousarray(pars, dtype=np.float64).reshape(self.nd, tape.n_pars).copy()
        self.active = np.ascontiguousarray(active_pars, dtype=np.int32)
        self.is_global = np.ascontiguousarray(is_global, dtype=np.int32)
        self.c = Problem(C.cast(C.pointer(tape.c), C.c_void_p), self.nd,
                         self.dp.ctypes.data_as(C.POINTER(C.c_int64)), _dp(self.x), _dp(self.y), _dp(self.w),
                         tape.n_pars, _dp(self.pars), self.active.size, _ip(self.active), _ip(self.is_global), int(loss), None, 0 if use_ad else 1)
        if self.variants is not None:
            n, arr = self.variants.c_array
            self._varr = arr
            self.c.n_variants = n
            self.c.variants = C.cast(arr, C.c_void_p)
        if hint is not None:
            self.hint = np.ascontiguousarray(hint, dtype=np.float64)
            assert self.hint.size == self.x.size
            self.c.hint = _dp(self.hint)
        if aux is not None:      # [n_aux][N] auxiliary per-point columns
            self.aux = np.ascontiguousarray(np.atleast_2d(np.asarray(aux, dtype=np.float64)))
            assert self.aux.shape[1] == self.x.size and self.aux.shape[0] >= tape.n_aux
            self.c.aux = _dp(self.aux)
        jac = np.zeros((self.nd, self.active.size), dtype=np.int32)
        self.dim = lib().orc_jacobian_indices(self.nd, self.active.size, _ip(self.active), _ip(self.is_global), _ip(jac))
        self.jac = jac

    @property
    def N(self):
        return int(self.dp[-1])

    def sweep(self, n_images=1, want_J=False):
        dim = self.dim
        JTJ = np.zeros((dim, dim)); JTr = np.zeros(dim); res = np.zeros(self.N)
        JT = np.zeros((self.N, dim)) if want_J else None
        _chk(lib().orc_sweep(C.byref(self.c), n_images, _dp(JTJ), _dp(JTr), _dp(res), _dp(JT) if want_J else None))
        return JTJ, JTr, res, JT

    def chi2(self, n_images=1):
        v = C.c_double(); res = np.zeros(self.N)
        _chk(lib().orc_chi2(C.byref(self.c), n_images, C.byref(v), _dp(res)))
        return v.value, res

    def omega(self, delta1, JT):
        om = np.zeros(self.N); jto = np.zeros(self.dim)
        d1 = np.ascontiguousarray(delta1, dtype=np.float64)
        _chk(lib().orc_omega(C.byref(self.c), _dp(d1), _dp(JT), _dp(om), _dp(jto)))
        return om, jto

    def fit(self, n_images=1, DTD_min=None, umnigh_a=0.5, **kw):
        o = FitOptions()
        for k, v in kw.items():
            if v is None:
                continue
            name = 'lambda' if k in ('lambda_', 'lam', 'lambda') else k
            setattr(o, 'lambda_' if name == 'lambda' else name, v)
            setattr(o, 'has_' + name, 1)
        o.n_images = n_images
        o.umnigh_a = umnigh_a
        if DTD_min is not None:
            self._dtd = np.ascontiguousarray(DTD_min, dtype=np.float64)
            o.DTD_min = _dp(self._dtd)
        r = FitResult()
        dim = self.dim
        self.JTJ0 = np.zeros((dim, dim)); self.JTres0 = np.zeros(dim)
        self.delta1_0 = np.zeros(dim); self.delta2_0 = np.zeros(dim)
        r.JTJ0 = _dp(self.JTJ0); r.JTres0 = _dp(self.JTres0); r.delta1_0 = _dp(self.delta1_0); r.delta2_0 = _dp(self.delta2_0)
        _chk(lib().orc_fit(C.byref(self.c), C.byref(o), C.byref(r)))
        self.umnigh_a = o.umnigh_a
        return r


def quad_counters():
    """integrand evaluations of the bisections / of the final passes, integrate() calls, final intervals since the last call"""
    out = (C.c_longlong * 4)()
    lib().orc_quad_counters(out)
    return dict(evals_bisect=int(out[0]), evals_final=int(out[1]), calls=int(out[2]), intervals=int(out[3]))


def potr(a, b):
    a = np.asfortranarray(a, dtype=np.float64).copy(order='F')
    b = np.ascontiguousarray(b, dtype=np.float64).copy()
    _chk(lib().orc_potr(a.shape[0], _dp(a), _dp(b)))
    return b
