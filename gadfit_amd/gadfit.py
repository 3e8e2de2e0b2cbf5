"""Procedural driver API -- mirror of the reference's ``module gadfit``
(fortran/gadfit/gadfit.F90:41-58): gadf_init, gadf_add_dataset, gadf_set, gadf_set_errors,
gadf_set_verbosity, gadf_fit, gadf_print, gadf_close and the readable ``fitfuncs``.

Same names, argument meaning (1-based dataset / parameter indices, parameter names) and
error behaviour (errors raise ``GadfitError`` with the reference's message instead of
``error stop``).  All state is module-global like the reference: one fit at a time per
process.  The hot path (STEP 1-3 and chi2 of gadf_fit) runs on the GPU through
libgadfit_hip.so; there is no CPU fallback.
"""
import copy
import os

import numpy as np

from . import _lib
from .fitfunction import fitfunc

NONE, SQRT_Y, PROPTO_Y, INVERSE_Y, USER = 0, 1, 2, 3, 4          # gadfit.F90:45-48
GLOBAL, LOCAL, GLOBAL_AND_LOCAL = 0, 1, 2
GAUSS_KRONROD_15P, GAUSS_KRONROD_21P, GAUSS_KRONROD_31P = 15, 21, 31
GAUSS_KRONROD_41P, GAUSS_KRONROD_51P, GAUSS_KRONROD_61P = 41, 51, 61


class GadfitError(RuntimeError):
    pass


class _State:
    def __init__(self):
        self.reset()

    def reset(self):
        self.fitfuncs = None
        self.active = None        # per parameter: bool
        self.is_global = None
        self.datasets = []        # (x, y, weights-or-None)
        self.n_datasets = 0
        self.error_type = NONE
        self.ctx = None
        self.tape = None
        self.uploaded = False
        self.set_count = 0
        self.verbosity = 0
        self.iterations = 0
        self.last_result = None
        self.umnigh_a = 0.5
        self.integration = {}
        self.device = 0
        self.comm = None
        self.loss = 0
        self.lb_on = False


_S = _State()
fitfuncs = None   # rebound by gadf_init (gadfit.F90:64)


def _need_init():
    if _S.fitfuncs is None:
        raise GadfitError('Number of datasets is undetermined. Call gadf_init first.')


def gadf_init(f, num_datasets=1, sweep_size=None, trace_size=None, const_size=None, ws_size=None,
              ws_size_inner=None, integration_rule=None, ad_memory=None, rel_error_inner=None, rel_error=None,
              device=None, comm=None):
    """gadfit.F90:133-184.  The AD tape-size arguments are accepted and ignored: the tape is
    recorded once per model, not per point; the quadrature workspace sizes are honoured on the device.  ``device``: HIP device index (default
    LOCAL_RANK or 0); ``comm``: (nranks, rank, unique_id) to shard over several GPUs."""
    global fitfuncs
    if not isinstance(f, fitfunc):
        raise GadfitError('f must extend fitfunc')
    _S.reset()
    _S.fitfuncs = []
    for _ in range(num_datasets):
        g = copy.copy(f)
        g.init()
        _S.fitfuncs.append(g)
    fitfuncs = _S.fitfuncs
    n = len(_S.fitfuncs[0].pars)
    _S.active = [False] * n
    _S.is_global = [False] * n
    _S.n_datasets = num_datasets
    _S.integration = dict(rel_error=rel_error, rel_error_inner=rel_error_inner, rule=integration_rule,
                          dbl=(rel_error_inner is not None or ws_size_inner is not None),
                          ws_size=ws_size, ws_size_inner=ws_size_inner)        # workspace sizes travel with the tape (NI:114-135)
    _S.device = int(os.environ.get('LOCAL_RANK', '0')) if device is None else device
    _S.comm = comm


def gadf_add_dataset(*args):
    """gadf_add_dataset(path) | gadf_add_dataset(x_data, y_data [, weights]) (gadfit.F90:189-246)."""
    _need_init()
    if len(_S.datasets) >= _S.n_datasets:
        raise GadfitError('Too many calls to gadf_add_dataset (%d/%d).' % (len(_S.datasets) + 1, _S.n_datasets))
    if len(args) == 1 and isinstance(args[0], str):
        # read_data (gadfit.F90:212-215, 422-437) through the library's reader (reader.cpp): records that do not begin with a number
        # are skipped; three columns if every record has them (the third is used under USER errors), else two
        if not os.path.exists(args[0]):
            raise GadfitError('Cannot open ' + args[0])
        try:
            x, y, w = _lib.read_columns(args[0], 3)
        except _lib.GadfitHipError:
            try:
                x, y = _lib.read_columns(args[0], 2); w = None
            except _lib.GadfitHipError as e:
                raise GadfitError(str(e))
        if x.size == 0:
            raise GadfitError(args[0] + ' contains no valid data points.')
        _S.datasets.append((x, y, w))
    else:
        x = np.asarray(args[0], dtype=np.float64); y = np.asarray(args[1], dtype=np.float64)
        w = np.asarray(args[2], dtype=np.float64) if len(args) > 2 and args[2] is not None else None
        _S.datasets.append((x, y, w))
    _S.uploaded = False


def gadf_set(*args):
    """gadf_set(dataset_i, par, val [, active]) (local) | gadf_set(par, val [, active]) (global);
    par is a 1-based index or a name (8 specifics of gadfit.F90:54-58, 255-342)."""
    _need_init()
    a = list(args)
    active = False
    if isinstance(a[-1], (bool, np.bool_)):
        active = bool(a.pop())
    if len(a) == 3:
        ds, par, val = a
        if ds > len(_S.fitfuncs):
            raise GadfitError('Invalid dataset index. Call gadf_init with the correct number of datasets.')
        i = _S.fitfuncs[0].get_index(par) if isinstance(par, str) else int(par)
        _S.is_global[i - 1] = False
        _S.fitfuncs[ds - 1].set(i, float(val))
        _S.active[i - 1] = active
        _S.set_count += 1
    elif len(a) == 2:
        par, val = a
        i = _S.fitfuncs[0].get_index(par) if isinstance(par, str) else int(par)
        for g in _S.fitfuncs:
            g.set(i, float(val))
            _S.set_count += 1
        _S.active[i - 1] = active
        _S.is_global[i - 1] = True
    else:
        raise GadfitError('gadf_set: wrong number of arguments')


def gadf_set_errors(e):
    _need_init()
    _S.error_type = e     # gadfit.F90:392-395


LOSS_LINEAR, LOSS_CAUCHY, LOSS_HUBER = 0, 1, 2


def gadf_set_loss(loss):
    """Robust cost function -- the C++ solver's `settings.loss` (c++/gadfit/lm_solver.h:76-83, 208); the
    Fortran reference has no such switch.  Applies to the fits that follow."""
    _need_init()
    if loss not in (LOSS_LINEAR, LOSS_CAUCHY, LOSS_HUBER):
        raise GadfitError('gadf_set_loss: unknown loss function')
    _S.loss = loss


def gadf_set_verbosity(scope=None, digits=None, timings=None, memory=None, workloads=None, delta1=None, delta2=None,
                       cos_phi=None, grad_chi2=None, uphill=None, acc=None, output=None):
    """gadfit.F90:356-385; only on/off of the per-iteration log is honoured."""
    _S.verbosity = 0 if output in ('/dev/null', os.devnull) else 1


RECORD_SAMPLE = 2048       # abscissas per dataset at which a branching eval() is recorded before the fit


def _ensure_device():
    if len(_S.datasets) != _S.n_datasets:        # read_data, gadfit.F90:403-405 (checked before touching the device)
        raise GadfitError('Some datasets are missing. gadf_add_dataset must be called %d times.' % _S.n_datasets)
    if _S.ctx is None:
        _S.ctx = _lib.Context(_S.device)
        if 'GADFIT_HIP_KEEP_J' not in os.environ:
            # the Jacobian has no reader behind this API (private in the reference): written only for the fits whose
            # options read it back; same J^T J / J^T r / chi2 bit for bit (gfh_set_keep_jacobian mode 2)
            _S.ctx.set_keep_jacobian(2)
        if _S.comm is not None:
            _S.ctx.comm_init(*_S.comm)
    if _S.tape is not None and not _tape_is_current():
        _S.tape = None                    # eval() has changed since it was recorded (a global it reads): record it again
    if _S.tape is None:
        ig = _S.integration
        configure = None
        if ig.get('dbl') or ig.get('rel_error') is not None or ig.get('rule') is not None or ig.get('ws_size') is not None:
            def configure(t):
                t.set_integration(rel_error=ig['rel_error'], rel_error_inner=ig['rel_error_inner'], rule=ig['rule'],
                                  dbl=ig['dbl'], ws_size=ig.get('ws_size'), ws_size_inner=ig.get('ws_size_inner'))
        try:
            _S.tape = _S.fitfuncs[0].trace()
            if configure is not None:
                configure(_S.tape)
        except TypeError as e:
            if 'eval() branches' not in str(e) and 'comparison inside an integrand' not in str(e):
                raise
            # eval() compares AD variables: one tape per path (gfh_set_model_variants).  Recorded over the data at the current
            # parameter values -- first, last and up to RECORD_SAMPLE evenly spaced abscissas of every dataset; a path the sample
            # misses, or one that only appears once the parameters have moved, is reported by the device during the fit and
            # recorded then (the handler _lib.Context.set_model installs)
            _S.tape = _S.fitfuncs[0].trace_variants(configure)
            for d, f in enumerate(_S.fitfuncs):
                x = np.asarray(_S.datasets[d][0], dtype=np.float64)
                if x.size == 0:
                    continue
                idx = np.unique(np.concatenate([[0, x.size - 1], np.linspace(0, x.size - 1, min(x.size, RECORD_SAMPLE)).astype(np.int64)]))
                _S.tape.explore(x[idx], [p.val for p in f.pars])
            if len(_S.tape) == 0:
                raise GadfitError('There are no data points.')
        _S.ctx.set_model(_S.tape)
    if not _S.uploaded:
        if len(_S.datasets) != _S.n_datasets:
            raise GadfitError('Some datasets are missing. gadf_add_dataset must be called %d times.' % _S.n_datasets)
        xs = np.concatenate([d[0] for d in _S.datasets]); ys = np.concatenate([d[1] for d in _S.datasets])
        if _S.error_type == USER:
            if any(d[2] is None for d in _S.datasets):
                raise GadfitError('USER errors requested but a dataset has no weights column')
            ws = np.concatenate([d[2] for d in _S.datasets])
        else:
            ws = np.ones_like(xs)
        pos = np.zeros(_S.n_datasets + 1, dtype=np.int64)
        for i, d in enumerate(_S.datasets):
            pos[i + 1] = pos[i] + len(d[0])
        _S.ctx.set_data(xs, ys, ws, pos)
        _S.ctx.init_weights(_S.error_type)     # init_weights on the device (gadfit.F90:445-470)
        _S.uploaded = True


def _f32(v):
    return None if v is None else float(np.float32(v))   # the first ten arguments are real(real32)


def _tape_is_current():
    """a later gadf_fit: is the recorded model still what eval() does?  (the reference calls eval() afresh at every point of every
    fit, gadfit.F90:679-690: a global the model reads and the program changed between two fits takes effect in the second)"""
    try:
        if hasattr(_S.tape, 'is_current'):
            return _S.tape.is_current([p.val for p in _S.fitfuncs[0].pars])
        return _S.fitfuncs[0].trace().signature()[0] == _S.tape.signature()[0]
    except TypeError:
        return False


def gadf_fit(lambda_=None, lam_up=None, lam_down=None, accth=None, grad_chi2=None, cos_phi=None, rel_error=None,
             rel_error_global=None, chi2_rel=None, chi2_abs=None, DTD_min=None, lam_incs=None, uphill=None,
             max_iter=None, damp_max=None, nielsen=None, umnigh=None, load_balancing=None, use_ad=None, **kw):
    """gadfit.F90:502-1035.  ``lambda`` is spelled ``lambda_`` (also accepted through **kw)."""
    _need_init()
    if 'lambda' in kw:
        lambda_ = kw.pop('lambda')
    if kw:
        raise GadfitError('gadf_fit: unknown arguments %s' % sorted(kw))
    active = [i for i, a in enumerate(_S.active) if a]
    if not active:
        raise GadfitError('There are no active parameters.')
    _ensure_device()
    _S.ctx.set_loss(_S.loss)
    if bool(load_balancing) != _S.lb_on:                # adaptive parallelism: the library copies the data when they are set
        _S.ctx.set_load_balancing(bool(load_balancing)); _S.lb_on = bool(load_balancing)
        if _S.lb_on:
            _S.uploaded = False
            _ensure_device()
    _S.ctx.set_use_ad(use_ad is None or bool(use_ad))      # gadfit.F90:583-584; False: fitfunction.F90:155-203 on the device
    pars = np.array([[p.val for p in g.pars] for g in _S.fitfuncs])
    out, r = _S.ctx.fit(pars, active, [int(g) for g in _S.is_global], DTD_min=DTD_min, verbosity=_S.verbosity,
                        umnigh_a=_S.umnigh_a,
                        lambda_=_f32(lambda_), lam_up=_f32(lam_up), lam_down=_f32(lam_down), accth=_f32(accth),
                        grad_chi2=_f32(grad_chi2), cos_phi=_f32(cos_phi), rel_error=_f32(rel_error),
                        rel_error_global=_f32(rel_error_global), chi2_rel=_f32(chi2_rel), chi2_abs=_f32(chi2_abs),
                        lam_incs=lam_incs, uphill=uphill, max_iter=max_iter,
                        damp_max=None if damp_max is None else int(damp_max),
                        nielsen=None if nielsen is None else int(nielsen),
                        umnigh=None if umnigh is None else int(umnigh))
    _S.umnigh_a = _S.ctx.umnigh_a
    for g, row in zip(_S.fitfuncs, out):
        for p, v in zip(g.pars, row):
            p.val = float(v)
    _S.iterations = r.iterations
    _S.last_result = r
    return r


def gadf_print(begin=None, end=None, points=None, output=None, grouped=None, logplot=None):
    """gadfit.F90:1255-1395 writes curve / parameter files; here only the parameter table."""
    _need_init()
    lines = []
    for i, g in enumerate(_S.fitfuncs):
        for j, p in enumerate(g.pars):
            lines.append('%d %s %.17g' % (i + 1, p.name or ('par%d' % (j + 1)), p.val))
    text = '\n'.join(lines) + '\n'
    if output:
        with open(output + '_parameters', 'w') as fh:
            fh.write(text)
    return text


def gadf_close():
    """gadfit.F90:1399-1412: frees everything."""
    global fitfuncs
    if _S.ctx is not None:
        _S.ctx.close()
    _S.reset()
    fitfuncs = None


def context():
    """The device context of the current fit (timers, read-back); creates it if needed."""
    _need_init()
    _ensure_device()
    return _S.ctx
