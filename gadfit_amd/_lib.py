"""ctypes binding of libgadfit_hip.so (include/gadfit_hip.h).  No torch types, no fallback:
if the HIP library is missing or no GPU is present the calls raise."""
import ctypes as C
import os

import numpy as np

from . import tape as T

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'lib', 'libgadfit_hip.so')
_LIB = None


class GadfitHipError(RuntimeError):
    pass


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
                 ('verbosity', C.c_int), ('umnigh_a', C.c_double)])


class FitResult(C.Structure):
    _fields_ = [('iterations', C.c_int), ('dim', C.c_int), ('dof', C.c_int), ('exit_reason', C.c_int),
                ('lambda_', C.c_double), ('chi2', C.c_double), ('n_sweeps', C.c_int), ('n_chi2', C.c_int),
                ('n_omega', C.c_int), ('n_lookahead', C.c_int), ('seconds', C.c_double)]


# every symbol include/gadfit_hip.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _dp, _ip = C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_int32)
SYMBOLS = {
    'gfh_create': (_i, [_i, C.POINTER(_vp)]),
    'gfh_create_begin': (_i, [_i, C.POINTER(_vp)]),
    'gfh_create_group': (_i, [_i, _ip, C.POINTER(_vp)]),
    'gfh_group_size': (_i, [_vp]),
    'gfh_debug_group_allreduce': (_i, [_vp, _dp, _i, _ip, _i]),
    'gfh_debug_group_latency': (_i, [_vp, _i, _i, _dp]),
    'gfh_debug_allreduce_latency': (_i, [_vp, _i, _i, _dp]),
    'gfh_destroy': (None, [_vp]),
    'gfh_last_error': (C.c_char_p, [_vp]),
    'gfh_version': (_i, []),
    'gfh_comm_unique_id': (_i, [_vp]),
    'gfh_comm_init': (_i, [_vp, _i, _i, _vp]),
    'gfh_comm_init_from_env': (_i, [_vp]),
    'gfh_debug_set_rank': (_i, [_vp, _i, _i]),
    'gfh_comm_info': (_i, [_vp, C.POINTER(_i), C.POINTER(_i64)]),
    'gfh_debug_packed_layout': (_i, [_i, _i, _i64, _i, C.POINTER(_i64), _i, _ip, _i, _i, C.POINTER(_i64), _ip, _ip, _i]),
    'gfh_partition': (None, [_i64, _i, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    'gfh_gk_rule': (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'gfh_set_data': (_i, [_vp, _i64, _dp, _dp, _dp, _i, C.POINTER(_i64)]),
    'gfh_set_data_begin': (_i, [_vp, _i64, _dp, _dp, _dp, _i, C.POINTER(_i64)]),
    'gfh_queue_host_copy': (_i, [_vp, _vp, _vp, _i64]),
    'gfh_wait_host_copy': (_i, [_vp]),
    'gfh_read_columns': (_i, [C.c_char_p, _i, C.POINTER(_vp), C.POINTER(_i64)]),
    'gfh_take_columns': (_i, [_vp, _dp, _dp, _dp]),
    'gfh_free_columns': (None, [_vp]),
    'gfh_get_abscissas': (_i, [_vp, _dp]),
    'gfh_set_data_local': (_i, [_vp, _i64, _i, C.POINTER(_i64), _i64, _i64, _dp, _dp, _dp]),
    'gfh_init_weights': (_i, [_vp, _i]),
    'gfh_set_model': (_i, [_vp, C.POINTER(T.gfh_tape)]),
    'gfh_set_model_variants': (_i, [_vp, _i, C.POINTER(C.POINTER(T.gfh_tape)), _i]),
    'gfh_set_variant_hint_columns': (_i, [_vp, _i, _ip]),
    'gfh_model_needs_hint': (_i, [_vp]),
    'gfh_model_n_variants': (_i, [_vp]),
    'gfh_model_n_tapes': (_i, [_vp]),
    'gfh_set_unseen_handler': (_i, [_vp, _vp, _vp]),
    'gfh_set_pars_hook': (_i, [_vp, _vp, _vp]),
    'gfh_get_counters': (_i, [_vp, C.POINTER(_i64)]),
    'gfh_device_memory': (_i, [_vp, C.POINTER(_i64)]),
    'gfh_debug_mesh_stats': (_i, [_vp, C.POINTER(_i64)]),
    'gfh_model_source': (_i64, [_vp, _i, _ip, C.c_char_p, _i64]),
    'gfh_model_prepare': (_i, [_vp, _i, _ip]),
    'gfh_set_active': (_i, [_vp, _ip, _i, _ip, _i]),
    'gfh_sweep': (_i, [_vp, _dp, _ip, _i, _ip, _i, _dp, _dp, _dp]),
    'gfh_set_aux': (_i, [_vp, _i, _dp]),
    'gfh_set_aux_local': (_i, [_vp, _i, _dp]),
    'gfh_chi2': (_i, [_vp, _dp, _dp]),
    'gfh_omega': (_i, [_vp, _dp, _dp, _dp]),
    'gfh_aux': (_i, [_vp, _i, _dp, _dp]),
    'gfh_fit': (_i, [_vp, _dp, _i, _ip, _ip, C.POINTER(FitOptions), C.POINTER(FitResult)]),
    'gfh_set_lookahead': (_i, [_vp, _i]),
    'gfh_set_keep_jacobian': (_i, [_vp, _i]),
    'gfh_set_use_ad': (_i, [_vp, _i]),
    'gfh_set_fd_column_sets': (_i, [_vp, _i]),
    'gfh_set_load_balancing': (_i, [_vp, _i]),
    'gfh_repartition': (_i, [_vp, _dp]),
    'gfh_rebalance': (_i, [_vp, C.POINTER(_i)]),
    'gfh_group_ranges': (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    'gfh_set_loss': (_i, [_vp, _i]),
    'gfh_lm_iterate': (_i, [_vp, _dp, _i, _ip, _ip, _i, _dp, _dp]),
    'gfh_jacobian_indices': (_i, [_i, _i, _ip, _ip, _ip]),
    'gfh_potr': (_i, [_i, _dp, _dp]),
    'gfh_solve_damped': (_i, [_i, _i, _ip, _i, _dp, _dp, C.c_double, _dp, _dp, _i]),
    'gfh_get_timers': (_i, [_vp, _dp]),
    'gfh_reset_timers': (None, [_vp]),
    'gfh_set_timer_detail': (_i, [_vp, _i]),
    'gfh_set_placement_tries': (_i, [_vp, _i]),
    'gfh_set_placement_after': (_i, [_vp, _i]),
    'gfh_get_placement': (_i, [_vp, _dp]),
    'gfh_get_timer_spread': (_i, [_vp, _dp]),
    'gfh_launch_sweep': (_i, [_vp]),
    'gfh_launch_gram': (_i, [_vp]),
    'gfh_launch_chi2': (_i, [_vp]),
    'gfh_sync': (_i, [_vp]),
    'gfh_stream': (_vp, [_vp]),
    'gfh_time_kernel': (_i, [_vp, _i, _i, _dp]),
    'gfh_get_residuals': (_i, [_vp, _dp]),
    'gfh_get_jacobian': (_i, [_vp, _dp]),
    'gfh_get_omega': (_i, [_vp, _dp]),
    'gfh_get_points': (_i, [_vp, _i, C.POINTER(_i64), _dp, _dp]),
    'gfh_get_weights': (_i, [_vp, _dp]),
    'gfh_local_count': (_i64, [_vp]),
    'gfh_local_begin': (_i64, [_vp]),
}


def lib():
    """Loads libgadfit_hip.so; raises if it was not built (no fallback path exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GadfitHipError('libgadfit_hip.so is not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                                 'or gadfit_amd/build.py (there is no CPU fallback)')
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def dp(a):
    return a.ctypes.data_as(_dp)


def ip(a):
    return a.ctypes.data_as(_ip)


# gfh_unseen_handler (include/gadfit_hip.h)
UNSEEN_HANDLER = C.CFUNCTYPE(_i, _vp, _vp, _i, C.POINTER(_i64), _ip, _dp, C.POINTER(C.c_uint64), _ip, _dp)


class Context:
    """One GPU = one image.  device=-1 gives a compile-only context (usable without a GPU)."""

    def __init__(self, device=0, devices=None):
        """devices: a list of device indices (or 'all') makes a single-process device group, one member
        context and host thread per entry, behind this one handle (gfh_create_group)."""
        self._h = _vp()
        L = lib()
        if devices is not None:
            if isinstance(devices, str):
                rc = L.gfh_create_group(0, None, C.byref(self._h))
            elif isinstance(devices, int):          # devices 0 .. n-1 (modulo the visible ones under GADFIT_HIP_GROUP_WRAP=1)
                rc = L.gfh_create_group(devices, None, C.byref(self._h))
            else:
                d = np.ascontiguousarray(devices, dtype=np.int32)
                rc = L.gfh_create_group(d.size, ip(d), C.byref(self._h))
            device = -1
        else:
            rc = L.gfh_create(device, C.byref(self._h))
        if rc != 0:
            raise GadfitHipError(L.gfh_last_error(None).decode())
        self.device = device
        self._tape = None
        self.n_pars = 0
        self.nd = 0

    def close(self):
        if self._h:
            lib().gfh_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def group_size(self):
        return lib().gfh_group_size(self._h)

    def debug_group_allreduce(self, bufs, status, fail_member=-1):
        """test hook: rows of bufs [members][n] summed over the members in place, status -> max (see gadfit_hip.h)"""
        self._chk(lib().gfh_debug_group_allreduce(self._h, dp(bufs), bufs.shape[1], ip(status), fail_member))

    def debug_group_latency(self, n, rounds):
        """(microseconds per host sum of n doubles inside one task, microseconds per fan-out of an empty call) -- gfh_debug_group_latency"""
        out = np.zeros(2)
        self._chk(lib().gfh_debug_group_latency(self._h, n, rounds, dp(out)))
        return float(out[0]), float(out[1])

    def debug_allreduce_latency(self, n, rounds):
        """dict(median_us, p95_us, min_us, max_us, host_round_trip_median_us, nranks) of ONE cross-rank sum of n doubles on this context's
        own path (RCCL all-reduce, or the group's host sum) -- gfh_debug_allreduce_latency; collective"""
        out = np.zeros(6)
        self._chk(lib().gfh_debug_allreduce_latency(self._h, n, rounds, dp(out)))
        return dict(median_us=float(out[0]), p95_us=float(out[1]), min_us=float(out[2]), max_us=float(out[3]),
                    host_round_trip_median_us=float(out[4]), nranks=int(out[5]))

    def _chk(self, rc):
        if rc != 0:
            raise GadfitHipError(lib().gfh_last_error(self._h).decode())

    # --- communicator
    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        if lib().gfh_comm_unique_id(buf) != 0:
            raise GadfitHipError(lib().gfh_last_error(None).decode())
        return buf.raw

    def comm_init(self, nranks, rank, uid):
        self._chk(lib().gfh_comm_init(self._h, nranks, rank, C.create_string_buffer(uid, 128)))

    # --- data / model
    def set_data(self, x, y, w, data_positions):
        x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        w = np.ascontiguousarray(w, dtype=np.float64)
        pos = np.ascontiguousarray(data_positions, dtype=np.int64)
        self.nd = pos.size - 1; self.n_total = int(x.size)
        self._keep_sample(x, pos)
        self._chk(lib().gfh_set_data(self._h, x.size, dp(x), dp(y), dp(w), self.nd, pos.ctypes.data_as(C.POINTER(_i64))))

    def _keep_sample(self, x, pos):
        """up to 64 abscissas per dataset, for the handler that records eval() again when an integrand meets an unrecorded path"""
        self._x_sample = []
        for d in range(pos.size - 1):
            seg = x[pos[d]:pos[d + 1]]
            if seg.size:
                self._x_sample.append((d, seg[np.unique(np.linspace(0, seg.size - 1, min(seg.size, 64)).astype(np.int64))].copy()))

    def set_data_begin(self, x, y, w, data_positions):
        """gfh_set_data_begin: returns at once, the copies run on a thread of the library; the arrays are kept alive here until the
        next call has waited for them"""
        x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        w = np.ascontiguousarray(w, dtype=np.float64)
        pos = np.ascontiguousarray(data_positions, dtype=np.int64)
        self.nd = pos.size - 1; self.n_total = int(x.size)
        self._keep_sample(x, pos)
        self._inflight = (x, y, w, pos)
        self._chk(lib().gfh_set_data_begin(self._h, x.size, dp(x), dp(y), dp(w), self.nd, pos.ctypes.data_as(C.POINTER(_i64))))

    def queue_host_copy(self, dst, src):
        """gfh_queue_host_copy: dst[:] = src on a thread of the library beside the NEXT set_data_begin; wait_host_copy joins it"""
        assert dst.flags['C_CONTIGUOUS'] and src.flags['C_CONTIGUOUS'] and dst.nbytes == src.nbytes
        self._hc = (dst, src)
        self._chk(lib().gfh_queue_host_copy(self._h, dst.ctypes.data_as(_vp), src.ctypes.data_as(_vp), src.nbytes))

    def wait_host_copy(self):
        self._chk(lib().gfh_wait_host_copy(self._h))
        self._hc = None

    def set_aux(self, columns, local=False):
        """auxiliary per-point columns [n_aux][n_total] (or this rank's slice with local=True), gfh_set_aux"""
        a = np.ascontiguousarray(np.atleast_2d(np.asarray(columns, dtype=np.float64)))
        self._chk((lib().gfh_set_aux_local if local else lib().gfh_set_aux)(self._h, a.shape[0], dp(a)))

    def set_data_local(self, n_total, data_positions, begin, x, y, w):
        x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        w = np.ascontiguousarray(w, dtype=np.float64)
        pos = np.ascontiguousarray(data_positions, dtype=np.int64)
        self.nd = pos.size - 1; self.n_total = int(n_total)
        self._chk(lib().gfh_set_data_local(self._h, n_total, self.nd, pos.ctypes.data_as(C.POINTER(_i64)), begin, x.size,
                                           dp(x), dp(y), dp(w)))

    def init_weights(self, error_type):
        self._chk(lib().gfh_init_weights(self._h, error_type))

    def set_model(self, tape, hint_aux=-1):
        """tape: a Tape (straight-line eval()) or tape.Variants (a branching eval(): gfh_set_model_variants; the handler that
        records the paths the device meets during a fit is installed with it)"""
        self._tape = tape
        self.n_pars = tape.n_pars
        if isinstance(tape, T.Variants):
            self._hint_aux = hint_aux
            n, arr = tape.c_array
            self._chk(lib().gfh_set_model_variants(self._h, n, arr, hint_aux))
            self._install_handler()
            self.unseen_log = []      # (x, dataset, outcomes forced, variant index) of every point the device reports from here on
        else:
            self._chk(lib().gfh_set_model(self._h, C.byref(tape.c)))

    def _install_handler(self):
        if getattr(self, '_cb', None) is not None:
            return

        def on_unseen(user, target, n, index, dataset, x, path, n_guards, pars):
            try:
                V = self._tape
                np_ = V.n_pars
                if n == 0:
                    # an integrand met a path through its comparisons that no recording has: eval() over a sample of the data again, at
                    # the parameters of this pass (add_point places the integration variable at its several points)
                    before = len(V)
                    for d, xs in getattr(self, '_x_sample', []):
                        for xv in xs:
                            v = V.add_point(xv, [pars[d * np_ + q] for q in range(np_)])
                            self.unseen_log.append((xv, d, 'integrand', v))
                    if len(V) == before:
                        return 1
                    cnt, arr = V.c_array
                    return lib().gfh_set_model_variants(_vp(target), cnt, arr, getattr(self, '_hint_aux', -1))
                for k in range(n):
                    script = [bool((path[k] >> j) & 1) for j in range(n_guards[k])]
                    d = dataset[k]
                    v = V.add_point(x[k], [pars[d * np_ + q] for q in range(np_)], script=script)
                    self.unseen_log.append((x[k], d, script, v))
                cnt, arr = V.c_array
                return lib().gfh_set_model_variants(_vp(target), cnt, arr, getattr(self, '_hint_aux', -1))
            except Exception as e:      # nothing may propagate through the C frames
                self._handler_error = e
                return 1
        self._cb = UNSEEN_HANDLER(on_unseen)
        self._chk(lib().gfh_set_unseen_handler(self._h, C.cast(self._cb, _vp), None))

    def abscissas(self):
        """the abscissas as they lie on the device, in the order of the concatenated array of set_data (this rank's range filled)"""
        out = np.zeros(self.n_total, dtype=np.float64)
        self._chk(lib().gfh_get_abscissas(self._h, dp(out)))
        return out

    def counters(self):
        """dict(unseen_rounds, mesh_replays, variants, ws_size, ws_size_inner) -- gfh_get_counters"""
        out = (_i64 * 4)()
        self._chk(lib().gfh_get_counters(self._h, out))
        return dict(unseen_rounds=out[0], mesh_replays=out[1], variants=out[2], ws_size=out[3] >> 32, ws_size_inner=out[3] & 0xffffffff)

    def mesh_stats(self):
        """dict(integrals, bisections, unrecorded, sites) of the last recording pass of a quadrature model -- gfh_debug_mesh_stats"""
        out = (_i64 * 4)()
        self._chk(lib().gfh_debug_mesh_stats(self._h, out))
        return dict(integrals=out[0], bisections=out[1], unrecorded=out[2], sites=out[3])

    def device_memory(self):
        """dict(free, total, workspace_pool) in bytes -- gfh_device_memory"""
        out = (_i64 * 3)()
        self._chk(lib().gfh_device_memory(self._h, out))
        return dict(free=out[0], total=out[1], workspace_pool=out[2])

    def n_variants(self):
        return lib().gfh_model_n_variants(self._h)

    def model_needs_hint(self):
        return lib().gfh_model_needs_hint(self._h)

    def model_source(self, active):
        a = np.ascontiguousarray(active, dtype=np.int32)
        n = lib().gfh_model_source(self._h, a.size, ip(a), None, 0)
        if n < 0:
            self._chk(1)
        buf = C.create_string_buffer(n)
        lib().gfh_model_source(self._h, a.size, ip(a), buf, n)
        return buf.value.decode()

    def model_prepare(self, active):
        a = np.ascontiguousarray(active, dtype=np.int32)
        self._chk(lib().gfh_model_prepare(self._h, a.size, ip(a)))

    # --- hot path
    def jacobian_indices(self, active, is_global):
        a = np.ascontiguousarray(active, dtype=np.int32); g = np.ascontiguousarray(is_global, dtype=np.int32)
        jac = np.zeros((self.nd, a.size), dtype=np.int32)
        dim = lib().gfh_jacobian_indices(self.nd, a.size, ip(a), ip(g), ip(jac))
        return jac, dim

    def sweep(self, pars, active, jac, dim):
        p = np.ascontiguousarray(pars, dtype=np.float64); a = np.ascontiguousarray(active, dtype=np.int32)
        j = np.ascontiguousarray(jac, dtype=np.int32)
        JTJ = np.zeros((dim, dim)); JTr = np.zeros(dim); chi2 = C.c_double()
        self._chk(lib().gfh_sweep(self._h, dp(p), ip(a), a.size, ip(j), dim, dp(JTJ), dp(JTr), C.cast(C.byref(chi2), _dp)))
        return JTJ, JTr, chi2.value

    def chi2(self, pars):
        p = np.ascontiguousarray(pars, dtype=np.float64); v = C.c_double()
        self._chk(lib().gfh_chi2(self._h, dp(p), C.cast(C.byref(v), _dp)))
        return v.value

    def omega(self, pars, delta1):
        p = np.ascontiguousarray(pars, dtype=np.float64); d = np.ascontiguousarray(delta1, dtype=np.float64)
        out = np.zeros(d.size)
        self._chk(lib().gfh_omega(self._h, dp(p), dp(d), dp(out)))
        return out

    def aux(self, what, delta1=None, dim=None):
        if what == 0:
            out = np.zeros(dim)
            self._chk(lib().gfh_aux(self._h, 0, None, dp(out)))
        else:
            d = np.ascontiguousarray(delta1, dtype=np.float64); out = np.zeros(3)
            self._chk(lib().gfh_aux(self._h, 1, dp(d), dp(out)))
        return out

    def fit(self, pars, active, is_global, DTD_min=None, verbosity=0, umnigh_a=0.5, **kw):
        p = np.ascontiguousarray(pars, dtype=np.float64).copy()
        a = np.ascontiguousarray(active, dtype=np.int32); g = np.ascontiguousarray(is_global, dtype=np.int32)
        # the C ABI takes plain pointers: the lengths it will read and write are checked here
        if p.size != self.nd * self.n_pars:
            raise GadfitHipError('fit: pars must hold n_datasets x n_pars = %d x %d values, got %d' % (self.nd, self.n_pars, p.size))
        if g.size != self.n_pars:
            raise GadfitHipError('fit: is_global must hold one flag per parameter (%d), got %d' % (self.n_pars, g.size))
        if a.size < 1 or a.min() < 0 or a.max() >= self.n_pars:
            raise GadfitHipError('fit: active parameter indices must lie in [0, %d)' % self.n_pars)
        o = FitOptions()
        for k, v in kw.items():
            if v is None:
                continue
            name = 'lambda' if k in ('lambda_', 'lam', 'lambda') else k
            setattr(o, 'lambda_' if name == 'lambda' else name, v)
            setattr(o, 'has_' + name, 1)
        o.verbosity = verbosity
        o.umnigh_a = umnigh_a
        if DTD_min is not None:
            dm = np.ascontiguousarray(DTD_min, dtype=np.float64)
            dim = self.jacobian_indices(a, g)[1]
            if dm.size != dim:
                raise GadfitHipError('fit: DTD_min must hold dim = %d values, got %d' % (dim, dm.size))
            o.DTD_min = dp(dm)
        r = FitResult()
        self._chk(lib().gfh_fit(self._h, dp(p), a.size, ip(a), ip(g), C.byref(o), C.byref(r)))
        self.umnigh_a = o.umnigh_a
        return p.reshape(np.shape(pars)), r

    def lm_iterate(self, pars, active, is_global, n_iter, state3, DTD):
        """n_iter LM iterations without convergence exits; pars, state3, DTD are updated in place."""
        a = np.ascontiguousarray(active, dtype=np.int32); g = np.ascontiguousarray(is_global, dtype=np.int32)
        assert pars.dtype == np.float64 and pars.flags['C_CONTIGUOUS']
        self._chk(lib().gfh_lm_iterate(self._h, dp(pars), a.size, ip(a), ip(g), n_iter, dp(state3), dp(DTD)))

    def set_loss(self, loss):
        """0 linear (default), 1 cauchy, 2 huber -- the C++ solver's robust costs (lm_solver.cpp:255-284)"""
        self._chk(lib().gfh_set_loss(self._h, int(loss)))

    def set_keep_jacobian(self, mode):
        """0 never, 1 always (default, as the reference), 2 gfh_fit decides from its options"""
        self._chk(lib().gfh_set_keep_jacobian(self._h, int(mode)))

    def set_use_ad(self, on):
        """False: finite differences as gadf_fit(use_ad=.false.) (fitfunction.F90:155-203)"""
        self._chk(lib().gfh_set_use_ad(self._h, 1 if on else 0))

    def set_fd_column_sets(self, on):
        """use_ad = 0 over columns that follow the parameters: set_aux then holds 1 + n_active sets (gfh_set_fd_column_sets)"""
        self._chk(lib().gfh_set_fd_column_sets(self._h, 1 if on else 0))

    def set_load_balancing(self, on):
        """gadf_fit(load_balancing=.true.): adaptive ranges per rank (before set_data; see gadfit_hip.h)"""
        self._chk(lib().gfh_set_load_balancing(self._h, 1 if on else 0))

    def repartition(self, weights):
        w = np.ascontiguousarray(weights, dtype=np.float64)
        self._chk(lib().gfh_repartition(self._h, dp(w)))

    def group_ranges(self):
        n = max(1, self.group_size()); b = np.zeros(n, dtype=np.int64); c = np.zeros(n, dtype=np.int64)
        self._chk(lib().gfh_group_ranges(self._h, b.ctypes.data_as(C.POINTER(_i64)), c.ctypes.data_as(C.POINTER(_i64))))
        return b, c

    def rebalance(self):
        m = _i(0)
        self._chk(lib().gfh_rebalance(self._h, C.byref(m)))
        return bool(m.value)

    def set_lookahead(self, on):
        self._chk(lib().gfh_set_lookahead(self._h, int(bool(on))))

    def debug_set_rank(self, nranks, rank):
        self._chk(lib().gfh_debug_set_rank(self._h, nranks, rank))

    def comm_info(self):
        """(ranks of the RCCL communicator as ncclCommCount reports them -- 0 = none --, all-reduces since reset_timers)"""
        n = _i(0); k = _i64(0)
        self._chk(lib().gfh_comm_info(self._h, C.byref(n), C.byref(k)))
        return n.value, k.value

    def comm_init_from_env(self):
        self._chk(lib().gfh_comm_init_from_env(self._h))

    # --- read-back / timing
    def local_count(self):
        return lib().gfh_local_count(self._h)

    def local_begin(self):
        return lib().gfh_local_begin(self._h)

    def residuals(self):
        out = np.zeros(self.local_count()); self._chk(lib().gfh_get_residuals(self._h, dp(out))); return out

    def weights(self):
        out = np.zeros(self.local_count()); self._chk(lib().gfh_get_weights(self._h, dp(out))); return out

    def omega_vector(self):
        out = np.zeros(self.local_count()); self._chk(lib().gfh_get_omega(self._h, dp(out))); return out

    def jacobian(self, n_act):
        out = np.zeros((self.local_count(), n_act)); self._chk(lib().gfh_get_jacobian(self._h, dp(out))); return out

    def points(self, index, n_act):
        """(residuals [n], Jacobian rows [n][n_act]) of single points by local index -- gfh_get_points"""
        idx = np.ascontiguousarray(index, dtype=np.int64)
        res = np.zeros(idx.size); J = np.zeros((idx.size, n_act))
        self._chk(lib().gfh_get_points(self._h, idx.size, idx.ctypes.data_as(C.POINTER(_i64)), dp(res), dp(J)))
        return res, J

    def timers(self):
        out = np.zeros(8); self._chk(lib().gfh_get_timers(self._h, dp(out))); return out

    def timer_spread(self):
        """{shortest, longest, last} duration in seconds of the STEP 1(+2) kernel and the number of launches counted"""
        out = np.zeros(4); self._chk(lib().gfh_get_timer_spread(self._h, dp(out))); return out

    def set_placement_tries(self, tries):
        """candidate allocations of a large Jacobian buffer that are timed with the kernel about to run (1: take the first)"""
        self._chk(lib().gfh_set_placement_tries(self._h, int(tries)))

    def set_placement_after(self, sweeps):
        """sweeps that must have written a large Jacobian buffer before its candidates are timed (default 48; 0: at the first sweep)"""
        self._chk(lib().gfh_set_placement_after(self._h, int(sweeps)))

    def placement(self):
        """kernel time (ms) on the Jacobian buffer in use, then on the candidates that were freed"""
        out = np.zeros(8); self._chk(lib().gfh_get_placement(self._h, dp(out))); return [float(v) for v in out[:7] if v > 0]

    def placement_copy_GBps(self):
        """device-to-device copy rate the placement's thresholds were scaled with (0: no placement ran)"""
        out = np.zeros(8); self._chk(lib().gfh_get_placement(self._h, dp(out))); return float(out[7])

    def set_timer_detail(self, level):
        self._chk(lib().gfh_set_timer_detail(self._h, int(level)))

    def reset_timers(self):
        lib().gfh_reset_timers(self._h)

    def time_kernel(self, which, reps):
        v = C.c_double(); self._chk(lib().gfh_time_kernel(self._h, which, reps, C.cast(C.byref(v), _dp))); return v.value

    def sync(self):
        self._chk(lib().gfh_sync(self._h))


def read_columns(path, n_columns):
    """gfh_read_columns + gfh_take_columns: the first 2 or 3 numeric columns of a text file as float64 arrays (no GPU needed)"""
    h = _vp(); n = _i64()
    if lib().gfh_read_columns(os.fsencode(path), n_columns, C.byref(h), C.byref(n)) != 0:
        raise GadfitHipError(lib().gfh_last_error(None).decode())
    x = np.empty(n.value); y = np.empty(n.value); w = np.empty(n.value if n_columns == 3 else 0)
    if lib().gfh_take_columns(h, dp(x), dp(y), dp(w) if n_columns == 3 else None) != 0:
        raise GadfitHipError(lib().gfh_last_error(None).decode())
    return (x, y, w) if n_columns == 3 else (x, y)


def partition(n_total, nranks, rank):
    b = _i64(); c = _i64()
    lib().gfh_partition(n_total, nranks, rank, C.byref(b), C.byref(c))
    return b.value, c.value


def debug_packed_layout(nranks, rank, n_total, data_positions, jac_idx, dim, sparse_ok=True):
    """gfh_debug_packed_layout: dict of what rank `rank` of `nranks` derives (no GPU needed)"""
    pos = np.ascontiguousarray(data_positions, dtype=np.int64); jac = np.ascontiguousarray(jac_idx, dtype=np.int32)
    out = np.zeros(8, dtype=np.int64)
    cap = dim * (dim + 1) // 2
    nz_row = np.zeros(cap, dtype=np.int32); nz_col = np.zeros(cap, dtype=np.int32)
    if lib().gfh_debug_packed_layout(nranks, rank, n_total, pos.size - 1, pos.ctypes.data_as(C.POINTER(_i64)), jac.shape[1], ip(jac), dim,
                                     1 if sparse_ok else 0, out.ctypes.data_as(C.POINTER(_i64)), ip(nz_row), ip(nz_col), cap) != 0:
        raise GadfitHipError(lib().gfh_last_error(None).decode())
    d = dict(zip(('packed_n', 'pattern_only', 'nnz', 'hash', 'begin', 'count', 'datasets_held', 'gram_blocks'), (int(v) for v in out)))
    d['nz_row'] = nz_row[:d['nnz']].copy(); d['nz_col'] = nz_col[:d['nnz']].copy()
    return d


def solve_damped(jac_idx, dim, JTJ, DTD, lambda_, rhs, use_structure=True):
    """(JTJ + lambda*diag(DTD)) x = rhs as gfh_fit solves it (gfh_solve_damped); jac_idx [n_datasets][n_act]"""
    jac = np.ascontiguousarray(jac_idx, dtype=np.int32)
    a = np.asfortranarray(JTJ, dtype=np.float64); d = np.ascontiguousarray(DTD, dtype=np.float64)
    b = np.ascontiguousarray(rhs, dtype=np.float64); out = np.zeros(dim)
    if lib().gfh_solve_damped(jac.shape[0], jac.shape[1], jac.ctypes.data_as(_ip), dim, dp(a), dp(d), float(lambda_), dp(b), dp(out),
                              1 if use_structure else 0) != 0:
        raise GadfitHipError(lib().gfh_last_error(None).decode())
    return out


def potr(a, b):
    a = np.asfortranarray(a, dtype=np.float64).copy(order='F'); b = np.ascontiguousarray(b, dtype=np.float64).copy()
    if lib().gfh_potr(a.shape[0], dp(a), dp(b)) != 0:
        raise GadfitHipError(lib().gfh_last_error(None).decode())
    return b
