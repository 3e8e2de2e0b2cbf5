"""Host logic of the Python mirror of the driver API (gadfit_amd/gadfit.py): argument forms of
gadf_set, the data-file reader (read_data semantics, gadfit.F90:212-215, 422-437) and the reference's
error messages -- everything that happens before the first device call.  CPU only."""
import numpy as np
import pytest

from gadfit_amd import gadfit as gf
from gadfit_amd.ad import exp, Real, advar, trace_model
from gadfit_amd import tape as T


class decay(gf.fitfunc):
    def init(self):
        self.allocate(3)
        self.set(1, 'I0'); self.set(2, 'tau'); self.set(3, 'bgr')

    def eval(self, x):
        return self.pars[0] * exp(-(x / self.pars[1])) + self.pars[2]


def test_gadf_set_forms_and_state():
    gf.gadf_init(decay(), 2)
    gf.gadf_set(1, 'I0', 2.0, True)          # local by name
    gf.gadf_set(2, 1, 3.0, True)             # local by index
    gf.gadf_set('tau', np.float32(1.5), True)  # global, real32 value
    gf.gadf_set(3, 0.25)                     # global, passive (active absent)
    assert [p.val for p in gf.fitfuncs[0].pars] == [2.0, 1.5, 0.25]
    assert [p.val for p in gf.fitfuncs[1].pars] == [3.0, 1.5, 0.25]
    assert gf._S.active == [True, True, False] and gf._S.is_global == [False, True, True]
    assert gf.fitfuncs[0].get_index('bgr') == 3 and gf.fitfuncs[0].get_name(2) == 'tau'
    with pytest.raises(gf.GadfitError, match='Invalid dataset index'):
        gf.gadf_set(3, 1, 1.0, True)
    with pytest.raises(KeyError):
        gf.gadf_set('nope', 1.0)
    gf.gadf_close()
    with pytest.raises(gf.GadfitError, match='Call gadf_init first'):
        gf.gadf_set(1, 1.0)


def test_data_file_reader_skips_non_numeric_lines(tmp_path):
    f = tmp_path / 'd.txt'
    f.write_text('# x y sigma\n1.0 2.0 0.1\nfoo bar\n\n2.0, 3.5, 0.2\n3.0 1.25 0.3 trailing\n')
    gf.gadf_init(decay(), 1)
    gf.gadf_add_dataset(str(f))
    x, y, w = gf._S.datasets[0]
    assert x.tolist() == [1.0, 2.0, 3.0] and y.tolist() == [2.0, 3.5, 1.25] and w.tolist() == [0.1, 0.2, 0.3]
    with pytest.raises(gf.GadfitError, match='Too many calls to gadf_add_dataset'):
        gf.gadf_add_dataset([1.0], [2.0])
    gf.gadf_close()
    empty = tmp_path / 'e.txt'
    empty.write_text('no numbers here\n')
    gf.gadf_init(decay(), 1)
    with pytest.raises(gf.GadfitError, match='contains no valid data points'):
        gf.gadf_add_dataset(str(empty))
    gf.gadf_close()


def test_fit_preconditions_raise_before_any_device_call():
    gf.gadf_init(decay(), 2)
    gf.gadf_add_dataset([1.0, 2.0], [1.0, 0.5])
    with pytest.raises(gf.GadfitError, match='There are no active parameters'):
        gf.gadf_fit(max_iter=1)
    gf.gadf_set(1, 1.0, True)
    with pytest.raises(gf.GadfitError, match='Some datasets are missing'):
        gf.gadf_fit(max_iter=1)
    # use_ad=.false. is a device path like any other (gfh_set_use_ad): without a GPU it fails in the library, loudly
    with pytest.raises(Exception, match='Some datasets are missing'):
        gf.gadf_fit(use_ad=False)
    gf.gadf_close()


def test_tracer_overload_resolution_matches_fortran():
    """(advar,real) / (real,advar) / **integer / unary minus / real arithmetic on x."""
    def m(p, x):
        a = p[0]
        return (-a) * 2 + (x - 1.0) / a + a ** 2 + 2 ** a + a ** 0.5 + abs(x) * a
    t = trace_model(m, 1)
    nodes, res = t.subtapes[0]
    ops = [n[0] for n in nodes]
    assert ops.count(T.POWI) == 1 and ops.count(T.POW) == 2
    # -a is 0.0 - a (AD:598-601): a SUB whose first operand is the literal 0
    sub = [n for n in nodes if n[0] == T.SUB and nodes[n[1]][0] == T.CONST and nodes[n[1]][4] == 0.0]
    assert len(sub) == 1
    # x - 1.0 and abs(x) stay real-typed
    assert any(n[0] == T.SUB and n[3] & T.F_REAL for n in nodes) and any(n[0] == T.ABS and n[3] & T.F_REAL for n in nodes)
    # comparisons on traced values are refused (data-dependent control flow)
    with pytest.raises(TypeError):
        trace_model(lambda p, x: p[0] if p[0] > 1.0 else p[0] * 2, 1)
    with pytest.raises(RuntimeError, match='outside of model tracing'):
        advar(1.0)
