"""What hipcc makes of the generated kernels (no GPU needed: cross-compilation to gfx950 and a look at the code object's
metadata).  Pins two assumptions the generated source makes about its own compilation:
  * no kernel of the headline model uses scratch memory (a parameter block copied to the stack would);
  * the tangent block of the STEP 3 kernels, which the source addresses as kernarg segment + 16 + sizeof(gfh_parg)
    (codegen.cpp, GFH_DPARS_CONST), sits at that offset."""
import os
import re
import shutil
import subprocess

import pytest

from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M

HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def _kernels(asm):
    md = asm[asm.index('amdhsa.kernels:'):]
    out = {}
    for blk in md.split('  - .agpr_count:')[1:]:
        name = re.findall(r'\n    \.name:\s+(\S+)', blk)[0]
        out[name] = dict(args=[(int(o), int(s)) for o, s in re.findall(r'\.offset:\s+(\d+)\n\s+\.size:\s+(\d+)', blk)],
                         scratch=int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk).group(1)),
                         vgprs=int(re.search(r'\n    \.vgpr_count:\s+(\d+)', blk).group(1)))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='no hipcc')
def test_headline_kernels_no_scratch_and_tangent_offset(tmp_path):
    ctx = _lib.Context(-1)
    ctx.set_model(trace_model(M.model_gauss8, 32))
    src = ctx.model_source(list(range(32)))
    ctx.close()
    parg = int(re.search(r'#define GFH_PARG (\d+)', src).group(1))
    assert parg == 32                                  # one dataset: the parameter block travels with the kernel arguments
    f = tmp_path / 'g8.hip'
    f.write_text('#include <hip/hip_runtime.h>\n' + src)
    asm = tmp_path / 'g8.s'
    # the options hiprtc gets (rtc.cpp)
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-ffp-contract=on', '-std=c++17', '-S', '--cuda-device-only',
                           '-w', '-o', str(asm), str(f)])
    ks = _kernels(asm.read_text())
    for name in ('gfh_k_sweep', 'gfh_k_sweep_gram', 'gfh_k_chi2', 'gfh_k_omega', 'gfh_k_omega_jt'):
        assert ks[name]['scratch'] == 0, (name, ks[name])
        assert ks[name]['vgprs'] <= 256, (name, ks[name])        # (beyond 256 the unified register file halves the waves per SIMD)
    for name in ('gfh_k_omega', 'gfh_k_omega_jt'):
        a = ks[name]['args']
        assert a[:4] == [(0, 8), (8, 8), (16, 8 * parg), (16 + 8 * parg, 8 * parg)], (name, a[:5])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='no hipcc')
@pytest.mark.parametrize('which', ['single', 'double'])
@pytest.mark.parametrize('form', ['fast', 'user'])
def test_quadrature_kernels_stay_within_8_kb_of_scratch_per_lane(tmp_path, which, form):
    """The interval workspaces of integrate() (numerical_integration.F90:40-51, 128-134): the fast form (up to 100 intervals per
    level, fewer where the per-interval gradients of nested integrals would not fit) is private scratch within 8 KB per lane; the
    user's sizes (default 1000 intervals = 32 KB per lane and level) live in the context's pool in global memory -- no generated
    kernel asks the runtime for more than 8 KB of scratch per lane (beyond that the runtime's device-wide reservation runs to
    gigabytes and two queues asking at once can end the process)."""
    from tests.golden import goldens as G
    if which == 'single':
        t = trace_model(G.model_integral_single, 2); t.set_integration(rel_error=1e-12)
    else:
        t = trace_model(G.model_integral_double, 2); t.set_integration(rel_error=1e-5, rel_error_inner=1e-6, dbl=True)
    old = os.environ.get('GADFIT_HIP_WS_FAST')
    if form == 'user':
        os.environ['GADFIT_HIP_WS_FAST'] = '0'
    try:
        ctx = _lib.Context(-1)
    finally:
        if form == 'user':
            if old is None:
                del os.environ['GADFIT_HIP_WS_FAST']
            else:
                os.environ['GADFIT_HIP_WS_FAST'] = old
    ctx.set_model(t)
    src = ctx.model_source([0, 1])
    carried = ctx.counters()
    ctx.close()
    assert ('#define GFH_WSG 1' in src) == (form == 'user')
    if form == 'user':
        # (round 5) the pool's rows hold the panels' gradients too -- 4 + NQ fields of 64 lanes per interval and level -- so that the
        # sweep's final pass has nothing to re-evaluate in this form either: single: 2 integrand parameters; double: 3 outside, 1 inside
        rows = ('#define GFH_WSG_ROW1 384LL', '#define GFH_WSG_ROW2 256LL') if which == 'single' else ('#define GFH_WSG_ROW1 448LL', '#define GFH_WSG_ROW2 320LL')
        assert all(r in src for r in rows) and 'wl_[(4 + j) * 64 + (long long)q * row_]' in src
        assert ('#define GFH_WSG_WAVE %dLL' % (1000 * 384 if which == 'single' else 1000 * 448 + 1000 * 320)) in src
    assert carried['ws_size'] == (1000 if form == 'user' else 100 if which == 'single' else 82)
    f = tmp_path / 'q.hip'
    f.write_text('#include <hip/hip_runtime.h>\n' + src)
    asm = tmp_path / 'q.s'
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-ffp-contract=on', '-std=c++17', '-S', '--cuda-device-only',
                           '-w', '-o', str(asm), str(f)])
    ks = _kernels(asm.read_text())
    assert {'gfh_k_sweep', 'gfh_k_chi2', 'gfh_k_omega'} <= set(ks)
    for name, k in ks.items():
        assert k['scratch'] <= 8192, (name, k)
        if form == 'user':
            assert k['scratch'] <= 1024, (name, k)          # register spills only: the workspaces are not there


def test_committed_ad_module_is_what_its_generator_writes():
    """gadfit_amd/fortran/ad.F90 is generated (python gen_ad.py > ad.F90): the committed module must be the generator's output"""
    import subprocess
    import sys
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gadfit_amd', 'fortran')
    out = subprocess.run([sys.executable, os.path.join(here, 'gen_ad.py')], capture_output=True, text=True, timeout=120, check=True).stdout
    assert out == open(os.path.join(here, 'ad.F90')).read()
