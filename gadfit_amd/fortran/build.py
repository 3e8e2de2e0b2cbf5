#!/usr/bin/env python3
"""Builds the Fortran binding (modules ad, fitfunction, gadfit over ISO_C_BINDING) with amdflang
into gadfit_amd/fortran/build/ (libgadfit_f.a + .mod files) and the Fortran test programs
under tests/fortran/ into tests/fortran/build/."""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, 'build')
LIBDIR = os.path.join(os.path.dirname(HERE), 'lib')
MODULES = ['gadf_constants.F90', 'messaging.F90', 'misc.F90', 'ad.F90', 'fitfunction.F90', 'numerical_integration.F90',
           'gadfit_hip_c.F90', 'gadfit.F90']


def fc():
    for c in ('amdflang', 'flang'):
        p = shutil.which(c) or (os.path.join('/opt/rocm/bin', c) if os.path.exists(os.path.join('/opt/rocm/bin', c)) else None)
        if p:
            return p
    return None


def build(verbose=False):
    comp = fc()
    if comp is None:
        print('no Fortran compiler (amdflang) found: Fortran binding not built')
        return False
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for m in MODULES:
        o = os.path.join(OUT, os.path.splitext(m)[0] + '.o')
        src = os.path.join(HERE, m)
        if not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(os.path.join(HERE, x)) for x in MODULES):
            # (OpenMP only where the directives are: gadfit.F90's parallel loop over recordings)
            cmd = [comp, '-O3', '-cpp'] + (['-fopenmp'] if m == 'gadfit.F90' else []) + ['-fPIC', '-module-dir', OUT, '-I', OUT, '-c', src, '-o', o]
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    # the recorder's per-thread checking state: plain C with native thread-local storage (see ad_tls.c)
    co = os.path.join(OUT, 'ad_tls.o')
    csrc = os.path.join(HERE, 'ad_tls.c')
    if not os.path.exists(co) or os.path.getmtime(co) < os.path.getmtime(csrc):
        subprocess.check_call(['gcc', '-O2', '-fPIC', '-ftls-model=initial-exec', '-c', csrc, '-o', co])
    objs.append(co)
    lib = os.path.join(OUT, 'libgadfit_f.a')
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(o) for o in objs):
        if os.path.exists(lib):
            os.remove(lib)
        subprocess.check_call(['ar', 'rcs', lib] + objs)
    tdir = os.path.join(ROOT, 'tests', 'fortran')
    tout = os.path.join(tdir, 'build')
    os.makedirs(tout, exist_ok=True)
    hip_lib = os.path.join(LIBDIR, 'libgadfit_hip.so')
    for src in sorted(glob.glob(os.path.join(tdir, '*.F90'))):
        exe = os.path.join(tout, os.path.splitext(os.path.basename(src))[0])
        # (the programs link libgadfit_hip.so dynamically: a rebuilt library needs no relink)
        if os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(src), os.path.getmtime(lib)) and os.path.exists(hip_lib):
            continue
        cmd = [comp, '-O2', '-cpp', '-fopenmp', '-I', OUT, '-module-dir', tout, src, lib, '-L' + LIBDIR, '-lgadfit_hip',
               '-Wl,-rpath,' + LIBDIR, '-Wl,-rpath,/opt/rocm/lib', '-Wl,-rpath,/opt/rocm/lib/llvm/lib', '-o', exe]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return True


if __name__ == '__main__':
    ok = build(verbose='-v' in sys.argv)
    sys.exit(0 if ok or fc() is None else 1)
