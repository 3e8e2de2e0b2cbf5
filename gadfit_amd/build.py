"""Builds libgadfit_hip.so (hand-written HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libgadfit_hip.so')
SOURCES = ['kernels.hip', 'codegen.cpp', 'rtc.cpp', 'context.cpp', 'group.cpp', 'lm.cpp', 'reader.cpp']
HEADERS = ['exports.map', 'kernels.h', 'model.h', 'rtc.h', 'context.h', 'group.h', '../../include/gadfit_hip.h', '../../include/gadfit_tape.h']
ROCM = os.environ.get('ROCM_PATH', '/opt/rocm')


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_lib(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, os.path.splitext(s)[0] + '.o')
        cmd = [os.path.join(ROCM, 'bin', 'hipcc'), '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '--offload-arch=gfx950',
               '-Wall', '-Wno-unused-result', '-Wno-unused-value', '-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [os.path.join(ROCM, 'bin', 'hipcc'), '-shared', '-fPIC', '--offload-arch=gfx950', '-o', LIB] + objs + \
          ['-L' + os.path.join(ROCM, 'lib'), '-lhiprtc', '-lrccl', '-pthread', '-Wl,-rpath,' + os.path.join(ROCM, 'lib'),
           '-Wl,--version-script=' + os.path.join(CSRC, 'exports.map')]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build_lib(force='--force' in sys.argv, verbose=True))
