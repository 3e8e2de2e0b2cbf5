"""oracle/_ref: the reference's OWN four fit programs (fortran/tests/1_gaussian.F90 ... 4_multiple_curves.F90 with their *_data.F90
modules), compiled UNCHANGED from where they lie under /root/reference and linked against this repository's modules and library.
Test infrastructure (tests/test_reference_programs.py runs them on the GPU): each program fits its data through gadf_init /
gadf_add_dataset / gadf_set / gadf_fit / gadf_print, compares the fitted parameter with the constant the reference holds
(1_gaussian.F90:64-65: 1e-13 absolute; 2_integral_single.F90:73-75: 1e-11; 3_integral_double.F90:95-97: 1e-9;
4_multiple_curves.F90:54-65: 1e-13) and `error stop`s when it is off -- the reference's own acceptance test, with the product
under it.

Nothing of the reference enters the repository: the outputs are executables in oracle/_ref/ (git-ignored; they travel to the GPU
box with the snapshot like the built libraries).  Where /root/reference is absent (the GPU box) this script does nothing and the
prebuilt files are used.

The one compile-time mapping: `-Dthis_image()=1`.  The programs ask `this_image() == 1` before their self-check; flang has no
coarray lowering, with any library, and this build's images are processes / device-group members that the
library counts itself (one image here).  The sources are not touched.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/fortran/tests'
OUT = os.path.join(ROOT, 'oracle', '_ref')
MODS = os.path.join(ROOT, 'gadfit_amd', 'fortran', 'build')
LIBDIR = os.path.join(ROOT, 'gadfit_amd', 'lib')
PROGRAMS = ['1_gaussian', '2_integral_single', '3_integral_double', '4_multiple_curves']


def main():
    fc = shutil.which('amdflang') or ('/opt/rocm/bin/amdflang' if os.path.exists('/opt/rocm/bin/amdflang') else None)
    if not os.path.isdir(REF) or fc is None:
        print('oracle/_ref: reference or Fortran compiler absent, nothing built')
        return 0
    lib_f = os.path.join(MODS, 'libgadfit_f.a')
    if not os.path.exists(lib_f) or not os.path.exists(os.path.join(LIBDIR, 'libgadfit_hip.so')):
        print('oracle/_ref: build gadfit_amd/build.py and gadfit_amd/fortran/build.py first', file=sys.stderr)
        return 1
    os.makedirs(OUT, exist_ok=True)
    for name in PROGRAMS:
        work = os.path.join(OUT, 'obj_' + name)
        os.makedirs(work, exist_ok=True)
        data_o = os.path.join(work, 'data.o')
        subprocess.check_call([fc, '-O2', '-cpp', '-I', MODS, '-module-dir', work, '-c', os.path.join(REF, name + '_data.F90'), '-o', data_o])
        subprocess.check_call([fc, '-O2', '-cpp', '-fopenmp', '-Dthis_image()=1', '-I', MODS, '-I', work, '-module-dir', work,
                               os.path.join(REF, name + '.F90'), data_o, lib_f, '-L' + LIBDIR, '-lgadfit_hip',
                               '-Wl,-rpath,$ORIGIN/../../gadfit_amd/lib', '-Wl,-rpath,' + LIBDIR, '-Wl,-rpath,/opt/rocm/lib', '-Wl,-rpath,/opt/rocm/lib/llvm/lib',
                               '-o', os.path.join(OUT, name)])
        shutil.rmtree(work)
    # example.F90 (the user guide's worked example, user_guide.tex:921-960): reads two data files from the directory the build names
    # (DATA_DIR).  The files are the committed fixtures tests/golden/curve{1,2}_xy.txt (the same 100 records each) under the names
    # the program asks for.
    data_dir = os.path.join(OUT, 'example_data')
    os.makedirs(data_dir, exist_ok=True)
    for k in (1, 2):
        shutil.copyfile(os.path.join(ROOT, 'tests', 'golden', 'curve%d_xy.txt' % k), os.path.join(data_dir, 'example_data%d' % k))
    work = os.path.join(OUT, 'obj_example')
    os.makedirs(work, exist_ok=True)
    subprocess.check_call([fc, '-O2', '-cpp', '-fopenmp', "-DDATA_DIR='%s'" % data_dir, '-I', MODS, '-module-dir', work,
                           os.path.join(REF, 'example.F90'), lib_f, '-L' + LIBDIR, '-lgadfit_hip',
                           '-Wl,-rpath,$ORIGIN/../../gadfit_amd/lib', '-Wl,-rpath,' + LIBDIR, '-Wl,-rpath,/opt/rocm/lib', '-Wl,-rpath,/opt/rocm/lib/llvm/lib',
                           '-o', os.path.join(OUT, 'example')])
    shutil.rmtree(work)
    print('oracle/_ref: ' + ' '.join(PROGRAMS) + ' example')
    return 0


if __name__ == '__main__':
    sys.exit(main())
