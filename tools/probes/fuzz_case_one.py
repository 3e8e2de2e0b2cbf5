#!/usr/bin/env python3
"""one single-dataset case of the Fortran fuzz with the program's output: python tools/probes/fuzz_case_one.py SEED N [branching|integral]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['FUZZ_VERBOSE'] = '1'
from tests import test_gpu_fortran_fuzz as T
kind = sys.argv[3] if len(sys.argv) > 3 else ''
try:
    print(T.run_case(int(sys.argv[1]), int(sys.argv[2]), tempfile.mkdtemp(prefix='fzone'), branching='branching' in kind, integral='integral' in kind))
except AssertionError as e:
    print('FAILED', str(e)[:2000])
