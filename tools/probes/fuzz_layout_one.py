#!/usr/bin/env python3
"""one layout case of the Fortran fuzz with the program's iteration log: python tools/probes/fuzz_layout_one.py SEED [branching]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['FUZZ_VERBOSE'] = '1'
from tests import test_gpu_fortran_fuzz as T
try:
    print(T.run_layout_case(int(sys.argv[1]), tempfile.mkdtemp(prefix='fzone'), branching=len(sys.argv) > 2))
except AssertionError as e:
    print('FAILED', str(e)[:3000])
