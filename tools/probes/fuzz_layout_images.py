#!/usr/bin/env python3
"""one layout case of the Fortran fuzz as 1, 2 and 3 images: python tools/probes/fuzz_layout_images.py SEED [branching]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import fortran_fuzz as FZ
from tests import test_gpu_fortran_fuzz as T
seed = int(sys.argv[1])
branching = len(sys.argv) > 2
orig = FZ.make_layout_case
for img in (1, 2, 3):
    def patched(s, img=img, **kw):
        c = orig(s, **kw); c['images'] = img; return c
    FZ.make_layout_case = patched
    try:
        print('images', img, T.run_layout_case(seed, tempfile.mkdtemp(prefix='fzimg'), branching=branching))
    except AssertionError as e:
        print('images', img, 'FAILED', str(e)[:600])
