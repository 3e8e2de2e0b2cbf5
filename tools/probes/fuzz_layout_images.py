#!/usr/bin/env python3
"""one layout case of the Fortran fuzz as 1, 2 and 3 images: python tools/probes/fuzz_layout_images.py SEED"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import fortran_fuzz as FZ
from tests import test_gpu_fortran_fuzz as T
seed = int(sys.argv[1])
orig = FZ.make_layout_case
for img in (1, 2, 3):
    def patched(s, img=img):
        c = orig(s); c['images'] = img; return c
    FZ.make_layout_case = patched
    try:
        print('images', img, T.run_layout_case(seed, tempfile.mkdtemp(prefix='fzimg')))
    except AssertionError as e:
        print('images', img, 'FAILED', str(e)[:600])
