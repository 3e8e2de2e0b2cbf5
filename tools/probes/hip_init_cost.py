#!/usr/bin/env python3
"""Where does the time of the first gfh_create of a process go?  (fresh process, no torch)"""
import ctypes as C, time
t0 = time.perf_counter()
hip = C.CDLL('/opt/rocm/lib/libamdhip64.so')
t1 = time.perf_counter()
n = C.c_int()
hip.hipGetDeviceCount(C.byref(n)); t2 = time.perf_counter()
hip.hipSetDevice(0); t3 = time.perf_counter()
s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1); t4 = time.perf_counter()
p = C.c_void_p(); hip.hipMalloc(C.byref(p), 3248); t5 = time.perf_counter()
h = C.c_void_p(); hip.hipHostMalloc(C.byref(h), 64, 0); t6 = time.perf_counter()
print('dlopen %.1f ms  hipGetDeviceCount %.1f  hipSetDevice %.1f  stream %.1f  hipMalloc %.1f  hipHostMalloc %.1f' %
      tuple(1e3 * d for d in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)))
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t0 = time.perf_counter()
from gadfit_amd import _lib
t1 = time.perf_counter()
c = _lib.Context(0); t2 = time.perf_counter()
print('import _lib (loads libgadfit_hip, hiprtc, rccl) %.1f ms   first Context %.1f ms' % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
c.close()
