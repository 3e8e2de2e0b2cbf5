"""Does a long batch of small fits leak?  2000 context cycles (create, model, data, fit, destroy) of two alternating models plus every
50th cycle a model of its own (so that the per-process cache of loaded code objects, rtc.cpp, also evicts): resident set size of the
process and free device memory at the start, in the middle and at the end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model, exp


def rss_mb():
    for ln in open('/proc/self/status'):
        if ln.startswith('VmRSS'):
            return int(ln.split()[1]) / 1024.0


def models(k):
    if k % 50 == 49:
        c = 1.0 + 0.001 * k
        return trace_model(lambda p, x: p[0] * exp(-((x - p[1]) / p[2]) ** 2) * c + p[3], 4)
    if k % 2:
        return trace_model(lambda p, x: p[0] * exp(-((x - p[1]) / p[2]) ** 2) + p[3], 4)
    return trace_model(lambda p, x: p[0] * exp(-((x - p[1]) / p[2]) ** 2) + p[3] + 0.0 * x, 4)


n = 1000
x = 10.0 * (np.arange(n) + 0.5) / n
y = 3.0 * np.exp(-((x - 4.5) / 0.8) ** 2) + 0.5 + 1e-3 * np.sin(977.0 * x)
t0 = time.time()
for k in range(2000):
    ctx = _lib.Context(0)
    ctx.set_model(models(k))
    ctx.set_data(x, y, np.ones(n), [0, n])
    ctx.fit(np.array([[2.5, 4.3, 1.0, 0.3]]), [0, 1, 2, 3], [0, 0, 0, 0], lambda_=1.0, max_iter=10)
    ctx.close()
    if k in (0, 9, 99, 999, 1999):
        free, total = torch.cuda.mem_get_info(0)
        print('cycle %4d: RSS %.1f MB, device memory in use %.1f MB, %.2f ms per cycle so far' % (k + 1, rss_mb(), (total - free) / 2**20, 1e3 * (time.time() - t0) / (k + 1)), flush=True)
