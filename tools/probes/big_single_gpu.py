#!/usr/bin/env python3
"""N = 1e8 points x 32 active parameters on ONE MI355X (28.8 GB of Jacobian in the 288 GB of HBM): the whole of
BASELINE config 5 on a single GPU.  One-off capability/timing check, not part of the test suite."""
import os
import sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M
n=100_000_000
truth=M.gauss8_truth()
t0=time.time()
x,y,s=M.make_single_slice(M.gauss8_numpy, truth, n, 0, n, 0.0, 100.0)
print('data %.1f s'%(time.time()-t0), flush=True)
ctx=_lib.Context(0); ctx.set_model(trace_model(M.model_gauss8,32))
t0=time.time(); ctx.set_data_local(n,[0,n],0,x,y,s); ctx.init_weights(4); print('upload %.1f s'%(time.time()-t0), flush=True)
act=list(range(32)); start=M.start_values(truth).reshape(1,32)
jac,dim=ctx.jacobian_indices(act,[0]*32)
JTJ,JTr,chi2=ctx.sweep(start,act,jac,dim)
c2=ctx.chi2(start)
print('chi2 sweep %.10e chi2 kernel %.10e rel %.1e'%(chi2,c2,abs(chi2-c2)/c2), flush=True)
for i in range(30): ctx.sweep(start,act,jac,dim)
ctx.reset_timers()
for i in range(20): ctx.sweep(start,act,jac,dim)
tm=ctx.timers(); ms=1e3*tm[0]/tm[6]
print('N=1e8: fused kernel %.3f ms = %.0f GB/s'%(ms, 288*n/(ms*1e-3)/1e9), flush=True)
out,r=ctx.fit(start,act,[0]*32,lambda_=1.0,max_iter=6)
print('fit: iterations %d chi2/dof %.6f  %.2f ms per iteration'%(r.iterations, r.chi2/(n-32), 1e3*r.seconds/r.iterations))
ctx.close()
