#!/usr/bin/env python3
"""Per-pass latency of a small global fit (2 datasets x 100 points, 7 active parameters each): the size most gadfit
fits have.  Compare GADFIT_HIP_TAIL=0 / GADFIT_HIP_MERGE_SMALL=0."""
import os
import sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gadfit_amd import _lib
from gadfit_amd.ad import trace_model
from tests import models as M
sizes=[100,100]
xs,ys,ss,truths=M.make_global7(2,sizes)
t=trace_model(M.model_global7,7)
pars=np.array([M.start_values(tr) for tr in truths]); pars[:,4:]=M.start_values(M.GLOBAL7_TAUS)
c=_lib.Context(0); c.set_model(t)
c.set_data(np.concatenate(xs),np.concatenate(ys),np.concatenate([1/s for s in ss]),[0,100,200])
act=list(range(7)); glob=[0,0,0,0,1,1,1]
jac,dim=c.jacobian_indices(act,glob)
for i in range(50): c.sweep(pars,act,jac,dim)
t0=time.perf_counter()
for i in range(500): c.sweep(pars,act,jac,dim)
print('2 datasets x 100 points: %.1f us per sweep call'%((time.perf_counter()-t0)/500*1e6))
c.close()
