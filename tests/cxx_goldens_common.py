"""Shared set-up of the C++-side golden cases (tests/golden/goldens.py CXX_*)."""
import numpy as np

from gadfit_amd.ad import trace_model
from oracle import binding as orc
from tests.golden import goldens as G


def case(k):
    """problem k of CXX_INDEXING -> (tape, xs, ys, ws, pars[2][3], active list, is_global, expected dict)"""
    d = G.data()['cxx_lm_solver']
    fd = d['fix_d']
    idx, act, chi2, tau, i00, b0, i01, b1 = G.CXX_INDEXING[k]
    pars = np.array([[fd[idx[0]], fd[G.CXX_TAU_START_IDX], fd[idx[1]]], [fd[idx[2]], fd[G.CXX_TAU_START_IDX], fd[idx[3]]]])
    # a parameter is active iff it is active in ANY dataset in the C++ API; the Fortran-style driver has one
    # active flag per parameter index.  The sections where I0 (or bgr) is active in one dataset only are
    # expressed by freezing nothing here: they are skipped by the caller (not representable by gadf_set).
    xs = [np.array(d['x_data_1']), np.array(d['x_data_2'])]
    ys = [np.array(d['y_data_1']), np.array(d['y_data_2'])]
    ws = [np.ones(100), np.ones(100)]
    exp = dict(chi2=chi2, tau=tau, I0=(i00, i01), bgr=(b0, b1))
    return trace_model(G.model_exponential_cxx, 3), xs, ys, ws, pars, act, exp


def representable(act):
    """gadf_set keeps one active flag per parameter index (gadfit.F90:255-273): both datasets alike."""
    return act[0] == act[2] and act[1] == act[3]


def active_list(act):
    a = []
    if act[0]:
        a.append(0)
    a.append(1)
    if act[1]:
        a.append(2)
    return a
