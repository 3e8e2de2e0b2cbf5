import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
from tests import test_gpu_fortran_fuzz as T
w = tempfile.mkdtemp()
orig = T.first_pass_deviation
def verbose(path, first, record=0):
    lines = open(path).read().splitlines()
    head = lines[3*record].split(); dim = int(head[2]); chi2 = float(head[4])
    JTr = np.array(lines[3*record+1].split()[1:], dtype=float); JTJ = np.array(lines[3*record+2].split()[1:], dtype=float).reshape(dim, dim)
    J0, r0, c0 = first['JTJ'], first['JTres'], first['chi2']
    d = np.sqrt(np.abs(np.diag(J0)))
    print('dim', dim, 'chi2 dev', abs(chi2-c0)/c0, 'JTJ dev', np.abs(JTJ-J0)/np.outer(d,d), 'JTres dev', np.abs(JTr-r0)/(d*np.sqrt(c0)), 'JTJ', J0, 'JTres', r0, 'chi2', c0)
    return orig(path, first, record)
T.first_pass_deviation = verbose
T.TOL_FIRST = 1e-9
print(T.run_case(100, 40000, w))
print(T.run_case(100, 4000, w))
