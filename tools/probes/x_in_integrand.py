import sys
sys.path.insert(0,'/root/repo')
import numpy as np
from gadfit_amd import _lib, ad
from gadfit_amd.ad import trace_model, integrate, exp
from oracle import binding as orc
def model(p, x):
    def f(t, q):
        return q[0]*exp(-q[1]*t*t)*(1.0 + 0.1*x) + ad.aux(0)*t      # x and an auxiliary column reach the integrand without pars(:)
    return integrate(f, [p[0], p[1]], 0.0, 2.0)
t = trace_model(model, 2)
t.set_integration(rel_error=1e-10)
dev = int(sys.argv[1]) if len(sys.argv) > 1 else -1
c=_lib.Context(dev); c.set_model(t)
c.model_prepare([0,1]); print('compiled')
src=c.model_source([0,1]); print([l for l in src.splitlines() if 'gfh_lane' in l][:6])
x=np.linspace(0.1,3,700); y=np.ones(700); w=np.ones(700); aux=np.sin(x)[None,:]
p=orc.OracleProblem(t,[x],[y],[w],[[1.3,0.7]],[0,1],[0,0], aux=aux)
JTJ0,JTr0,res0,JT0=p.sweep(want_J=True); chi0,_=p.chi2()
print('oracle', chi0)
if dev >= 0:
    c.set_data(x,y,w,[0,700]); c.set_aux(aux)
    jac,dim=c.jacobian_indices([0,1],[0,0])
    JTJ,JTr,chi2=c.sweep([[1.3,0.7]],[0,1],jac,dim)
    print('device', chi2, abs(chi2-chi0)/chi0, np.max(np.abs(JTJ-JTJ0)/np.abs(JTJ0)), abs(c.chi2([[1.3,0.7]])-chi0)/chi0)
    d1=np.array([0.3,-0.05]); om0,jto0=p.omega(d1,JT0); jto=c.omega([[1.3,0.7]],d1)
    print('omega', np.max(np.abs(c.omega_vector()-om0))/np.max(np.abs(om0)), np.max(np.abs(jto-jto0)/np.abs(jto0)))
