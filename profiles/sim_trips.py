"""profiles/sim_trips.py -- per-source evaluation counts of the reference algorithm on the C2 workload when (i) f(x1) is not
evaluated, (ii) the provably idle bisection trips are skipped (ideal bound) and (iii) the first iteration of solve() is
taken in the trip in which solvebst leaves: a numpy restatement, for the histogram only (the kernels are checked
against the oracle)."""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracerfortran_amd.synthetic import c2_workload
n=100000
v,d,off,dep=c2_workload(n)
NL=d.size
nl=np.searchsorted(d,dep,side='right')+1   # 1-based layer count of stack (nl==1: top layer)
res_trips=np.zeros(n,int); kinds=np.zeros(n,int)
def fvals(x,H,V,R):
    s=np.sqrt(1-(x*x)*(V*V))
    f=R-np.sum((H*V)*x/s); fp=-np.sum((H*V)/(s*s*s))
    return f,fp
for k in range(n):
    L=nl[k]
    if L==1: continue
    V=v[:L]; H=np.empty(L); H[0]=d[0]; H[1:L-1]=d[1:L-1]-d[:L-2]; H[L-1]=dep[k]-d[L-2]
    R=off[k]; z=dep[k]
    c=z/np.sum(H/V); ct=z/np.sqrt(R*R+z*z); p0=ct/c
    vmax=V.max(); vmi=1/vmax
    while not ((p0*p0)*((vmax+1)**2)<=1): p0/=2
    trips=1
    f,fp=fvals(p0,H,V,R); chk=p0-f/fp
    def newton(x,f,fp,trips,first_known):
        # returns trips; first_known: f,fp at x already evaluated (fused)
        it=0
        for kk in range(1,16):
            if not first_known or kk>1:
                f,fp=fvals(x,H,V,R); trips+=1
            if abs(f)<0.1: return trips
            x=x-f/fp
            if x>vmi: x=vmi-1e-9
        trips+=1
        return trips
    if f<0 or chk<vmi-1e-10:
        kinds[k]=1
        trips=newton(p0,f,fp,trips,True)
    else:
        kinds[k]=2
        x1,x2=1e-10,vmi-1e-12
        x=x2; dx=x1-x2
        # bisect trips (count only those not skipped: emulate skip by bound? count all, and also count 'needed' = trips where f<0.1001)
        bt=0; needed=0
        for kk in range(1,21):
            dx*=0.5; xm=x+dx; fm,fpm=fvals(xm,H,V,R); bt+=1
            if not (fm>=0.1001): needed+=1
            if fm<0: x=xm
            if fm==0: break
            ck=x-fm/fpm
            if ck<vmi-1e-10: x=xm; fused=True; break
            if abs(fm)<0.1: fused=False; break
        else:
            fused=False
        trips+=max(needed,1)   # with ideal skipping
        trips=newton(x,fm,fpm,trips,fused)
    res_trips[k]=trips
m=nl>1
print("mean trips/source (ideal skip, fused):", res_trips[m].mean())
for kd,name in ((1,"newton-branch"),(2,"bisect-branch")):
    a=res_trips[kinds==kd]; print(name, a.size, "mean", a.mean(), "pcts 50/90/99/99.9/max", np.percentile(a,[50,90,99,99.9]), a.max())
print("hist:", np.bincount(res_trips[m])[:25])
