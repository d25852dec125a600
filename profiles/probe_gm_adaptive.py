import sys, os, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import pymoc_amd as gpu
from pymoc_amd import configs
from pymoc_amd.device import DeviceArray
from conftest import load_golden, relerr
import test_oracle_golden as T
import oracle as O
g=load_golden('psi_so')
worst=0; worst_o=0
for k in range(int(g['ncases'])):
    p='c%02d_'%k
    kw=T._so_kwargs(g,p)
    if kw['c'] is None: continue
    tau=g[p+'tau']
    KGM=kw.pop('KGM')
    tau_in = float(tau) if tau.ndim==0 else tau[None]
    t=gpu.PsiSOBatch(g[p+'z'],g[p+'y'],1,tau=tau_in,KGM=KGM,bvp_refine=-1,**kw)
    t.update(DeviceArray.from_host(g[p+'b'][None]), DeviceArray.from_host(g[p+'bs'][None]))
    GM=t.Psi_GM.download()[0]; st=t.status.download()[0]
    e=relerr(GM,g[p+'Psi_GM']); worst=max(worst,e)
    ro=O.psi_so_solve(g[p+'z'],g[p+'y'],g[p+'b'],g[p+'bs'],float(tau) if tau.ndim==0 else tau,KGM=KGM,bvp_refine=-1,**kw)
    eo=relerr(GM,ro[2]); worst_o=max(worst_o,eo)
    print(k,'c',kw['c'],'vs ref',e,'vs oracle',eo,'status',st)
print('worst vs ref',worst,'vs oracle',worst_o)
g6=load_golden('twocol_so')
m=configs.twocol_so_member(nz=100,ny=40)
cfg=dict(m,kappa=m['kappa'][None],b_basin0=m['b_basin0'][None],b_north0=m['b_north0'][None],bs_SO=m['bs_SO'][None],bvp_refine=-1)
ens=gpu.TwoColEnsemble(cfg)
done=0; w=0
for s in (1,24,25,26,2400):
    ens.run(s-done); done=s
    st=ens.state()
    for kk in ('b_basin','b_north','Psi','Psi_SO'):
        w=max(w,relerr(st[kk][0],g6['s%05d_%s'%(s,kk)]))
print('G6 trajectory adaptive on GPU: worst',w)
for refine in (8,-1):
    c=dict(configs.config4(N=8192),bvp_refine=refine)
    e=gpu.TwoColEnsemble(c); e.run(241); gpu.synchronize()
    t0=time.perf_counter(); e.run(2400); gpu.synchronize(); el=time.perf_counter()-t0
    print('config4 refine',refine,'coupled steps/s %.3e'%(8192*2400/el),'nonfinite',e.nonfinite_members().size,'status bits',np.bincount(e.so.status.download()&8)[1:] )
