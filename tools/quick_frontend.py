import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import openlte_amd as m
from openlte_amd import synth
from oracle import pyoracle as po
import lte_testdata as td
P=po.port(); ctx=m.Context(0)
cfg=m.DlCfg(2048,100,1,0)
iq,_=synth.dl_units(cfg,[1],[17],td.small_allocs(0,100,3,2024,8),1,snr_db=30,max_delay=5,seed=1)
got=ctx.dl_frontend(cfg,iq.reshape(-1,2),[0],[1],[17])
_,s=td.oracle_frontend(P,2048,100,1,iq[0],1,17)
def rl(a,b): return np.linalg.norm((a-b).ravel())/np.linalg.norm(b.ravel())
print("symb re",rl(got[0,0],s.arr("rx_symb_re")),"im",rl(got[0,1],s.arr("rx_symb_im")))
print("ce re",rl(got[0,2,:14],s.arr("rx_ce_re")[0,:14]),"im",rl(got[0,3,:14],s.arr("rx_ce_im")[0,:14]))
n=10000
U=64
sfs=np.arange(U)%10; cells=(np.arange(U)*37)%504
allocs=[]
for u in range(U): allocs+=td.w4_allocs(u)
iq,_=synth.dl_units(cfg,sfs,cells,allocs,9,seed=2)
ul=iq.shape[1]
idx=np.arange(n)%U
d_iq=ctx.to_device(iq[idx].reshape(-1,2)); d_st=ctx.to_device((np.arange(n)*ul).astype(np.uint64))
d_sf=ctx.to_device(sfs[idx].astype(np.uint32)); d_cell=ctx.to_device(cells[idx].astype(np.uint32))
d_out=ctx.alloc(n*ctx.subframe_floats(1)*4)
ctx.profile(True)
for it in range(3):
    ctx.timer_start(); ctx.dl_frontend_dev(cfg,d_iq,None,d_st,d_sf,d_cell,n,d_out); ms=ctx.timer_stop()
    print("frontend 10k subframes: %.3f ms -> %.0f subframes/s, %.1f GB/s algorithmic"%(ms,n/ms*1e3,339040*n/ms/1e6))
print(ctx.profile_report())
