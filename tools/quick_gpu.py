import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import openlte_amd as m
from oracle import pyoracle as po
import lte_testdata as td
P=po.port(); ctx=m.Context(0); print(ctx.device_name)
for K in [40,512,6144]:
    tx,soft=td.turbo_blocks(P,K,66,"awgn0.5",1)
    want=td.oracle_turbo_ref(P,soft,K); got=ctx.turbo_decode(soft,K)
    print(K,"mismatching blocks:",int((got!=want).any(axis=1).sum()))
K=6144; n_cb=65536
tx,soft=td.turbo_blocks(P,K,16,"hard127",3)
big=soft[np.arange(n_cb)%16]
d_in=ctx.to_device(big); d_out=ctx.alloc(n_cb*K)
for it in range(3):
    ctx.timer_start(); ctx.turbo_decode_dev(d_in,m.SOFT_I8,K,n_cb,d_out); ms=ctx.timer_stop()
    print("K=6144 n_cb=65536: %.3f ms -> %.1f Mbit/s, %.2f M CB/s"%(ms, K*n_cb/ms/1e3, n_cb/ms/1e3))
