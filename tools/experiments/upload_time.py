import time, numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
import bls12_381_amd as b
from bls12_381_amd import synthetic as sy
ctx=b.Context(0)
n=1<<20
k=sy.scalars(n, 5)
bases=ctx.bases_from_scalars(1,k)
xy,inf=bases.download()
for assume in (1,0,0):
    ctx.set_assume_subgroup(assume)
    t=time.perf_counter(); bb=ctx.upload_bases(1, xy, inf); st=bb.subgroup_state; t=time.perf_counter()-t
    print("assume",assume,"upload 2^20 G1 ms", round(1e3*t,2), "state", st)
