import time, numpy as np, sys, torch
sys.path.insert(0,'.')
import bls12_381_amd as b
ctx = b.default_context()
dev=torch.device('cuda',0)
n=1<<20
rs=np.random.RandomState(1)
kb = rs.randint(0,256,size=(n,32),dtype=np.uint8); kb[:,31]&=0x3f
sb = rs.randint(0,256,size=(n,32),dtype=np.uint8); sb[:,31]&=0x3f
bases = ctx.bases_from_scalars(2,kb)
d_s = torch.from_numpy(sb).to(dev)
d_o = [torch.zeros(36,dtype=torch.int64,device=dev) for _ in range(4)]
ctx.set_pipelining(True)
for i in range(2): ctx.msm_device(bases, d_s.data_ptr(), n, d_o[i].data_ptr())
ctx.join(0); torch.cuda.synchronize()
t0=time.time()
for i in range(8): ctx.msm_device(bases, d_s.data_ptr(), n, d_o[i&3].data_ptr())
ctx.join(0); torch.cuda.synchronize()
print("G2 msm 2^20 pipelined ms/step", (time.time()-t0)/8*1e3)
ctx.set_pipelining(False)
ctx.set_profiling(True)
ctx.msm_device(bases, d_s.data_ptr(), n, d_o[0].data_ptr()); ctx.synchronize()
print({k: round(v,2) for k,v in ctx.last_msm_phase_ms().items()})
