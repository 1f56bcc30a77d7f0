import time, numpy as np, sys, torch
sys.path.insert(0,'.')
import bls12_381_amd as b
ctx = b.default_context()
dev=torch.device('cuda',0)
r = np.random.RandomState(3)
for logn in (18,20):
    n=1<<logn
    kb = r.randint(0,256,size=(n,32),dtype=np.uint8); kb[:,31]&=0x3f
    sb = r.randint(0,256,size=(n,32),dtype=np.uint8); sb[:,31]&=0x3f
    bases = ctx.bases_from_scalars(2,kb)
    ctx.set_profiling(True); ctx.msm(bases,sb); ctx.msm(bases,sb)
    print(f"G2 2^{logn} phases", {k: round(v,2) for k,v in ctx.last_msm_phase_ms().items()})
    ctx.set_profiling(False)
    d_s=torch.from_numpy(sb).to(dev); outs=[torch.zeros(36,dtype=torch.int64,device=dev) for _ in range(4)]
    ctx.set_pipelining(True)
    for i in range(3): ctx.msm_device(bases,d_s.data_ptr(),n,outs[i&3].data_ptr())
    ctx.join(0); torch.cuda.synchronize(); t0=time.time()
    K=10
    for i in range(K): ctx.msm_device(bases,d_s.data_ptr(),n,outs[i&3].data_ptr())
    ctx.join(0); torch.cuda.synchronize(); dt=(time.time()-t0)/K
    ctx.set_pipelining(False)
    print(f"G2 2^{logn} pipelined {1e3*dt:.2f} ms/MSM -> {n/dt/1e6:.1f} M scalar-muls/s")
