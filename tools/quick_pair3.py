import time, numpy as np, sys, torch
sys.path.insert(0,'.')
import bls12_381_amd as b
ctx = b.default_context()
n=1<<17
r = np.random.RandomState(1)
a = r.randint(0,256,size=(n,32),dtype=np.uint8); a[:,31]&=0x3f
bb = r.randint(0,256,size=(n,32),dtype=np.uint8); bb[:,31]&=0x3f
g1,f1 = ctx.bases_from_scalars(1,a).download(); g2,f2 = ctx.bases_from_scalars(2,bb).download()
dev=torch.device('cuda',0)
d_g1=torch.from_numpy(g1.view(np.int64)).to(dev); d_g2=torch.from_numpy(g2.view(np.int64)).to(dev); d_gt=torch.zeros((n,72),dtype=torch.int64,device=dev)
def T(fn, reps=2):
    for _ in range(reps):
        torch.cuda.synchronize(); t0=time.time(); fn(); ctx.synchronize(); t1=time.time()
    return 1e3*(t1-t0)
tp = T(lambda: b._lib.check(ctx.lib.blsgpu_pairing_batch_device(ctx.h, d_g1.data_ptr(), None, d_g2.data_ptr(), None, n, d_gt.data_ptr()),"p"))
ml = ctx.miller_loop_batch(g1,f1,g2,f2)
tm = T(lambda: ctx.miller_loop_batch(g1,f1,g2,f2))
tf = T(lambda: ctx.final_exponentiation_batch(ml))
print(f"2^16: pairing(device) {tp:.1f} ms = {n/tp*1e3:.0f}/s | miller(host io) {tm:.1f} ms | final_exp(host io) {tf:.1f} ms")
