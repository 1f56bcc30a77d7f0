import time, numpy as np, sys, torch
sys.path.insert(0,'.')
import bls12_381_amd as b
ctx = b.default_context()
dev=torch.device('cuda',0)
for log_n in (16, 20, 22, 24):
    n=1<<log_n
    rs=np.random.RandomState(1)
    raw = rs.randint(0,256,size=(n,32),dtype=np.uint8); raw[:,31]&=0x3f
    d = torch.from_numpy(raw.view(np.int64).reshape(n,4).copy()).to(dev)
    for inv in (False, True):
        ctx.fr_ntt_device(d.data_ptr(), log_n, inv); ctx.synchronize()
        torch.cuda.synchronize(); t0=time.time()
        reps=10
        for _ in range(reps): ctx.fr_ntt_device(d.data_ptr(), log_n, inv)
        ctx.synchronize(); t1=time.time()
        ms=(t1-t0)/reps*1e3
        print(f"fr_ntt 2^{log_n} inverse={inv}: {ms:.3f} ms  ({n*32*2/ms/1e6:.1f} GB/s of in+out, {n/ms/1e3:.1f} M elems/s)")
