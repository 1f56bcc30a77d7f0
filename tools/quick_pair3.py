import time, numpy as np, sys, torch
sys.path.insert(0,'.')
import bls12_381_amd as b
ctx = b.default_context()
n=1<<16
r = np.random.RandomState(1)
a = r.randint(0,256,size=(n,32),dtype=np.uint8); a[:,31]&=0x3f
bb = r.randint(0,256,size=(n,32),dtype=np.uint8); bb[:,31]&=0x3f
g1,f1 = ctx.bases_from_scalars(1,a).download(); g2,f2 = ctx.bases_from_scalars(2,bb).download()
dev=torch.device('cuda',0)
d_g1=torch.from_numpy(g1.view(np.int64)).to(dev); d_g2=torch.from_numpy(g2.view(np.int64)).to(dev); d_gt=torch.zeros((n,72),dtype=torch.int64,device=dev)
for rep in range(3):
    b._lib.check(ctx.lib.blsgpu_pairing_batch_device(ctx.h, d_g1.data_ptr(), None, d_g2.data_ptr(), None, n, d_gt.data_ptr()),"p")
    ctx.synchronize()
