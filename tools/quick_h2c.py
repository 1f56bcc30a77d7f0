import time, numpy as np, sys, torch
sys.path.insert(0,'.')
import bls12_381_amd as b
ctx = b.default_context()
dev=torch.device('cuda',0)
n=1<<18; L=32
rs=np.random.RandomState(3)
msgs=torch.from_numpy(rs.randint(0,256,size=n*L,dtype=np.uint8)).to(dev)
offs=torch.arange(0,(n+1)*L,L,dtype=torch.int64,device=dev)
dst=b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_"
d_dst=torch.from_numpy(np.frombuffer(dst,dtype=np.uint8).copy()).to(dev)
for group in (1,2):
    out=torch.zeros((n,18*group),dtype=torch.int64,device=dev)
    for m in (1<<16, 1<<18):
        for rep in range(2):
            torch.cuda.synchronize(); t0=time.time()
            b._lib.check(ctx.lib.blsgpu_hash_to_curve_device(ctx.h, group, msgs.data_ptr(), offs.data_ptr(), m, d_dst.data_ptr(), len(dst), 0, out.data_ptr()),"h2c")
            ctx.synchronize(); t1=time.time()
        print(f"hash_to_curve G{group} n={m} ({L}-byte messages): {1e3*(t1-t0):.2f} ms -> {m/(t1-t0):.0f} hashes/s")
