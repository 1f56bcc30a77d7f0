import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bls12_381_amd as bls
dev = torch.device("cuda", 0)
os.environ["BLSGPU_NTT_IMPL"] = "stage"; old = bls.Context(0); os.environ.pop("BLSGPU_NTT_IMPL")
new = bls.Context(0)
for c in (old, new): c.set_stream(torch.cuda.current_stream().cuda_stream)
for log_n in (15, 16, 20):
    n = 1 << log_n
    raw = np.random.RandomState(log_n).randint(0, 256, size=(n, 32), dtype=np.uint8); raw[:, 31] &= 0x3F
    x = torch.from_numpy(raw.view(np.int64).reshape(n, 4).copy()).to(dev)
    y = x.clone(); old.fr_ntt_device(y.data_ptr(), log_n); torch.cuda.synchronize()
    prev = {}
    for trial in range(3):
        for shape in (None, "12,10,1024", "10,5,256", "11,6,256", "10,3,256", "10,2,256", "10,1,256"):
            if shape: os.environ["BLSGPU_NTT_COLS"] = shape
            a = x.clone(); torch.cuda.synchronize(); new.fr_ntt_device(a.data_ptr(), log_n); torch.cuda.synchronize()
            os.environ.pop("BLSGPU_NTT_COLS", None)
            bad = (a != y).any(dim=1).nonzero().flatten()
            rep = None if shape not in prev else bool(torch.equal(prev[shape], a))
            prev[shape] = a.clone()
            print(log_n, trial, shape, "bad rows", bad.numel(), "same as previous run of this shape:", rep, flush=True)
