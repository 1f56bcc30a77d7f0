#!/usr/bin/env python3
"""Run ONE secondary leg of bench.py a few times on the bench's own synthetic inputs: profiling target for tools/leg_traffic.py.
   usage: python tools/run_leg.py <h2c_g1|h2c_g2|decode_g1|decode_g2|g2_msm|mul_g1|mul_g2|ntt> [reps = 3]
   Prints `LEG_KERNEL_WINDOW <n>`: the number of calls made, so that the summariser can divide the counters of the run by it; set-up kernels
   (building the inputs) are named in SETUP_KERNELS and excluded there."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bls12_381_amd as bls
from bls12_381_amd import synthetic

leg = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
ctx = bls.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
sync = torch.cuda.synchronize
P = lambda t: t.data_ptr()
n16 = 1 << 16
if leg in ("h2c_g1", "h2c_g2"):
    g = 1 if leg == "h2c_g1" else 2
    hm = torch.from_numpy(np.random.RandomState(99).randint(0, 256, size=n16 * 32, dtype=np.uint8)).to(dev)
    ho = torch.arange(0, (n16 + 1) * 32, 32, dtype=torch.int64, device=dev)
    dst = b"BLS_SIG_BLS12381G%d_XMD:SHA-256_SSWU_RO_NUL_" % g
    hd = torch.from_numpy(np.frombuffer(dst, dtype=np.uint8).copy()).to(dev)
    out = torch.zeros((n16, 18 * g), dtype=torch.int64, device=dev)
    call = lambda: bls._lib.check(ctx.lib.blsgpu_hash_to_curve_device(ctx.h, g, P(hm), P(ho), n16, P(hd), len(dst), 0, P(out)), "h2c")
elif leg in ("decode_g1", "decode_g2"):
    g = 1 if leg == "decode_g1" else 2
    xy, _ = ctx.bases_from_scalars(g, synthetic.scalars(n16, 99 if g == 1 else 100)).download()
    enc = torch.from_numpy(ctx.points_to_bytes(g, xy, None, compressed=True)).to(dev)
    d_cx = torch.zeros((n16, 12 * g), dtype=torch.int64, device=dev); d_ci = torch.zeros(n16, dtype=torch.uint8, device=dev); d_ck = torch.zeros(n16, dtype=torch.uint8, device=dev)
    call = lambda: ctx.points_from_bytes_device(g, P(enc), n16, P(d_cx), P(d_ci), P(d_ck), compressed=True, checked=True)
elif leg == "g2_msm":
    n = 1 << 20
    b2 = ctx.bases_from_scalars(2, synthetic.scalars(n, 101))
    d_s = torch.from_numpy(synthetic.scalars(n, synthetic.SEED)).to(dev)
    d_o = torch.zeros(36, dtype=torch.int64, device=dev)
    call = lambda: ctx.msm_device(b2, P(d_s), n, P(d_o))
elif leg in ("mul_g1", "mul_g2"):
    g = 1 if leg == "mul_g1" else 2
    n = 1 << (20 if g == 1 else 18)
    xy, _ = ctx.bases_from_scalars(g, synthetic.scalars(n, synthetic.SEED + 1 if g == 1 else 101)).download()
    d_xy = torch.from_numpy(xy.view(np.int64)).to(dev)
    d_s = torch.from_numpy(synthetic.scalars(n, synthetic.SEED)).to(dev)
    d_o = torch.zeros((n, 18 * g), dtype=torch.int64, device=dev)
    call = lambda: ctx.mul_batch_device(g, P(d_xy), 0, P(d_s), n, P(d_o))
elif leg == "ntt":
    n = 1 << 20
    d = torch.from_numpy(synthetic.scalars(n, synthetic.SEED)).to(dev)
    ctx.fr_ntt_device(P(d), 20, False); sync()          # the twiddle tables are built by the first call
    call = lambda: ctx.fr_ntt_device(P(d), 20, False)
else:
    raise SystemExit("unknown leg " + leg)
call(); sync()                                           # warm-up: scratch growth, tables
sync()
for _ in range(reps):
    call(); sync()
print("LEG_KERNEL_WINDOW", reps + 1, flush=True)
