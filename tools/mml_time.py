#!/usr/bin/env python3
"""A/B timing of the shared-accumulator Miller kernels (round 5): the 2^18-term product of BASELINE configs[4] on the legacy
lane-pair kernel (k_multi_miller_shared) and the quad kernel (k_mml_prep_quad; its lane-pair twin: tools/experiments/mml_pair.hip.h), for
K = 2, 4, 8 terms per accumulator; the same product with every term prepared; three-term verification equations with two prepared
terms on both layouts.  Every variant's result is compared with the first one's (limb-identical or the run fails).
   usage: python tools/mml_time.py [log2 terms = 18]"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bls12_381_amd as bls

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 18
dev = torch.device("cuda", 0)
base = 1 << 16
rs = np.random.RandomState(99)
ka = rs.randint(0, 256, size=(base, 32), dtype=np.uint8); ka[:, 31] &= 0x3F
kq = rs.randint(0, 256, size=(base, 32), dtype=np.uint8); kq[:, 31] &= 0x3F
c0 = bls.Context(0)
g1xy, _ = c0.bases_from_scalars(1, ka).download()
g2xy, _ = c0.bases_from_scalars(2, kq).download()
n = 1 << logn
rep = max(1, n // base)
d_g1 = torch.from_numpy(g1xy.view(np.int64)).to(dev).repeat(rep, 1)[:n].contiguous()
d_g2 = torch.from_numpy(g2xy.view(np.int64)).to(dev).repeat(rep, 1)[:n].contiguous()
sync = torch.cuda.synchronize


def med(fn, reps=3):
    fn(); sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); sync(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))


def make_ctx(**env):
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        c = bls.Context(0)
    finally:
        for k in env:
            os.environ.pop(k)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    return c


out = {"terms": n}
d_res = torch.zeros(72, dtype=torch.int64, device=dev)
ref = None
rows = []
for impl, name in ((1, "k_multi_miller_shared"), (4, "k_mml_prep_quad")):
    for K in ((4,) if os.environ.get("MML_TIME_QUICK") else (2, 4, 8)):
        c = make_ctx(BLSGPU_MML_IMPL=impl, BLSGPU_MMLP_K=K)
        ms = med(lambda: c.multi_miller_loop_device(d_g1.data_ptr(), d_g2.data_ptr(), n, d_res.data_ptr()))
        got = d_res.cpu().numpy().copy()
        if ref is None:
            ref = got
        assert np.array_equal(got, ref), (name, K)
        rows.append({"kernel": name, "K": K, "ms": round(ms, 3)})
        print("unprepared", name, "K =", K, "%.3f ms" % ms, flush=True)
        c.close()
out["unprepared_product"] = rows
# every term prepared: a table of four points, indices cycling
key = g2xy[:4].copy()
d_qi = torch.from_numpy((np.arange(n) % 4).astype(np.uint32).view(np.int32)).to(dev)
d_gk = torch.from_numpy(key.view(np.int64)).to(dev)[(torch.arange(n, device=dev) % 4)].contiguous()
rows = []
ref = None
for lay in ("quad",):
    for K in (2, 4, 8):
        c = make_ctx(BLSGPU_MMLP_K=K)
        table = c.g2_prepare(key)
        ms = med(lambda: c.multi_miller_loop_prepared_device(d_g1.data_ptr(), table, d_qi.data_ptr(), n, d_res.data_ptr()))
        got = d_res.cpu().numpy().copy()
        if ref is None:
            c.multi_miller_loop_device(d_g1.data_ptr(), d_gk.data_ptr(), n, d_res.data_ptr()); sync()
            ref = d_res.cpu().numpy().copy()
        ok = bool(np.array_equal(got, ref))
        rows.append({"layout": lay, "K": K, "ms": round(ms, 3), "matches_unprepared": ok})
        print("all prepared", lay, "K =", K, "%.3f ms" % ms, "OK" if ok else "MISMATCH", flush=True)
        table.free(); c.close()
out["prepared_product"] = rows
# three-term equations, two prepared
rows = []
for le in (14, 16):
    ne, ke = 1 << le, 3
    m = ne * ke
    reps_ = max(1, (m + base - 1) // base)
    e_g1 = torch.from_numpy(g1xy.view(np.int64)).to(dev).repeat(reps_, 1)[:m].contiguous()
    e_g2 = torch.from_numpy(g2xy.view(np.int64)).to(dev).repeat(reps_, 1)[:m].contiguous()
    qi = np.full(m, bls.UNPREPARED, dtype=np.uint32); qi[1::3] = 0; qi[2::3] = 1
    e_qi = torch.from_numpy(qi.view(np.int32)).to(dev)
    e_off = torch.arange(0, (ne + 1) * ke, ke, dtype=torch.int64, device=dev)
    e_out = torch.zeros((ne, 72), dtype=torch.int64, device=dev)
    ref = None
    for lay in ("quad",):
        for fe in (False, True):
            c = make_ctx()
            table = c.g2_prepare(key[:2])
            ms = med(lambda: c.multi_miller_loop_prepared_many_device(e_g1.data_ptr(), table, e_qi.data_ptr(), e_off.data_ptr(), ne, m, e_out.data_ptr(), max_seg_terms=ke, final_exp=fe,
                                                                      d_g2=e_g2.data_ptr()))
            if fe:
                got = e_out[:256].cpu().numpy().copy()
                if ref is None:
                    ref = got
                if not np.array_equal(got, ref):
                    print("MISMATCH between layouts", lay, le, flush=True)
            rows.append({"log2_equations": le, "layout": lay, "final_exp": fe, "ms": round(ms, 3)})
            print("equations 2^%d" % le, lay, "final_exp" if fe else "miller only", "%.3f ms" % ms, flush=True)
            table.free(); c.close()
out["prepared_equations"] = rows
print(json.dumps(out))
