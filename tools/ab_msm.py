#!/usr/bin/env python3
"""A/B of the G1 / G2 MSM for alternative builds / environment switches:  python tools/ab_msm.py [--log-n 20] [--group 2] name=ENV=1,ENV2=x[@lib.so] ...
Each variant runs in its own subprocess; prints pipelined ms/step, accumulate ms, phases and single-call latency."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(log_n, group=1):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import bls12_381_amd as bls
    from bls12_381_amd import synthetic
    dev = torch.device("cuda", 0)
    ctx = bls.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 1 << log_n
    kb = synthetic.scalars(n, 1); sb = synthetic.scalars(n, 2)
    bases = ctx.bases_from_scalars(group, kb)
    d_s = torch.from_numpy(sb).to(dev)
    d_o = [torch.zeros(18 * group, dtype=torch.int64, device=dev) for _ in range(4)]
    ctx.set_pipelining(True)
    for i in range(5):
        ctx.msm_device(bases, d_s.data_ptr(), n, d_o[i & 3].data_ptr())
    ctx.join(0); torch.cuda.synchronize()
    if not os.environ.get('AB_NO_STATS'):
        ctx.msm_accumulate_stats(True)
    K = 60
    t = time.perf_counter()
    for i in range(K):
        ctx.msm_device(bases, d_s.data_ptr(), n, d_o[i & 3].data_ptr())
    ctx.join(0); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    acc_ms = 0.0
    if not os.environ.get('AB_NO_STATS'):
        acc_ms, _ = ctx.msm_accumulate_stats(False)
    ctx.set_pipelining(False)
    ts = []
    for i in range(8):
        t = time.perf_counter(); ctx.msm_device(bases, d_s.data_ptr(), n, d_o[0].data_ptr()); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    ctx.set_profiling(True)
    ctx.msm_device(bases, d_s.data_ptr(), n, d_o[0].data_ptr())
    ph = ctx.last_msm_phase_ms()
    ctx.set_profiling(False)
    aff = ctx.batch_normalize(group, d_o[0].cpu().numpy().view(np.uint64)[None, :])[0][0]
    import hashlib
    print(json.dumps({"ms_per_step": 1e3 * dt, "acc_ms_in_pipeline": acc_ms, "single_call_ms": 1e3 * float(np.median(ts[2:])),
                      "phases": {k: round(v, 3) for k, v in ph.items()}, "result_sha": hashlib.sha256(aff.tobytes()).hexdigest()[:12]}))


if __name__ == "__main__":
    if os.environ.get("AB_CHILD"):
        child(int(os.environ["AB_LOG_N"]), int(os.environ.get("AB_GROUP", "1")))
    else:
        args = sys.argv[1:]
        log_n = 20
        group = 1
        while args and args[0] in ("--log-n", "--group"):
            if args[0] == "--log-n":
                log_n = int(args[1])
            else:
                group = int(args[1])
            args = args[2:]
        for spec in args:
            name, _, rest = spec.partition("=")
            envs, _, lib = rest.partition("@")
            env = dict(os.environ, AB_CHILD="1", AB_LOG_N=str(log_n), AB_GROUP=str(group))
            for kv in filter(None, envs.split(",")):
                k, _, v = kv.partition("=")
                env[k] = v
            if lib:
                env["BLSGPU_LIB_PATH"] = os.path.abspath(lib)
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(name, line[-1] if line else ("FAILED: " + r.stderr[-1500:]))
