#!/usr/bin/env python3
"""Occupancy of the chip over ONE launch of the G1 bucket accumulation (2^20 points, default path).

   python tools/acc_trace.py build      here: a second library, build/libblsgpu_trace.so, whose api_msm unit is compiled with -DBLS_ACC_TRACE
                                        (every wavefront of k_msm_accumulate records start, end, hardware id and item length); the product
                                        library is not touched
   python tools/acc_trace.py run        on the GPU box: one traced call, summary on stdout, gpurun_out/acc_trace.json

   What it answers: the kernel issues 0.80 of the multiply-add rate; is the rest idle SIMDs (imbalance, tail) or stalls of resident waves?"""
import ctypes
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE_LIB = os.path.join(ROOT, "build", "libblsgpu_trace.so")


def build():
    import __graft_entry__ as g
    g.build()
    obj = os.path.join(ROOT, "build", "trace_api_msm.o")
    extra = os.environ.get("ACC_TRACE_FLAGS", "").split()
    subprocess.check_call(["hipcc"] + g.HIPCC_FLAGS + ["-DBLS_ACC_TRACE"] + extra + ["-I" + os.path.join(ROOT, "include"), "-c",
                           os.path.join(ROOT, "bls12_381_amd", "csrc", "api_msm.hip"), "-o", obj], cwd=ROOT)
    objs = [obj if u == "api_msm" else os.path.join(g.OBJ_DIR, u + ".o") for u in g.UNITS]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", TRACE_LIB])
    print("built", TRACE_LIB)


def run():
    os.environ["BLSGPU_LIB_PATH"] = TRACE_LIB
    import numpy as np
    import torch
    import bls12_381_amd as bls
    from bls12_381_amd import synthetic
    n = 1 << 20
    ctx = bls.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    bases = ctx.bases_from_scalars(1, synthetic.scalars(n, synthetic.SEED + 1))
    d_s = torch.from_numpy(synthetic.scalars(n, synthetic.SEED)).cuda()
    d_o = torch.zeros(18, dtype=torch.int64, device="cuda")
    call = lambda: ctx.msm_device(bases, d_s.data_ptr(), n, d_o.data_ptr())
    for _ in range(3):
        call(); torch.cuda.synchronize()
    nw = 1 << 14
    tr = torch.zeros(nw * 6 + 64 * 256, dtype=torch.int64, device="cuda")
    setp = ctx.lib._handle if hasattr(ctx.lib, "_handle") else None
    lib = ctypes.CDLL(TRACE_LIB)
    lib.blsgpu_diag_set_acc_trace.argtypes = [ctypes.c_void_p]
    lib.blsgpu_diag_set_acc_trace.restype = ctypes.c_int
    assert lib.blsgpu_diag_set_acc_trace(tr.data_ptr()) == 0
    torch.cuda.synchronize()
    call(); torch.cuda.synchronize()
    assert lib.blsgpu_diag_set_acc_trace(None) == 0
    full = tr.cpu().numpy()
    t = full[:nw * 6].reshape(nw, 6)
    samples = full[nw * 6:].reshape(64, 256)
    np.save(os.path.join(ROOT, "gpurun_out", "acc_trace_samples.npy"), samples)
    t = t[t[:, 1] != 0]
    t0, t1, hw, ln = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2].astype(np.uint64), (t[:, 3] & 0xffffffff).astype(np.int64)
    base = t0.min()
    tick_ns = 10.0                                           # the constant-rate clock of wall_clock64: 100 MHz
    s = (t0 - base) * tick_ns * 1e-3                         # us
    e = (t1 - base) * tick_ns * 1e-3
    span = e.max()
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
    simd = (hwid >> 4) & 3
    cu = (hwid >> 8) & 15
    sh = (hwid >> 12) & 1
    se = (hwid >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    simds = np.unique(key)
    # resident waves over time
    grid = np.linspace(0, span, 401)
    resident = [(int(((s <= g) & (e > g)).sum())) for g in grid]
    # per SIMD: time with 2 / 1 / 0 waves
    two = one = 0.0
    ends = []
    for k in simds:
        m = key == k
        ev = sorted([(x, 1) for x in s[m]] + [(x, -1) for x in e[m]])
        cur = 0; last = 0.0
        for x, d in ev:
            if cur >= 2: two += x - last
            elif cur == 1: one += x - last
            last = x; cur += d
        ends.append(e[m].max())
    tot = span * len(simds)
    ends = np.array(ends)
    dur = e - s
    mhz = (t[:, 5] - t[:, 4]) / np.maximum(dur, 1e-3)        # shader cycles per us over each wave's life
    out = {
        "waves": int(len(t)), "simds_seen": int(len(simds)), "span_us": round(float(span), 1),
        "frac_simd_time_two_waves": round(two / tot, 4), "frac_simd_time_one_wave": round(one / tot, 4),
        "frac_simd_time_idle": round(1 - (two + one) / tot, 4),
        "simd_last_end_us": {"min": round(float(ends.min()), 1), "p10": round(float(np.percentile(ends, 10)), 1), "median": round(float(np.median(ends)), 1),
                             "p90": round(float(np.percentile(ends, 90)), 1), "max": round(float(ends.max()), 1)},
        "wave_start_us": {"p50": round(float(np.median(s)), 1), "p90": round(float(np.percentile(s, 90)), 1), "max": round(float(s.max()), 1)},
        "wave_duration_us": {"min": round(float(dur.min()), 1), "median": round(float(np.median(dur)), 1), "max": round(float(dur.max()), 1)},
        "us_per_entry_by_start_order": [round(float(np.median((dur / np.maximum(ln, 1))[i::8])), 3) for i in range(8)],
        "shader_mhz_by_start_order": [round(float(np.median(mhz[(dur > 50)][i::8])), 1) for i in range(8)],
        "shader_mhz_first_and_last_waves": [round(float(np.median(mhz[:256])), 1), round(float(np.median(mhz[-1024:-512])), 1)],
        "item_len": {"min": int(ln.min()), "median": int(np.median(ln)), "max": int(ln.max())},
        "resident_waves_over_time": resident[::10],
        "waves_per_simd": {"min": int(min((key == k).sum() for k in simds)), "max": int(max((key == k).sum() for k in simds))},
    }
    # speed of a wave alone on its SIMD vs with a neighbour: us per entry against the overlap fraction
    print(json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "acc_trace.npy"), np.stack([s, e, key.astype(np.float64), ln.astype(np.float64), mhz], 1))
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "acc_trace.json"), "w"), indent=1)


def ramp():
    """the timed region of bench.py as the driver runs it (5 warm-up steps, sync, K pipelined steps, join, sync) with every accumulation launch's start,
    end and shader clock recorded: where do the ~4 ms go that a 20-step region costs beyond 20 steady-state steps?"""
    os.environ["BLSGPU_LIB_PATH"] = TRACE_LIB
    import time
    import numpy as np
    import torch
    import bls12_381_amd as bls
    from bls12_381_amd import synthetic
    n = 1 << 20
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    ctx = bls.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    bases = ctx.bases_from_scalars(1, synthetic.scalars(n, synthetic.SEED + 1))
    d_s = torch.from_numpy(synthetic.scalars(n, synthetic.SEED)).cuda()
    d_o = torch.zeros((8, 18), dtype=torch.int64, device="cuda")
    call = lambda i: ctx.msm_device(bases, d_s.data_ptr(), n, d_o[i % 8].data_ptr())
    lib = ctypes.CDLL(TRACE_LIB)
    lib.blsgpu_diag_set_acc_trace.argtypes = [ctypes.c_void_p]
    lib.blsgpu_diag_set_acc_trace_seq.argtypes = [ctypes.c_uint]
    cap = 1 << 18
    tr = torch.zeros(cap * 6, dtype=torch.int64, device="cuda")
    ctx.set_pipelining(True)
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    assert lib.blsgpu_diag_set_acc_trace(tr.data_ptr()) == 0 and lib.blsgpu_diag_set_acc_trace_seq(1) == 0
    lib.blsgpu_diag_set_acc_trace_seq(0)              # symbols set before the warm-up: hipMemcpyToSymbol afterwards would be a longer idle gap than the driver's fence
    for i in range(W):
        call(i)
    ctx.join(); torch.cuda.synchronize()
    lib.blsgpu_diag_set_acc_trace_seq(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        call(i)
    ctx.join(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.blsgpu_diag_set_acc_trace_seq(0); lib.blsgpu_diag_set_acc_trace(None)
    t = tr.cpu().numpy().reshape(cap, 6)
    t = t[t[:, 1] != 0]
    order = np.argsort(t[:, 0])
    t = t[order]
    s = (t[:, 0] - t[0, 0]) * 0.01
    e = (t[:, 1] - t[0, 0]) * 0.01
    mhz = (t[:, 5] - t[:, 4]) / np.maximum(e - s, 1e-3)
    # launches: consecutive calls use different pipeline slots (the kernel records its slot's control-block address); within a slot, launches are
    # separated by more than 5 ms
    slot = (t[:, 3] >> 32)
    rows = []
    for sid in np.unique(slot):
        m = np.where(slot == sid)[0]
        m = m[np.argsort(s[m])]
        cut = [0] + [k + 1 for k in range(len(m) - 1) if s[m[k + 1]] - s[m[k]] > 5000] + [len(m)]
        for a, b in zip(cut[:-1], cut[1:]):
            idx = m[a:b]
            if len(idx) < 1000:
                continue
            ll = idx[(e - s)[idx] > 100]
            rows.append({"waves": int(len(idx)), "start_us": round(float(s[idx].min()), 1), "end_us": round(float(e[idx].max()), 1),
                         "dur_us": round(float(e[idx].max() - s[idx].min()), 1), "mhz": round(float(np.median(mhz[ll])), 0)})
    rows.sort(key=lambda r: r["start_us"])
    for k, r in enumerate(rows):
        r["since_prev_start_us"] = round(r["start_us"] - rows[k - 1]["start_us"], 1) if k else 0.0
    bins = {}
    long_lived = (e - s) > 100
    for x, m in zip(s[long_lived], mhz[long_lived]):
        bins.setdefault(int(x // 3000), []).append(m)
    clock_by_3ms = [round(float(np.median(bins[k])), 0) for k in sorted(bins)]
    out = {"steps": K, "warmup": W, "timed_region_ms": round(1e3 * dt, 3), "ms_per_step": round(1e3 * dt / K, 4), "shader_mhz_by_3ms_of_the_region": clock_by_3ms, "launches": rows}
    print(json.dumps(out))
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "acc_ramp.json"), "w"), indent=1)


if __name__ == "__main__":
    {"build": build, "run": run, "ramp": ramp}[sys.argv[1]]()
