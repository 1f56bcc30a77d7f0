#!/usr/bin/env python3
"""bench.py -- G1 MSM throughput (scalar-muls/s) on MI355X, the headline metric of BASELINE.json.

  python bench.py --gpus N --steps K --warmup W            (N > 1: re-executes itself under torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workloads (synthetic inputs per SURVEY.md 8d: SplitMix64 scalars uniform in [0, r), bases [k_i]G built on the device):

  N = 1   (default)   BASELINE configs[1]: a 2^20-point G1 MSM on one MI355X.  A step is ONE MSM over input that is
                      already resident in HBM (2^20 affine bases in the library's resident form, 2^20 32-byte scalars).
  N > 1   (default)   BASELINE configs[3]: ONE 2^24-point G1 MSM sharded N ways (2^24 / N points per GPU), ending with the
                      path's single exchange: an RCCL all-gather of the N partial sums (144 B each) + fold on every rank
                      (SURVEY.md 8e).  "scaling": "strong".   --weak keeps 2^--log-n points PER GPU instead.
  --workload mixed    BASELINE configs[4]: 2^22 G1 MSM + 2^22 G2 MSM + one 2^18-term multi_miller_loop (+ its final
                      exponentiation) per step, sharded N ways, the three jobs overlapped on separate stream sets (one
                      library context per job type), one small all-gather each.  Prints its own JSON line.

K steps are timed between barrier + synchronize pairs; the reported time is the max over ranks; `value` = scalar-muls of
all ranks / that time.  The JSON line also carries
  roofline        the dominant kernel (bucket accumulation) against the integer-VALU roofline: canonical MAC32 per launch /
                  HIP-event launch time inside the timed region; peak = v_mad_u64_u32 rate measured live
  single_call_ms / end_to_end_h2d_ms   latency of ONE non-pipelined MSM (scalars in HBM -> result in HBM) and of one MSM
                  whose scalars start in pinned host memory and whose result ends on the host (SURVEY.md 8d protocol:
                  3 warm-ups, median of 11)
  cpu_baseline    the C restatement of the reference's own `sum(P_i * s_i)` (oracle/bls_oracle.c) timed on the host cores
                  over a bounded sample of the same workload and checked against the GPU result, plus ns/op for the
                  reference's criterion points (benches/groups.rs)
  extras          BASELINE configs[2] (2^20 G2 MSM, 2^16 pairings), the 2^18-term multi_miller_loop, each with its
                  canonical roofline fraction (SURVEY.md 8d work units)
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WINDOW_BITS = 16          # canonical c of SURVEY.md 8d; the library picks its own c
# canonical work units: one Fp multiplication = 300 MAC32
MAC32_G1_ADD = 11 * 300                    # one complete mixed addition per (point, window)
MAC32_G1_MSM_2_20 = 188 * 300              # whole 2^20-point MSM, per scalar-mul
MAC32_G2_MSM_2_20 = 666 * 300              # per scalar-mul
MAC32_PAIRING = 16000 * 300
MAC32_MML_TERM = 6900 * 300
MAC32_G1_MUL_REF = 5100 * 300              # one `&G1Affine * &Scalar` as the reference computes it: 255 x (double 8 + add 12) field multiplications
MAC32_G2_MUL_REF = 17085 * 300             # the same over Fp2 (row a16)
MAC32_G1_MUL = 2852 * 300                  # ... as k_mul_batch computes it: 256 doublings x 8 + 67 complete additions x 12 (signed 4-bit windows); the roofline counts THIS
MAC32_G2_MUL = 9554 * 300                  # the same over Fp2 (2852 x 17085 / 5100)


def static_traffic(tag):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc summary of the SAME workload (profiles/<round>_<tag>_pmc.json,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE).  Counters cannot be read inside a timed
    run, so this is a STATIC figure from a separate profiled run -- labelled as such in the bench line, and marked STALE when the kernel sources
    of the leg have changed since the counters were collected (tools/srcdigest.py: the digest stored in the file against the sources being timed)."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (rnd, tag))
        if os.path.exists(path):
            try:
                j = json.load(open(path))
                src = "profiles/%s_%s_pmc.json (static: separate rocprofv3 --pmc passes of this workload, not measured in this run)" % (rnd, tag)
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    from srcdigest import source_digest
                    if j.get("source_digest") != source_digest(tag):
                        src += " -- STALE: the kernel sources of this leg changed since the counters were collected"
                except Exception:
                    pass
                return j["hbm_bytes_per_launch_corrected"], src
            except Exception:
                pass
    return None, None


def valu_issue(tag, seconds, peak):
    """MEASURED counterpart of the canonical fractions: VALU wave-instructions the leg executes (rocprofv3 SQ_INSTS_VALU of a committed counter run of
    the same workload: profiles/<round>_<tag>_pmc.json, static like `traffic`) x 64 lanes / the time measured here / the v_mad_u64_u32 issue rate measured
    here.  Cheap VOP2 instructions can issue faster than multiply-adds, so this is a utilisation of the multiply-add issue rate by ALL instructions, not a
    share of a hard ceiling; it says how much of a leg's time is instruction issue and how much is waiting."""
    for rnd in ("r06",):
        path = os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (rnd, tag))
        if os.path.exists(path):
            try:
                j = json.load(open(path))
                w = j.get("valu_wave_instructions_per_call") or (j.get("counters") or {}).get("SQ_INSTS_VALU")
                if w and seconds:
                    return {"valu_wave_instructions": w, "valu_issue_frac": w * 64 / seconds / peak, "valu_source": "profiles/%s_%s_pmc.json" % (rnd, tag)}
            except Exception:
                pass
    return {}


def kernel_trace_facts():
    """static, like `traffic`: what a per-wavefront trace of the accumulation kernel showed (tools/acc_trace.py, profiles/r06_acc_trace.md) -- the
    shader clock of a LONE call (s_memtime against the 100 MHz clock: after every idle moment the clock restarts near 2.0 GHz and climbs to ~2.29 GHz over
    ~35 ms of load, so `launch_ms_isolated` / `single_call_ms` sit at the bottom of that ramp while `peak` is probed near its top, and a 20-step timed region
    spends its first half climbing) and the share of SIMD-time with two / one / no resident wavefront.  With two resident the SIMD issues one VALU
    instruction per 4.0 cycles (its limit)."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "r06_acc_trace_default.json")))
        mhz = j["shader_mhz_by_start_order"]
        return {"kernel_trace": {"shader_mhz_lone_call": round(float(np.mean(mhz)), 0),
                                 "simd_time_two_one_no_wavefront": [j["frac_simd_time_two_waves"], j["frac_simd_time_one_wave"], j["frac_simd_time_idle"]],
                                 "source": "profiles/r06_acc_trace.md"}}
    except Exception:
        return {}


# ---- the ONE JSON line ----------------------------------------------------------------------------------------------------------
# The driver keeps the last ~8 000 characters of stdout.  The full record (every note, per-op table and latency probe) goes to stderr
# and, when the directory exists, to gpurun_out/bench_detail.json; stdout carries a compact line (< 6 000 characters) whose LAST keys
# are the second half of BASELINE's metric -- batched pairings/sec -- so that they survive any truncation from the front.
_DROP = {"note", "traffic_source", "launch_sampling", "sample_detail", "per_op_ns", "host", "mac32_per_launch", "reference_algorithm_mac32_per_unit",
         "executed_mac_per_unit", "executed_frac_of_peak", "mac32_per_unit_is", "launch_ms_isolated", "hbm_frac_of_8TBs", "table_build_s", "resident_bytes",
         "single_thread_value", "parallel_speedup", "algorithmic_bytes", "scalars", "single_call_scalar_muls_per_s", "end_to_end_scalar_muls_per_s",
         "with_final_exponentiation_ms", "window_bits", "per_term_path_ms", "launches_timed", "peak", "unit", "bound", "kernel", "mac32_per_unit", "achieved", "valu_wave_instructions", "valu_source"}
_KEEP_SMALL = ("pairing_n1_ms", "pairing_n1024_ms", "final_exponentiation_n1_ms", "multi_miller_loop_n3_plus_final_exponentiation_ms")


def _num(x):
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float("%.5g" % x)


def _slim_timing(t):
    return {"min": _num(t.get("min_ms")), "med": _num(t.get("median_ms")), "max": _num(t.get("max_ms")), "reps": t.get("reps"),
            # (the five longest kernels of the call: the verification chain launches ten)
            "kernel_ms": dict(sorted(((k, (_num(v) if not isinstance(v, dict) else _num(v.get("total_ms")))) for k, v in (t.get("kernel_ms") or {}).items()), key=lambda kv: -kv[1])[:5])}


def _compact_extra(v, depth=0):
    """extras on the compact line: CPU-baseline samples cut to 60 characters, timing blocks only at the top level of a leg and of its `prepared` twin"""
    if not isinstance(v, dict):
        return v
    out = {}
    for k, x in v.items():
        if k == "sample" and isinstance(x, str):
            out[k] = x[:60]
        elif k == "unprepared_timing" or (k == "timing" and depth >= 2):
            continue
        elif k == "n65536" and isinstance(x, dict):
            out[k] = {a: b for a, b in _compact_extra(x, 2).items() if a in ("ms", "equations_per_s", "frac", "speedup_over_unprepared", "unprepared_same_equations_ms", "roofline")}
        elif k == "cpu_baseline" and isinstance(x, dict) and depth >= 1:
            out[k] = {a: b for a, b in x.items() if a in ("value", "cores", "kind")}
        else:
            out[k] = _compact_extra(x, depth + 1)
    return out


def _slim(v, top=False):
    if isinstance(v, dict):
        if "median_ms" in v and "kernel_ms" in v:
            return _slim_timing(v)
        return {k: _slim(x) for k, x in v.items() if (top or k not in _DROP) and x is not None}
    if isinstance(v, (list, tuple)):
        return [_slim(x) for x in v]
    if isinstance(v, str):
        return v if len(v) <= 200 else v[:197] + "..."
    return _num(v)


def slim_line(line):
    """the compact stdout form of a full bench record: required keys first, every block reduced to its numbers, the pairing half of the
    metric as top-level scalars at the very end"""
    out = {}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        out[k] = _num(line.get(k))
    cfg = line.get("config") or {}
    out["config"] = {k: (v if not isinstance(v, str) or len(v) <= 240 else v[:237] + "...") for k, v in cfg.items() if k != "scalars"}
    roof = line.get("roofline")
    if roof:
        r = {k: _num(roof.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_isolated", "launch_ms", "launch_ms_isolated", "whole_msm_frac_pipelined",
                                                "whole_msm_frac_single_call", "valu_issue_frac")}
        kt = roof.get("kernel_trace")
        if kt:
            r["kernel_trace"] = {"shader_mhz_lone_call": kt.get("shader_mhz_lone_call"), "simd_time_two_one_no_wavefront": kt.get("simd_time_two_one_no_wavefront")}
        out["roofline"] = r
    cpu = line.get("cpu_baseline")
    if cpu:
        out["cpu_baseline"] = {k: _slim(cpu.get(k)) for k in ("value", "unit", "cores", "kind", "sample", "gpu_result_matches") if cpu.get(k) is not None}
        if isinstance(out["cpu_baseline"].get("sample"), str):
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:140]
    if (line.get("latency") or {}).get("host_mirror_repeat_ms") is not None:
        out["host_mirror_repeat_ms"] = _num(line["latency"]["host_mirror_repeat_ms"])
    if (line.get("latency") or {}).get("end_to_end_from_scalar_limbs_ms") is not None:
        out["end_to_end_from_scalar_limbs_ms"] = _num(line["latency"]["end_to_end_from_scalar_limbs_ms"])
        out["host_to_bytes_one_thread_ms"] = _num(line["latency"].get("host_to_bytes_one_thread_ms"))
    for k in ("single_call_ms", "end_to_end_h2d_ms", "group_path", "result_matches_identity", "ranks"):
        if k == "ranks" and (line.get(k) or {}).get("world", 1) == 1:
            continue                                       # one rank: nothing to say
        if line.get(k) is not None:
            out[k] = _slim(line[k])
    ex = line.get("extras")
    tail = {}
    if ex:
        e2 = {}
        for k, v in ex.items():
            if k == "pairing_small_batches":
                e2[k] = {kk: _num(v[kk]) for kk in _KEEP_SMALL if kk in v}
            elif k in ("pairing_batch", "cpu_baseline_pairing", "pairings_per_s"):
                continue                                   # emitted LAST, below
            elif k == "gpu_state":
                keep = ("pci", "sclk", "mclk", "power_w", "power_cap_w", "temp_junction_c", "compute_partition", "memory_partition", "perf_level")
                e2[k] = {"before": {a: b for a, b in (v.get("before_extras") or {}).items() if a in keep}, "after": {a: b for a, b in (v.get("after_extras") or {}).items() if a in ("sclk", "power_w", "temp_junction_c")},
                         "env": {a: b for a, b in (v.get("runtime_env") or {}).items() if a.startswith(("HSA_", "GPU_", "BLSGPU_")) and a != "HSA_ENABLE_IPC_MODE_LEGACY"}}
            else:
                e2[k] = _compact_extra(_slim(v))
        out["extras"] = e2
        pb, mm, eq = ex.get("pairing_batch") or {}, ex.get("multi_miller_loop") or {}, ex.get("verification_equations") or {}
        if pb:
            # the second half of BASELINE's metric, with its roofline block and CPU baseline intact
            rf = pb.get("roofline") or {}
            tail["pairing_batch"] = {"n": pb.get("n"), "ms": _num(pb.get("ms")), "timing": _slim(pb.get("timing") or {}) or None, "clocks_under_load": _slim(pb.get("clocks_under_load") or {}) or None,
                                     "roofline": {k: _num(rf.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "mac32_per_unit", "valu_issue_frac")}}
            cp = ex.get("cpu_baseline_pairing")
            if cp:
                tail["cpu_baseline_pairing"] = {k: _slim(cp.get(k)) for k in ("value", "unit", "cores", "kind", "sample", "gpu_result_matches") if cp.get(k) is not None}
                if isinstance(tail["cpu_baseline_pairing"].get("sample"), str):
                    tail["cpu_baseline_pairing"]["sample"] = tail["cpu_baseline_pairing"]["sample"][:110]
            tail["pairings_per_s"] = _num(ex.get("pairings_per_s"))
            tail["pairing_ms"] = _num(pb.get("ms"))
            tail["pairing_frac"] = _num(rf.get("frac"))
        if mm:
            tail["mml_terms_per_s"] = _num(ex.get("multi_miller_loop_terms_per_s"))
            tail["mml_ms"] = _num(mm.get("ms"))
            tail["mml_frac"] = _num((mm.get("roofline") or {}).get("frac"))
        if eq:
            tail["equations_per_s"] = _num(eq.get("equations_per_s"))
            tail["equations_frac"] = _num((eq.get("roofline") or {}).get("frac"))
            pq = eq.get("prepared") or {}
            if pq:
                tail["prepared_equations_per_s"] = _num(pq.get("equations_per_s"))
                tail["prepared_equations_speedup"] = _num(pq.get("speedup_over_unprepared"))
    out.update(tail)
    return out


def emit(line):
    """full record -> stderr (+ gpurun_out/bench_detail.json); compact line -> stdout"""
    full = json.dumps(line)
    sys.stderr.write("bench-detail: " + full + "\n")
    try:
        d = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(d):
            with open(os.path.join(d, "bench_detail.json"), "w") as fh:
                fh.write(full + "\n")
    except OSError:
        pass
    sys.stderr.flush()
    print(json.dumps(slim_line(line)))
    sys.stdout.flush()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["msm", "mixed"], default="msm")
    ap.add_argument("--log-n", type=int, default=None, help="log2 of points per GPU (N=1 default 20 = BASELINE configs[1]; with --weak also for N>1)")
    ap.add_argument("--log-total", type=int, default=24, help="N>1: log2 of the points of the ONE sharded MSM (default 24 = BASELINE configs[3])")
    ap.add_argument("--weak", action="store_true", help="N>1: fixed 2^log-n points per GPU instead of one 2^log-total MSM split N ways")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for single-GPU plumbing tests)")
    ap.add_argument("--same-device", action="store_true", help="testing aid: all ranks use GPU 0 (needs --backend gloo)")
    ap.add_argument("--dist-single", action="store_true", help="testing aid: run the N>1 code path (process group over RCCL, all-gather of the partial "
                    "sums, fold, rank agreement) with ONE rank -- the only way to execute that path on a one-GPU box")
    ap.add_argument("--group", type=int, default=0, metavar="N", help="ONE process driving N members of a device group of the C library (blsgpu_group_*: one context + one "
                    "persistent host thread per member, asynchronous pipelined MSMs, partial sums folded on member 0) instead of one process per GPU over RCCL; "
                    "members are GPUs 0..N-1, or N logical members on GPU 0 when fewer GPUs are visible (or with --same-device)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (batched pairings, G2 MSM, latency probes)")
    ap.add_argument("--mixed-log", type=int, nargs=3, default=[22, 22, 18], metavar=("G1", "G2", "MML"),
                    help="--workload mixed: log2 sizes of the three jobs per node (default 22 22 18 = BASELINE configs[4])")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: become the launcher (one rank per GPU)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


class Env:
    pass


def setup(args):
    import torch
    import torch.distributed as dist
    e = Env()
    e.torch, e.dist = torch, dist
    e.rank = int(os.environ.get("RANK", "0"))
    e.local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    e.world = int(os.environ.get("WORLD_SIZE", "1"))
    if e.world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, e.world))
    torch.cuda.set_device(e.local_rank)
    e.dev = torch.device("cuda", e.local_rank)
    e.multi = e.world > 1 or args.dist_single          # the distributed code path (exchange + fold) is active
    if e.multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if e.world == 1:
            s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1])); s.close()
        if args.backend == "nccl" and not os.environ.get("BENCH_EAGER_PG"):
            # NO device_id: binding the process group to the device creates the RCCL communicator eagerly, and with it every launch of this
            # process got slower -- a one-rank run at 2^21 points took 6.9 ms per step against 5.7 ms, with NO collective inside the timed steps
            # and unchanged kernel durations (1.2 ms of launch gaps per step; round 5, profiles/r05_group_compare.log).  The communicator is
            # created by the first collective instead (torch.cuda.set_device above names the device).
            dist.init_process_group("nccl", rank=e.rank, world_size=e.world)
        elif args.backend == "nccl":                                              # diagnostic: the eager form of rounds 2-4
            dist.init_process_group("nccl", rank=e.rank, world_size=e.world, device_id=e.dev)
        else:
            dist.init_process_group(args.backend, rank=e.rank, world_size=e.world)
    e.xdev = e.dev if args.backend == "nccl" else torch.device("cpu")       # where the exchanged partials live
    import bls12_381_amd as bls
    e.bls = bls
    return e


def fence(e):
    e.torch.cuda.synchronize()
    if e.multi:
        e.dist.barrier(device_ids=[e.local_rank]) if e.dist.get_backend() == "nccl" else e.dist.barrier()
    e.torch.cuda.synchronize()


def max_over_ranks(e, dt):
    if not e.multi:
        return dt
    t = e.torch.tensor([dt], dtype=e.torch.float64, device=e.xdev)
    e.dist.all_reduce(t, op=e.dist.ReduceOp.MAX)
    return float(t.item())


def ranks_agree(e, limbs_np):
    """every rank must hold the same value: compare through a max/min all-reduce"""
    t = e.torch.from_numpy(limbs_np.view(np.int64).copy()).to(e.xdev)
    hi, lo = t.clone(), t.clone()
    e.dist.all_reduce(hi, op=e.dist.ReduceOp.MAX); e.dist.all_reduce(lo, op=e.dist.ReduceOp.MIN)
    return bool(e.torch.equal(hi, lo))


def median_ms(fn, sync, warm=3, reps=11):
    for _ in range(warm):
        fn(); sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); sync()
        ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))


def timed_leg(ctx, fn, sync, warm=2, reps=15):
    """One measured leg of the pairing family: `warm` untimed calls (every reserve() growth, table build and wide-program load happens
    there), `reps` timed calls reported as min / median / max, then ONE more call with the library's per-kernel HIP events switched
    on (blsgpu_kernel_timing: events on the streams the kernels are launched on) -> kernel_ms.  Returns (median_ms, stats)."""
    for _ in range(warm):
        fn(); sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); sync()
        ts.append(1e3 * (time.perf_counter() - t))
    ctx.kernel_timing(True)
    try:
        t = time.perf_counter(); fn(); sync()
        timed_call = 1e3 * (time.perf_counter() - t)
        rep_ = ctx.kernel_timing_report()
    finally:
        ctx.kernel_timing(False)
    kern = {k.strip("()"): (round(v["total_ms"], 4) if v["launches"] == 1 else {"launches": v["launches"], "total_ms": round(v["total_ms"], 4), "max_ms": round(v["max_ms"], 4)}) for k, v in rep_.items()}
    ksum = sum(v["total_ms"] for v in rep_.values())
    med = float(np.median(ts))
    return med, {"min_ms": float(np.min(ts)), "median_ms": med, "max_ms": float(np.max(ts)), "first_rep_ms": ts[0], "reps": reps, "warmups": warm,
                 "kernel_ms": kern, "kernel_sum_ms": ksum, "call_with_events_ms": timed_call}


def gpu_state(dev_index=0):
    """clocks / power / partition modes of the device as the kernel driver reports them (sysfs first: no process start; rocm-smi --json for
    what sysfs does not show).  Diagnostic context for the bench line: a box-to-box difference of a kernel time should be explainable from here."""
    import glob
    import subprocess
    out = {}
    try:
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        base = None
        # the card whose PCI address is the HIP device's (a container can see every card of the host in sysfs while HIP sees one GPU: reading
        # "card0" there gives the idle clocks of somebody else's GPU)
        try:
            import torch
            pr = torch.cuda.get_device_properties(dev_index)
            want = "%04x:%02x:%02x." % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            for c_ in cards:
                if os.path.basename(os.path.realpath(os.path.dirname(c_))).startswith(want):
                    base = os.path.dirname(c_); out["pci"] = want + "0"
        except Exception:
            pass
        out["cards_in_sysfs"] = len(cards)
        if base is None and cards:
            base = os.path.dirname(cards[0]); out["pci"] = "unmatched (first card in sysfs)"
        if base is not None:

            def cur(name):
                try:
                    lines = open(os.path.join(base, name)).read().strip().splitlines()
                    star = [l for l in lines if l.rstrip().endswith("*")]
                    return (star[0] if star else lines[-1]).split(":")[1].replace("*", "").strip()
                except Exception:
                    return None
            out["sclk"], out["mclk"], out["fclk"] = cur("pp_dpm_sclk"), cur("pp_dpm_mclk"), cur("pp_dpm_fclk")
            for name, key in (("current_compute_partition", "compute_partition"), ("current_memory_partition", "memory_partition"), ("power_dpm_force_performance_level", "perf_level"),
                              ("gpu_busy_percent", "busy_percent"), ("mem_info_vram_used", "vram_used")):
                try:
                    out[key] = open(os.path.join(base, name)).read().strip()
                except Exception:
                    pass
            for hw in glob.glob(os.path.join(base, "hwmon", "hwmon*")):
                for name, key, scale in (("power1_average", "power_w", 1e-6), ("power1_input", "power_w", 1e-6), ("power1_cap", "power_cap_w", 1e-6), ("temp1_input", "temp_c", 1e-3),
                                         ("temp2_input", "temp_junction_c", 1e-3), ("temp3_input", "temp_mem_c", 1e-3)):
                    try:
                        out.setdefault(key, round(float(open(os.path.join(hw, name)).read().strip()) * scale, 1))
                    except Exception:
                        pass
    except Exception as ex:
        out["sysfs_error"] = str(ex)[:120]
    if not out.get("sclk"):
        try:
            js = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--showcomputepartition", "--showmemorypartition", "--json"],
                                           capture_output=True, text=True, timeout=20).stdout)
            card = js.get("card%d" % dev_index) or next(iter(js.values()))
            out["rocm_smi"] = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "power", "partition", "performance"))}
        except Exception as ex:
            out["rocm_smi_error"] = str(ex)[:120]
    return out


def clocks_under_load(fn, sync, launches=8, dev_index=0):
    """sclk / power sampled from sysfs by a host thread WHILE `launches` back-to-back calls of fn run (one sample every ~20 ms; the driver's
    readings lag the hardware by a few hundred ms, so the window is ~0.8 s and the LAST quarter of the samples is what counts)"""
    import threading
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            st = gpu_state(dev_index)
            samples.append((st.get("sclk"), st.get("power_w"), st.get("temp_junction_c", st.get("temp_c"))))
            time.sleep(0.02)
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t = time.perf_counter()
    for _ in range(launches):
        fn()
    sync()
    dt = time.perf_counter() - t
    stop.set(); th.join(timeout=2)

    def mhz(v):
        try:
            return float(str(v).lower().replace("mhz", ""))
        except Exception:
            return None
    samples = samples[-max(1, len(samples) // 4):]
    sc = [mhz(a) for a, _, _ in samples if mhz(a) is not None]
    pw = [b for _, b, _ in samples if b is not None]
    tj = [c for _, _, c in samples if c is not None]
    return {"launches": launches, "ms_per_launch": 1e3 * dt / launches, "samples": len(samples),
            "sclk_mhz": {"min": min(sc), "median": float(np.median(sc)), "max": max(sc)} if sc else None,
            "power_w": {"median": float(np.median(pw)), "max": max(pw)} if pw else None, "temp_c_max": max(tj) if tj else None}


def runtime_env():
    """the HIP / ROCr switches that change how scratch-heavy kernels are dispatched, as this process sees them"""
    keys = [k for k in os.environ if k.startswith(("HSA_", "HIP_", "ROCR_", "GPU_", "AMD_", "BLSGPU_"))]
    return {k: os.environ[k][:60] for k in sorted(keys)}


# =====================================================================================================================
# workload "msm"
# =====================================================================================================================
def run_msm(args, e):
    torch, dist, bls = e.torch, e.dist, e.bls
    from bls12_381_amd import synthetic
    from bls12_381_amd.distributed import shard_range, all_gather_rows
    rank, world, dev = e.rank, e.world, e.dev
    multi = e.multi
    steps = args.steps if args.steps is not None else (100 if not multi else 20)
    warmup = args.warmup if args.warmup is not None else 5
    strong = multi and not args.weak
    if strong:
        total = 1 << args.log_total
        lo, hi = shard_range(total, rank, world)
        n = hi - lo
    else:
        n = 1 << (args.log_n if args.log_n is not None else 20)
        total = n * world
    ctx = bls.Context(e.local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    # this rank's shard of the synthetic input: scalars uniform in [0, r) (all 255 bits in play), bases [k_i]G1
    kb = synthetic.scalars(n, synthetic.SEED + 2 * rank + 1)
    sb = synthetic.scalars(n, synthetic.SEED + 2 * rank)
    bases = ctx.bases_from_scalars(1, kb)                       # resident bases, built on the device
    d_scalars = torch.from_numpy(sb).to(dev)
    d_out = [torch.zeros(18, dtype=torch.int64, device=dev) for _ in range(8)]      # up to four calls in flight + the results the exchanges still read
    gathered = torch.zeros((world, 18), dtype=torch.int64, device=e.xdev) if multi else None
    d_fold = torch.zeros(18, dtype=torch.int64, device=dev)
    ctx.set_pipelining(True)         # the latency-bound tail of MSM i overlaps the chip-filling phases of MSM i+1
    state = {"i": 0}
    LAG = 3                          # result i - 3 is consumed while MSMs i - 2 .. i are in flight (four pipeline slots)
    # The exchange runs on a stream of its OWN: the front of an MSM waits for whatever is queued on the context's stream when it is launched
    # (that is how it is ordered behind the producer of its scalars); with the exchange elsewhere a rank whose peers are late never holds its
    # own next front back.  The context's stream is switched to the exchange stream for the join and the fold.  (Measured with one rank the
    # two placements are equal -- 2.78 / 11.1 / 41.3 ms per step at 2^20 / 2^22 / 2^24 points against 2.77 / 11.0 / 41.0 on the context's
    # stream, BENCH_EXCHANGE_MAIN=1 -- the ~1 ms per step the RCCL branch used to cost came from the eagerly created communicator: setup().)
    main_stream = torch.cuda.current_stream()
    xs = torch.cuda.Stream(device=dev) if multi else None
    ex_ev = [torch.cuda.Event() for _ in range(8)] if xs is not None else None
    ex_used = [False] * 8
    probe = os.environ.get("BENCH_EXCHANGE_PROBE", "")        # diagnostic: "join" / "gather" / "fold" run only that part of the exchange
    if os.environ.get("BENCH_EXCHANGE_MAIN"):                 # diagnostic: the exchange on the context's stream (rounds 2-4)
        xs_keep, xs = xs, None

    def exchange(j, lag):
        """the path's single exchange step for MSM j: all-gather the per-rank partial sums, fold on every rank"""
        buf = d_out[j & 7]
        if xs is None:                           # diagnostic placement (BENCH_EXCHANGE_MAIN): on the context's stream
            ctx.join(lag)
            if e.xdev != dev:
                # torch's current stream is the NULL stream here, the context's stream (blsgpu_set_stream(NULL) = its own non-blocking stream) is
                # not: a torch copy is not ordered behind the join -- round 6 found the gloo path gathering zeros this way -- so wait on the host
                ctx.synchronize()
            all_gather_rows(gathered, buf.to(e.xdev), dist)
            g = gathered if gathered.device == dev else gathered.to(dev)
            ctx.point_sum_device(1, g.data_ptr(), world, d_fold.data_ptr())
            state["g"] = g
            return
        ctx.set_stream(xs.cuda_stream)
        try:
            ctx.join(lag)                        # the exchange stream waits for the tail of MSM j
            with torch.cuda.stream(xs):
                if e.xdev != dev:
                    # CPU collectives (gloo: the one-GPU plumbing tests): the partial sum goes through the host -- the copy is queued on the exchange
                    # stream BEHIND the join and blocks the host until it is done -- and the gathered block comes back to the GPU for the fold
                    all_gather_rows(gathered, buf.to(e.xdev), dist)
                    g = gathered.to(dev)
                    ctx.point_sum_device(1, g.data_ptr(), world, d_fold.data_ptr())
                    state["g"] = g
                else:
                    if probe in ("", "gather"):
                        all_gather_rows(gathered, buf, dist)
                    if probe in ("", "fold"):
                        ctx.point_sum_device(1, gathered.data_ptr(), world, d_fold.data_ptr())     # asynchronous fold on this rank's GPU
                ex_ev[j & 7].record(xs)
            ex_used[j & 7] = True
        finally:
            ctx.set_stream(main_stream.cuda_stream)

    def step():
        i = state["i"]; state["i"] = i + 1
        if xs is not None and ex_used[i & 7]:
            main_stream.wait_event(ex_ev[i & 7])     # the exchange of MSM i - 8 read this output buffer (long complete: no stall, formal ordering)
        ctx.msm_device(bases, d_scalars.data_ptr(), n, d_out[i & 7].data_ptr())
        if multi and i >= LAG and probe != "none":
            # consume result i - LAG: MSM i+1 reuses the pipeline slot of MSM i-3 and must wait for that tail in any case; consuming i - 2
            # (rounds 2-4) made the pipeline three calls deep
            exchange(i - LAG, LAG)

    def drain():
        i = state["i"]
        if multi:
            for j in range(max(0, i - LAG), i):
                exchange(j, i - 1 - j)
        ctx.join(0)
        if xs is not None:
            main_stream.wait_stream(xs)
        state["i"] = 0

    for _ in range(warmup):
        step()
    drain()
    fence(e)
    ctx.msm_accumulate_stats(3)                # HIP events around every 3rd launch of the dominant kernel inside the timed region (the
                                               # two event records cost ~0.05-0.1 ms of queue time per timed MSM in a pipelined run;
                                               # 3 is coprime to the four pipeline slots, so every slot is sampled)
    t0 = time.perf_counter()
    trace = os.environ.get("BENCH_TRACE")
    for _ in range(steps):
        step()
        if trace:
            sys.stderr.write("rank %d step done at %.3f s\n" % (rank, time.perf_counter() - t0))
    drain()
    if trace:
        sys.stderr.write("rank %d drained at %.3f s\n" % (rank, time.perf_counter() - t0))
    fence(e)
    dt = time.perf_counter() - t0
    live_acc_ms, live_acc_n = ctx.msm_accumulate_stats(False)
    dt = max_over_ranks(e, dt)
    aff_rccl = None
    identity_ok = None
    if multi:
        last = d_fold.cpu().numpy().view(np.uint64)
        aff = ctx.batch_normalize(1, last[None, :])[0][0]
        if not ranks_agree(e, aff):
            raise SystemExit("bench: ranks disagree on the folded MSM result")
        aff_rccl = aff.copy()
        # ... and the folded point IS the MSM (agreement alone would also hold for a wrong fold): the discrete-log identity
        # sum_i s_i [k_i]G = [sum_i s_i k_i mod r] G -- every rank contributes the share of its shard (eight u32 words), the shares are
        # all-gathered like the partial sums, and [t]G comes from the library's fixed-base path (itself pinned to the golden multiples)
        share = synthetic.dot_mod_r(sb, kb)
        sh_t = torch.from_numpy(np.frombuffer(share.to_bytes(32, "little"), dtype="<u4").astype(np.int64)).to(e.xdev)
        sh_all = torch.zeros((world, 8), dtype=torch.int64, device=e.xdev)
        all_gather_rows(sh_all, sh_t, dist)
        t_all = sum(int.from_bytes(np.ascontiguousarray(row.astype("<u4")).tobytes(), "little") for row in sh_all.cpu().numpy()) % synthetic.R_ORDER
        want_aff = ctx.bases_from_scalars(1, np.frombuffer(t_all.to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32)).download()[0][0]
        identity_ok = bool(np.array_equal(aff, want_aff))
        if not identity_ok:
            raise SystemExit("bench: the folded multi-rank MSM is not [sum s_i k_i] G")
    ctx.set_pipelining(False)
    d_out0 = d_out[0]
    log_n = int(round(np.log2(n))) if n & (n - 1) == 0 else None

    # ---- roofline of the dominant kernel, measured live with HIP events on the library's stream ----------
    roof = phases = latency = None
    if rank == 0:
        # v_mad_u64_u32 lane-ops/s = MAC32/s; the best of three probes (a probe right after a long chip-filling run can read
        # 10 % low, which would flatter every fraction below)
        peak = max(ctx.mad_throughput(2000) for _ in range(3))
        ctx.set_profiling(True)
        acc_ms, tot_ms = [], []
        for _ in range(5):
            ctx.msm_device(bases, d_scalars.data_ptr(), n, d_out0.data_ptr())
            ph = ctx.last_msm_phase_ms()
            acc_ms.append(ph["accumulate"]); tot_ms.append(ph["total"]); phases = ph
        ctx.set_profiling(False)
        windows = (256 + WINDOW_BITS - 1) // WINDOW_BITS
        mac32_per_launch = float(n) * windows * MAC32_G1_ADD
        # duration of the dominant kernel: average over its launches INSIDE the timed (pipelined) region, HIP events on the
        # stream it runs on; the isolated (one MSM at a time) duration is reported next to it
        dur = (live_acc_ms if live_acc_n else float(np.mean(acc_ms))) * 1e-3
        traffic, traffic_source = static_traffic("msm") if log_n == 20 else (None, None)
        roof = {
            "bound": "int-valu", "kernel": "k_msm_accumulate<G1>",
            "achieved": mac32_per_launch / dur / 1e12, "peak": peak / 1e12, "unit": "TMAC32/s",
            "frac": mac32_per_launch / dur / peak, "frac_isolated": mac32_per_launch / (float(np.mean(acc_ms)) * 1e-3) / peak, "traffic": traffic, "traffic_source": traffic_source,
            "launch_ms": dur * 1e3, "launches_timed": int(live_acc_n), "launch_sampling": "HIP events around every 3rd launch of the timed region", "launch_ms_isolated": float(np.mean(acc_ms)), "mac32_per_launch": mac32_per_launch,
            "whole_msm_frac_pipelined": (float(n) * MAC32_G1_MSM_2_20) / (dt / steps) / peak,
            "whole_msm_frac_single_call": (float(n) * MAC32_G1_MSM_2_20) / (float(np.mean(tot_ms)) * 1e-3) / peak,
            **valu_issue("msm", dur, peak),
            **kernel_trace_facts(),
            "note": "integer-VALU bound (no MFMA, HBM traffic ~13% of peak, see traffic): canonical 300 MAC32 per Fp mul, 11 Fp mul per mixed add, "
                    "16 windows (SURVEY.md 8d); peak = v_mad_u64_u32 issue rate measured in this run",
        }
    # ---- latency of ONE call (SURVEY.md 8d timing protocol) ------------------------------------------------
    if rank == 0 and not multi and not args.no_extras:
        sync = torch.cuda.synchronize
        single = median_ms(lambda: ctx.msm_device(bases, d_scalars.data_ptr(), n, d_out0.data_ptr()), sync)
        # the same call 40 more times, one after the other (a synchronisation between them): the shader clock climbs for ~35 ms after an idle moment
        # (profiles/r06_acc_trace.md), so the protocol's "3 warm-ups, median of 11" is taken part-way up that ramp; the last 11 of the 40 are at the top
        run40 = []
        for _ in range(40):
            t1 = time.perf_counter(); ctx.msm_device(bases, d_scalars.data_ptr(), n, d_out0.data_ptr()); sync(); run40.append(1e3 * (time.perf_counter() - t1))
        single_warm = float(np.median(run40[-11:]))
        pinned = torch.from_numpy(sb).pin_memory()
        host_out = np.zeros(18, dtype=np.uint64)
        import ctypes

        def e2e():
            bls._lib.check(ctx.lib.blsgpu_g1_msm(ctx.h, bases.handle, 0, ctypes.c_void_p(pinned.data_ptr()), n, ctypes.c_void_p(host_out.ctypes.data)), "g1_msm")
        e2e_ms = median_ms(e2e, lambda: None)
        # the same two latencies with the scalars as `&[Scalar]` memory holds them (Montgomery limbs): `Scalar::to_bytes`
        # runs inside k_glv_decompose, so a drop-in `msm(&[G1Affine], &[Scalar])` does no per-scalar host work at all
        limbs_np, ok_np = ctx.fr_from_bytes(sb)
        assert ok_np.all()
        d_limbs = torch.from_numpy(limbs_np.view(np.int64)).to(dev)
        d_out_m = torch.zeros(18, dtype=torch.int64, device=dev)
        single_mont = median_ms(lambda: ctx.msm_mont_device(bases, d_limbs.data_ptr(), n, d_out_m.data_ptr()), sync)
        mont_same = bool(np.array_equal(ctx.batch_normalize(1, d_out_m.cpu().numpy().view(np.uint64)[None, :])[0],
                                        ctx.batch_normalize(1, d_out0.cpu().numpy().view(np.uint64)[None, :])[0]))
        pinned_l = torch.from_numpy(limbs_np.view(np.int64)).pin_memory()
        host_out_m = np.zeros(18, dtype=np.uint64)

        def e2e_mont():
            bls._lib.check(ctx.lib.blsgpu_g1_msm_mont(ctx.h, bases.handle, 0, ctypes.c_void_p(pinned_l.data_ptr()), n, ctypes.c_void_p(host_out_m.ctypes.data)), "g1_msm_mont")
        e2e_mont_ms = median_ms(e2e_mont, lambda: None)
        # what the host-side conversion costs that the limb form removes: 2^16 `Scalar::to_bytes` of the C port on ONE thread, scaled to n
        to_bytes_host_ms = None
        if not args.no_cpu_baseline:
            from oracle import c_oracle
            if hasattr(c_oracle, "scalar_to_bytes_batch"):
                m_ = min(n, 1 << 16)
                t1 = time.perf_counter(); c_oracle.scalar_to_bytes_batch(limbs_np[:m_]); to_bytes_host_ms = 1e3 * (time.perf_counter() - t1) * n / m_
        del d_limbs, pinned_l
        # a drop-in caller that passes its base SLICE on every call (blsgpu_g1_msm_host, what the mirrored `msm_g1(&bases, &scalars)` does) with
        # the opt-in bases cache: first sight = one-shot upload, second = resident upload (subgroup test, images), then only the scalars move
        cctx = bls.Context(e.local_rank)
        cctx.set_bases_cache(2)
        xy_all, inf_all = bases.download(0, n)
        hm_t = []
        hm_out = np.zeros(18, dtype=np.uint64)
        sb_c = np.ascontiguousarray(sb)                      # pageable host memory, straight through the C ABI (the Python mirror's own argument checks are not what is timed)
        for _ in range(7):
            t1 = time.perf_counter()
            bls._lib.check(ctx.lib.blsgpu_g1_msm_host(cctx.h, ctypes.c_void_p(xy_all.ctypes.data), ctypes.c_void_p(inf_all.ctypes.data), ctypes.c_void_p(sb_c.ctypes.data), n,
                                                      ctypes.c_void_p(hm_out.ctypes.data)), "g1_msm_host")
            hm_t.append(1e3 * (time.perf_counter() - t1))
        hm_same = bool(np.array_equal(cctx.batch_normalize(1, hm_out[None, :])[0], ctx.batch_normalize(1, d_out0.cpu().numpy().view(np.uint64)[None, :])[0]))
        cctx.close()
        del xy_all
        latency = {"single_call_ms": single, "single_call_after_40_calls_ms": single_warm, "single_call_first_of_40_ms": run40[0], "end_to_end_h2d_ms": e2e_ms, "single_call_scalar_limbs_ms": single_mont, "end_to_end_from_scalar_limbs_ms": e2e_mont_ms,
                   "scalar_limbs_result_matches": mont_same, "host_to_bytes_one_thread_ms": to_bytes_host_ms, "host_mirror_first_ms": hm_t[0], "host_mirror_second_ms": hm_t[1],
                   "host_mirror_repeat_ms": float(np.median(hm_t[2:])), "host_mirror_matches": hm_same,
                   "single_call_scalar_muls_per_s": n / (single * 1e-3), "end_to_end_scalar_muls_per_s": n / (e2e_ms * 1e-3),
                   "note": "one MSM at a time, nothing else in flight: scalars in HBM -> projective result in HBM (single_call); scalars in pinned host "
                           "memory -> 32 MB H2D over PCIe -> MSM -> result on the host, synchronous blsgpu_g1_msm (end_to_end); 3 warm-ups, median of 11"}

    # ---- CPU baseline: the reference's own definition on the host cores (bounded sample) ------------------
    cpu = None
    if rank == 0 and not multi and not args.no_cpu_baseline:
        cpu = cpu_baseline_g1(ctx, bases, sb, n)

    # ---- secondary measurements of the same path (BASELINE configs[2]); never part of `value` ---------------
    extras = None
    if rank == 0 and not multi and not args.no_extras:
        extras = run_extras(args, e, ctx, bases, d_scalars, sb, n, peak, d_out0)

    # ---- the same sharded MSM from ONE process (the C library's device group: what a Rust host uses), timed by rank 0 while the other
    # ranks wait at a CPU-side barrier (gloo: an RCCL barrier would spin on their GPUs) ------------------------------------------------
    group_path = None
    # (BENCH_FORCE_GROUP_PATH: run this block with ONE rank too -- how the one-GPU box exercises it)
    if multi and args.backend == "nccl" and not args.same_device and ((world > 1 and not args.no_extras) or os.environ.get("BENCH_FORCE_GROUP_PATH")):
        # The group run is a CHILD process of rank 0 with a time limit (`bench.py --group N`, the mode tools/run_scale.sh also uses): its cross-device
        # paths have never run on real multi-GPU hardware (the build box has one GPU), and a fault or a hang there must not cost the line above.
        try:
            import datetime
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")        # the container's hostname may not resolve
            cpu_pg = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=900))
            torch.cuda.synchronize()
            dist.barrier(group=cpu_pg)
            if rank == 0:
                try:
                    group_path = group_child(args, world, strong, aff_rccl)
                except Exception as ex:           # never lose the headline over the secondary measurement
                    group_path = {"error": str(ex)[:200]}
            dist.barrier(group=cpu_pg)
        except Exception as ex:
            group_path = {"error": "no CPU-side process group: " + str(ex)[:160]}
    if rank == 0:
        if strong:
            workload = ("ONE 2^%d-point G1 MSM sharded over %d MI355X (%d points per GPU), bases and scalars resident in HBM; one result per step: "
                        "%s all-gather of the %d partial sums (144 B each) + fold on every rank" % (args.log_total, world, n, "RCCL" if args.backend == "nccl" else args.backend, world))
        else:
            workload = ("2^%d-point G1 MSM per MI355X, bases resident in HBM, scalars in HBM; one result per step%s"
                        % (log_n, "" if not multi else " (RCCL all-gather of N partial sums + fold)"))
        line = {
            "metric": "G1 MSM throughput (scalar-muls/sec) at 2^%d points" % (args.log_total if strong else log_n) + ("" if strong or not multi else " per GPU"),
            "value": float(total) * steps / dt, "unit": "scalar-muls/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u32 (14x28-bit limbs, 64-bit accumulators)", "data": "synthetic",
            "config": {"workload": workload, "points_per_gpu": n, "total_points": total, "parallelism": "shard%d" % world,
                       "scalars": "SplitMix64(0xB1512381 + 2*rank), uniform in [0, r) by rejection (SURVEY.md 8d)"},
            "result_matches_identity": identity_ok, "ranks": {"world": world, "backend": (dist.get_backend() if multi else None), "world_from_process_group": (dist.get_world_size() if multi else 1),
                                                                  "communicator": (None if not multi or args.backend != "nccl" else ("eager (device_id bound at init)" if os.environ.get("BENCH_EAGER_PG") else "lazy (created by the first collective)")),
                                                                  "same_device": bool(args.same_device), "devices_visible": torch.cuda.device_count()},
            "roofline": roof, "cpu_baseline": cpu, "latency": latency, "msm_phase_ms": phases, "extras": extras, "group_path": group_path,
        }
        if latency:
            line["single_call_ms"] = latency["single_call_ms"]; line["end_to_end_h2d_ms"] = latency["end_to_end_h2d_ms"]
        emit(line)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def host_cpu_allotment():
    """what this process may really use: affinity mask and cgroup CPU quota (a container can see 256 hardware threads and be
    allowed a dozen) -- reported next to the thread count so that the CPU baseline can be read correctly"""
    info = {"hw_threads_visible": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        info["cgroup_cpu_max"] = None if q[0] == "max" else float(q[0]) / float(q[1])
    except Exception:
        info["cgroup_cpu_max"] = None
    return info


def host_threads():
    """threads for the CPU baseline: what the cgroup quota / affinity mask really grants (oversubscribing a 16-CPU quota with
    256 threads only adds scheduler noise); 0 = let OpenMP decide"""
    h = host_cpu_allotment()
    k = h["affinity"]
    if h.get("cgroup_cpu_max"):
        k = min(k, max(1, int(h["cgroup_cpu_max"])))
    return int(k)


def cpu_baseline_g1(ctx, bases, sb, n):
    """oracle/bls_oracle.c (kind "port") on the host cores; also the per-op table of the reference's criterion points."""
    from oracle import c_oracle
    per_op = {}
    threads = host_threads()
    # single-thread figures from >= ~1 s samples after a warm-up
    xy1, inf1 = bases.download(0, 4096)
    c_oracle.g1_msm(xy1[:64], inf1[:64], sb[:64], 1)
    t1 = time.perf_counter(); c_oracle.g1_msm(xy1, inf1, sb[:4096], 1); one = 4096 / (time.perf_counter() - t1)
    per_op["g1_scalar_mul_ns"] = 1e9 / one
    # all cores: warm the thread pool, then a sample large enough that every thread gets hundreds of terms
    m = min(n, 1 << 16)
    xy, inf = bases.download(0, m)
    c_oracle.g1_msm(xy[:4096], inf[:4096], sb[:4096], threads)
    t1 = time.perf_counter()
    ref, used = c_oracle.g1_msm(xy, inf, sb[:m], threads)
    cdt = time.perf_counter() - t1
    got = ctx.msm(bases, sb[:m])
    same = bool(np.array_equal(ctx.batch_normalize(1, got[None, :])[0][0], c_oracle.g1_to_affine(ref)[0]))
    if not same:
        raise SystemExit("bench: GPU MSM over the CPU sample differs from the oracle")
    return {"value": m / cdt, "unit": "scalar-muls/s", "cores": used, "kind": "port",
            "sample": f"first 2^{int(np.log2(m))} (point, scalar) pairs of the workload: sum(P_i*s_i) by the reference's double-and-add + Sum (C port, OpenMP x{used}); 1 thread: {one:.0f}/s",
            "sample_detail": f"first 2^{int(np.log2(m))} (point, scalar) pairs of the same workload: sum(P_i*s_i) by 255-step double-and-add + Sum, "
                             f"C restatement of the reference algorithm (oracle/bls_oracle.c), OpenMP dynamic schedule over {used} threads (= the CPUs the cgroup quota / affinity grants this process) after a warm-up; "
                             f"single thread (4096-pair sample): {one:.0f}/s",
            "single_thread_value": one, "parallel_speedup": (m / cdt) / one, "gpu_result_matches": same, "per_op_ns": per_op,
            "host": host_cpu_allotment()}


def run_extras(args, e, ctx, bases, d_scalars, sb, n, peak, d_out):
    torch, bls, dev = e.torch, e.bls, e.dev
    from bls12_381_amd import synthetic
    extras = {}
    np_ = 1 << 16
    ka = synthetic.scalars(np_, 99)
    kq = synthetic.scalars(np_, 100)
    g1xy, g1f = ctx.bases_from_scalars(1, ka).download()
    g2xy, g2f = ctx.bases_from_scalars(2, kq).download()
    d_g1 = torch.from_numpy(g1xy.view(np.int64)).to(dev); d_g2 = torch.from_numpy(g2xy.view(np.int64)).to(dev)
    d_gt = torch.zeros((np_, 72), dtype=torch.int64, device=dev)
    sync = torch.cuda.synchronize
    state_before = gpu_state(e.local_rank)
    pfn = lambda: ctx.pairing_batch_device(d_g1.data_ptr(), d_g2.data_ptr(), np_, d_gt.data_ptr())
    pms, pstat = timed_leg(ctx, pfn, sync)
    load = clocks_under_load(pfn, sync, launches=40, dev_index=e.local_rank)
    pdt = pms * 1e-3
    extras["pairings_per_s"] = np_ / pdt
    ptraf, ptraf_src = static_traffic("pairing")
    extras["pairing_batch"] = {"n": np_, "ms": pms, "timing": pstat, "clocks_under_load": load, "note": "2^16 independent pairing(P_i, Q_i), inputs and outputs in HBM; quad layout (one pairing per four lanes, quad.hip.h)",
                               "roofline": {"bound": "int-valu", "kernel": "k_pairing_quad", "mac32_per_unit": MAC32_PAIRING, "achieved": np_ * MAC32_PAIRING / pdt / 1e12,
                                            "peak": peak / 1e12, "unit": "TMAC32/s", "frac": np_ * MAC32_PAIRING / pdt / peak,
                                            "traffic": ptraf, "traffic_source": ptraf_src, "algorithmic_bytes": np_ * 864}}
    # small batches: latency of ONE call with n = 1, 8, 64, 1024 pairs (pairing) and of one final exponentiation -- the sizes the
    # reference's own criterion points measure one at a time (benches/groups.rs:15-29); per_op_ns of the CPU port is next to them
    small = {}
    for k in (1, 8, 64, 256, 512, 1024, 4096):
        small["pairing_n%d_ms" % k] = median_ms(lambda: ctx.pairing_batch_device(d_g1.data_ptr(), d_g2.data_ptr(), k, d_gt.data_ptr()), sync, warm=1, reps=5)
    d_ml = torch.zeros((1024, 72), dtype=torch.int64, device=dev)
    ctx.miller_loop_batch_device(d_g1.data_ptr(), d_g2.data_ptr(), 1024, d_ml.data_ptr())
    for k in (1, 1024):
        small["final_exponentiation_n%d_ms" % k] = median_ms(
            lambda: bls._lib.check(ctx.lib.blsgpu_final_exponentiation_device(ctx.h, d_ml.data_ptr(), k, d_gt.data_ptr()), "final_exponentiation"), sync, warm=1, reps=5)
    small["miller_loop_n1_ms"] = median_ms(lambda: ctx.miller_loop_batch_device(d_g1.data_ptr(), d_g2.data_ptr(), 1, d_ml.data_ptr()), sync, warm=1, reps=5)
    small["multi_miller_loop_n3_plus_final_exponentiation_ms"] = median_ms(
        lambda: (ctx.multi_miller_loop_device(d_g1.data_ptr(), d_g2.data_ptr(), 3, d_ml.data_ptr()),
                 bls._lib.check(ctx.lib.blsgpu_final_exponentiation_device(ctx.h, d_ml.data_ptr(), 1, d_gt.data_ptr()), "final_exponentiation")), sync, warm=1, reps=5)
    small["note"] = ("one call, nothing else in flight, inputs and outputs in HBM.  Up to 1536 items a call takes the wide path (wide.hip.h: one "
                     "item per workgroup -- 1024 lanes, one workgroup per CU, for 1..256 items, which therefore cost the same; 512 lanes, two per "
                     "CU, above: 512 items in one pass, 1024 in two); larger batches take the quad kernels, flat at ~6 ms up to ~4096 items.  multi_miller_loop_n3 + final exponentiation is the shape of "
                     "one signature-verification equation (benches/groups.rs:15-29 measure the same operations one at a time on the CPU)")
    extras["pairing_small_batches"] = small
    ctx.pairing_batch_device(d_g1.data_ptr(), d_g2.data_ptr(), np_, d_gt.data_ptr()); sync()        # the CPU comparison below reads d_gt
    if not args.no_cpu_baseline:
        from oracle import c_oracle
        per_op = {}
        c_oracle.pairing_batch(0, g1xy[:8], g1f[:8], g2xy[:8], g2f[:8], 1)
        for mode, name, cnt in ((0, "full_pairing_ns", 512), (1, "miller_loop_ns", 1024), (2, "final_exponentiation_ns", 1024)):
            src = d_gt[:cnt].cpu().numpy().view(np.uint64) if mode == 2 else g1xy[:cnt]
            t1 = time.perf_counter()
            c_oracle.pairing_batch(mode, src, g1f[:cnt], g2xy[:cnt], g2f[:cnt], 1)
            per_op[name] = 1e9 * (time.perf_counter() - t1) / cnt
        mp = 1 << 14
        c_oracle.pairing_batch(0, g1xy[:1024], g1f[:1024], g2xy[:1024], g2f[:1024], host_threads())           # thread-pool warm-up
        t1 = time.perf_counter()
        cref, cused = c_oracle.pairing_batch(0, g1xy[:mp], g1f[:mp], g2xy[:mp], g2f[:mp], host_threads())
        cpdt = time.perf_counter() - t1
        same_p = bool(np.array_equal(d_gt[:mp].cpu().numpy().view(np.uint64), cref))
        one_p = 1e9 / per_op["full_pairing_ns"]
        extras["cpu_baseline_pairing"] = {"value": mp / cpdt, "unit": "pairings/s", "cores": cused, "kind": "port",
                                          "sample": f"first 2^14 of the same pairs: pairing() of pairings.rs (C port, OpenMP x{cused}); 1 thread: {one_p:.0f}/s",
                                          "single_thread_value": one_p, "parallel_speedup": (mp / cpdt) / one_p, "gpu_result_matches": same_p, "per_op_ns": per_op}
        if not same_p:
            raise SystemExit("bench: GPU pairings differ from the CPU oracle on the sample")
    # multi_miller_loop at BASELINE configs[4]'s size (2^18 terms, the 2^16 pairs tiled four times): one shared accumulator
    # per four terms, partial products multiplied up; no final exponentiation in the timed region
    nm = 4 * np_
    d_g1m, d_g2m = d_g1.repeat(4, 1), d_g2.repeat(4, 1)
    mms, mstat = timed_leg(ctx, lambda: ctx.multi_miller_loop_device(d_g1m.data_ptr(), d_g2m.data_ptr(), nm, d_gt.data_ptr()), sync, warm=2, reps=9)
    mdt = mms * 1e-3
    d_one = torch.zeros(72, dtype=torch.int64, device=dev)
    mfe = median_ms(lambda: (ctx.multi_miller_loop_device(d_g1m.data_ptr(), d_g2m.data_ptr(), nm, d_gt.data_ptr()),
                             bls._lib.check(ctx.lib.blsgpu_final_exponentiation_device(ctx.h, d_gt.data_ptr(), 1, d_one.data_ptr()), "final_exponentiation")), sync, warm=1, reps=3)
    extras["multi_miller_loop_terms_per_s"] = nm / mdt
    extras["multi_miller_loop"] = {"n": nm, "ms": mms, "timing": mstat, "with_final_exponentiation_ms": mfe,
                                   "note": "one product of 2^18 Miller values; `ms` and the roofline are the product alone, `with_final_exponentiation_ms` adds the ONE "
                                           "final exponentiation of BASELINE configs[4] (SURVEY.md 8d: one product + one final exp; wide path, ~0.8 ms)",
                                   "roofline": {"bound": "int-valu", "kernel": "k_multi_miller_shared", "mac32_per_unit": MAC32_MML_TERM, "achieved": nm * MAC32_MML_TERM / mdt / 1e12,
                                                "peak": peak / 1e12, "unit": "TMAC32/s", "frac": nm * MAC32_MML_TERM / mdt / peak,
                                                "traffic": static_traffic("mml")[0], "traffic_source": static_traffic("mml")[1], "algorithmic_bytes": nm * 288}}
    del d_g1m, d_g2m
    # N independent multi_miller_loops of k = 3 terms + final exponentiation in ONE call: bulk signature verification
    # (blsgpu_multi_miller_loop_many; pairings.rs:554-603 + :48-176 once per equation).  Canonical work: 3 Miller
    # loops of 6 900 + one final exponentiation of 9 100 field multiplications = 29 800 x 300 MAC32 per equation.
    ne, ke = 1 << 14, 3
    d_off = torch.arange(0, (ne + 1) * ke, ke, dtype=torch.int64, device=dev)
    d_eq = torch.zeros((ne, 72), dtype=torch.int64, device=dev)
    eqms, eqstat = timed_leg(ctx, lambda: ctx.multi_miller_loop_many_device(d_g1.data_ptr(), d_g2.data_ptr(), d_off.data_ptr(), ne, ne * ke, d_eq.data_ptr(), max_seg_terms=ke), sync)
    mac_eq = (ke * 6900 + 9100) * 300
    eq = {"n": ne, "terms_per_equation": ke, "ms": eqms, "timing": eqstat, "equations_per_s": ne / (eqms * 1e-3),
          "note": "2^14 equations prod_{j<3} e(P_ij, Q_ij) in one blsgpu_multi_miller_loop_many_device call (Miller values of the 3 x 2^14 terms on the quad kernels, segmented "
                  "Fp12 product, batched final exponentiation), inputs and outputs in HBM",
          "roofline": {"bound": "int-valu", "kernel": "k_pairing_quad (Miller) + k_fp12_prod_seg_quad + k_final_exp_quad", "mac32_per_unit": mac_eq,
                       "achieved": ne * mac_eq / (eqms * 1e-3) / 1e12, "peak": peak / 1e12, "unit": "TMAC32/s", "frac": ne * mac_eq / (eqms * 1e-3) / peak,
                       "algorithmic_bytes": ne * (ke * 288 + 576), "traffic": static_traffic("equations")[0], "traffic_source": static_traffic("equations")[1]}}
    if not args.no_cpu_baseline:
        from oracle import c_oracle
        from oracle import bls12_381_ref as o_ref
        ms_ = 1 << 11                                     # 2^11 equations = 6 144 Miller loops + 2 048 final exponentiations: a few seconds of CPU work
        t1 = time.perf_counter()
        cml, cused = c_oracle.pairing_batch(1, g1xy[:ms_ * ke], g1f[:ms_ * ke], g2xy[:ms_ * ke], g2f[:ms_ * ke], host_threads())
        t_ml = time.perf_counter() - t1
        # the two Fp12 products per equation are done here by the Python oracle on a 64-equation sample (checker only; ~1 % of an equation's work, not timed)
        chk = 64
        prods = np.zeros((chk, 72), dtype=np.uint64)
        lim = lambda w: o_ref.fp12_unflatten([o_ref.fp_from_mont_limbs([int(x) for x in w[6 * i:6 * i + 6]]) for i in range(12)])
        for s_ in range(chk):
            acc = o_ref.fp12_mul(o_ref.fp12_mul(lim(cml[ke * s_]), lim(cml[ke * s_ + 1])), lim(cml[ke * s_ + 2]))
            prods[s_] = np.concatenate([np.array(o_ref.fp_to_mont_limbs(c_), dtype=np.uint64) for c_ in o_ref.fp12_flatten(acc)])
        t1 = time.perf_counter()
        c_oracle.pairing_batch(2, cml[:ms_], None, None, None, host_threads())
        t_fe = time.perf_counter() - t1
        want_eq = c_oracle.pairing_batch(2, prods, None, None, None, 1)[0]
        eq["gpu_result_matches"] = bool(np.array_equal(d_eq[:chk].cpu().numpy().view(np.uint64), want_eq))
        eq["cpu_baseline"] = {"value": ms_ / (t_ml + t_fe), "unit": "equations/s", "cores": cused, "kind": "port",
                              "sample": f"first 2^11 equations: 3 x 2^11 Miller loops + 2^11 final exponentiations of the C restatement (oracle/bls_oracle.c), OpenMP over {cused} threads; "
                                        "the two Fp12 products per equation (~1 % of its work) are not in the CPU figure"}
        if not eq["gpu_result_matches"]:
            raise SystemExit("bench: GPU multi_miller_loop_many differs from the CPU oracle on the sample")
    # the same shape at 2^16 equations (the 2^16 pairs tiled three times): from 49 152 segments on the library shares the squarings inside a
    # segment (one lane pair per equation, k_multi_miller_seg); max_seg_terms = 0 forces the per-term path for comparison
    ne2 = 1 << 16
    d_g1e, d_g2e = d_g1.repeat(ke, 1), d_g2.repeat(ke, 1)
    d_off2 = torch.arange(0, (ne2 + 1) * ke, ke, dtype=torch.int64, device=dev)
    d_eq2 = torch.zeros((ne2, 72), dtype=torch.int64, device=dev)
    eq2 = median_ms(lambda: ctx.multi_miller_loop_many_device(d_g1e.data_ptr(), d_g2e.data_ptr(), d_off2.data_ptr(), ne2, ne2 * ke, d_eq2.data_ptr(), max_seg_terms=ke), sync, warm=1, reps=3)
    keep = d_eq2[:256].clone()
    eq2p = median_ms(lambda: ctx.multi_miller_loop_many_device(d_g1e.data_ptr(), d_g2e.data_ptr(), d_off2.data_ptr(), ne2, ne2 * ke, d_eq2.data_ptr(), max_seg_terms=0), sync, warm=1, reps=3)
    eq["n65536"] = {"ms": eq2, "equations_per_s": ne2 / (eq2 * 1e-3), "frac": ne2 * mac_eq / (eq2 * 1e-3) / peak, "per_term_path_ms": eq2p,
                    "paths_agree": bool(torch.equal(keep, d_eq2[:256])),
                    "note": "2^16 equations: shared accumulator per equation (k_multi_miller_seg) + batched final exponentiation; per_term_path_ms = the same call on the per-term quads"}
    # The same shape with two of the three G2 arguments FIXED (a verification key: Groth16 has three of four fixed, BLS a fixed
    # generator) and prepared once (blsgpu_g2_prepare: `G2Prepared::from`, pairings.rs:504-546): the prepared terms only evaluate their
    # stored lines and every equation shares ONE accumulator (blsgpu_multi_miller_loop_prepared_many_device).  Canonical work per
    # equation as the REFERENCE does it with prepared arguments: one unprepared term 6 900 + two prepared terms 2 900 each
    # + one final exponentiation 9 100 field multiplications.
    kkey = synthetic.scalars(2, 977)
    key_xy, _ = ctx.bases_from_scalars(2, kkey).download()
    table = ctx.g2_prepare(key_xy)
    mac_eq_prep = (6900 + 2 * 2900 + 9100) * 300
    prep = {"fixed_g2_arguments": 2, "terms_per_equation": ke, "table_bytes": 2 * 26112}
    for nn, d_o, d_e in ((ne, d_off, d_eq), (ne2, d_off2, d_eq2)):
        qi_np = np.full(nn * ke, bls.UNPREPARED, dtype=np.uint32); qi_np[1::3] = 0; qi_np[2::3] = 1
        d_qi = torch.from_numpy(qi_np.view(np.int32)).to(dev)
        d_gq = (d_g2e if nn == ne2 else d_g2)[:nn * ke].clone()
        d_gp = (d_g1e if nn == ne2 else d_g1)[:nn * ke]
        kt = torch.from_numpy(key_xy.view(np.int64)).to(dev)
        d_gq[1::3] = kt[0]; d_gq[2::3] = kt[1]                  # the same equations for the unprepared path: the fixed points written out per term
        tp, tpstat = timed_leg(ctx, lambda: ctx.multi_miller_loop_prepared_many_device(d_gp.data_ptr(), table, d_qi.data_ptr(), d_o.data_ptr(), nn, nn * ke, d_e.data_ptr(), max_seg_terms=ke,
                                                                                        d_g2=d_gq.data_ptr()), sync, warm=2, reps=15 if nn == ne else 5)
        got_p = d_e[:512].clone()
        tu, tustat = timed_leg(ctx, lambda: ctx.multi_miller_loop_many_device(d_gp.data_ptr(), d_gq.data_ptr(), d_o.data_ptr(), nn, nn * ke, d_e.data_ptr(), max_seg_terms=ke), sync, warm=2,
                               reps=15 if nn == ne else 5)
        rec = {"ms": tp, "timing": tpstat, "equations_per_s": nn / (tp * 1e-3), "unprepared_same_equations_ms": tu, "unprepared_timing": tustat, "speedup_over_unprepared": tu / tp,
               "paths_agree": bool(torch.equal(got_p, d_e[:512])),
               "roofline": {"bound": "int-valu", "kernel": "k_mml_prep_quad + k_final_exp_quad", "mac32_per_unit": mac_eq_prep, "achieved": nn * mac_eq_prep / (tp * 1e-3) / 1e12,
                            "peak": peak / 1e12, "unit": "TMAC32/s", "frac": nn * mac_eq_prep / (tp * 1e-3) / peak, "algorithmic_bytes": nn * (ke * 96 + 192 + 576), "traffic": None}}
        if nn == ne2:
            # counter traffic of the Miller kernel of this very call shape (2^16 equations), from the committed rocprofv3 --pmc passes
            rec["roofline"]["traffic"], rec["roofline"]["traffic_source"] = static_traffic("equations_prepared")
        if nn == ne:
            prep.update(rec); prep["n"] = nn
            if not args.no_cpu_baseline:
                # the reference's own schedule on the host cores: ONE accumulator per equation, the fixed arguments prepared once outside the
                # timed region, the variable one prepared inside it (`G2Prepared::from` is part of every call for a fresh point)
                from oracle import c_oracle
                ms_ = 1 << 11
                tabs = c_oracle.g2_prepare(key_xy)
                hg1 = d_gp[:ms_ * ke].cpu().numpy().view(np.uint64); hg2 = d_gq[:ms_ * ke].cpu().numpy().view(np.uint64)
                hoff = (np.arange(ms_ + 1) * ke).astype(np.uint64)
                c_oracle.multi_miller_prepared_many(hg1[:64 * ke], None, hg2[:64 * ke], None, qi_np[:64 * ke], tabs, None, hoff[:65], True, host_threads())
                t1 = time.perf_counter()
                cw, cused = c_oracle.multi_miller_prepared_many(hg1, None, hg2, None, qi_np[:ms_ * ke], tabs, None, hoff, True, host_threads())
                ct = time.perf_counter() - t1
                ctx.multi_miller_loop_prepared_many_device(d_gp.data_ptr(), table, d_qi.data_ptr(), d_o.data_ptr(), nn, nn * ke, d_e.data_ptr(), max_seg_terms=ke, d_g2=d_gq.data_ptr()); sync()
                prep["gpu_result_matches"] = bool(np.array_equal(d_e[:ms_].cpu().numpy().view(np.uint64), cw))
                prep["cpu_baseline"] = {"value": ms_ / ct, "unit": "equations/s", "cores": cused, "kind": "port",
                                        "sample": f"first 2^11 equations: multi_miller_loop over G2Prepared terms (pairings.rs:554-603 schedule, C port) + final exponentiation, OpenMP x{cused}"}
                if not prep["gpu_result_matches"]:
                    raise SystemExit("bench: GPU prepared multi_miller_loop_many differs from the CPU oracle on the sample")
        else:
            prep["n65536"] = rec
        if not rec["paths_agree"]:
            raise SystemExit("bench: prepared and unprepared Miller loops disagree")
        del d_qi, d_gq
    table.free()
    eq["prepared"] = prep
    extras["verification_equations"] = eq
    del d_off, d_eq, d_g1e, d_g2e, d_off2, d_eq2
    # Bulk signature verification from bytes, every stage on the device (blsgpu_bls_verify_batch_device): 2^14 (public key in G1, signature in
    # G2, 32-byte message) triples -> verdict bytes.  Work per signature in the units of SURVEY.md 8d, stage by stage: checked G1 decoding
    # ~2 500 (square root + subgroup test), checked G2 decoding ~9 000, hash_to_curve to G2 ~8 700, one two-term multi_miller_loop 2 x 6 900
    # + one final exponentiation 9 100 (the decoding / hashing counts are estimates of the algorithms' multiplication counts).
    nv = 1 << 14
    vdst = b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_"
    skb = synthetic.scalars(nv, 4242)
    vmsgs = np.random.RandomState(7).randint(0, 256, size=(nv, 32), dtype=np.uint8)
    pkxy, pkinf = ctx.bases_from_scalars(1, skb).download()
    hxy, hinf = ctx.batch_normalize(2, ctx.hash_to_curve(2, [m.tobytes() for m in vmsgs], vdst))
    sgxy, sginf = ctx.batch_normalize(2, ctx.mul_batch(2, hxy, hinf, skb))
    pk_b = ctx.points_to_bytes(1, pkxy, pkinf, compressed=True).copy()
    sg_b = ctx.points_to_bytes(2, sgxy, sginf, compressed=True).copy()
    sg_b[3] = sg_b[4]; sgxy = sgxy.copy(); sgxy[3] = sgxy[4]      # one forged entry (bytes for the device, affine limbs for the CPU check)
    d_pk, d_sg, d_vm = torch.from_numpy(pk_b).to(dev), torch.from_numpy(sg_b).to(dev), torch.from_numpy(vmsgs).to(dev)
    d_vo = torch.arange(0, (nv + 1) * 32, 32, dtype=torch.int64, device=dev)
    d_vd = torch.from_numpy(np.frombuffer(vdst, dtype=np.uint8).copy()).to(dev)
    d_vv = torch.zeros(nv, dtype=torch.uint8, device=dev)
    vms, vstat = timed_leg(ctx, lambda: ctx.bls_verify_batch_device(0, d_pk.data_ptr(), d_sg.data_ptr(), d_vm.data_ptr(), d_vo.data_ptr(), nv, d_vd.data_ptr(), len(vdst), d_vv.data_ptr()), sync)
    verd = d_vv.cpu().numpy()
    mac_ver = (2500 + 9000 + 8700 + 2 * 6900 + 9100) * 300
    ver = {"n": nv, "ms": vms, "timing": vstat, "signatures_per_s": nv / (vms * 1e-3), "verdicts_as_expected": bool(verd[3] == 0 and verd.sum() == nv - 1),
           "note": "compressed public keys (48 B) + signatures (96 B) + 32-byte messages in HBM -> verdict bytes in HBM: checked decoding, hash_to_curve, normalisation, "
                   "multi_miller_loop + final exponentiation per signature, identity test (blsgpu_bls_verify_batch_device)",
           "roofline": {"bound": "int-valu", "kernel": "k_point_decode x 2 + k_hash_to_curve<G2> + k_pairing_quad (Miller) + k_fp12_prod_seg_quad + k_final_exp_quad",
                        "mac32_per_unit": mac_ver, "mac32_per_unit_is": "estimate for the decoding and hashing stages", "achieved": nv * mac_ver / (vms * 1e-3) / 1e12, "peak": peak / 1e12,
                        "unit": "TMAC32/s", "frac": nv * mac_ver / (vms * 1e-3) / peak, "algorithmic_bytes": nv * (48 + 96 + 32 + 1), "traffic": static_traffic("bls_verify")[0], "traffic_source": static_traffic("bls_verify")[1]}}
    if not ver["verdicts_as_expected"]:
        raise SystemExit("bench: bulk verification verdicts are wrong")
    if not args.no_cpu_baseline:
        from oracle import c_oracle
        mv = 1 << 10
        gen1 = np.tile(np.array(bls.G1Affine.generator().xy, dtype=np.uint64), (mv, 1))
        t1 = time.perf_counter()
        lhs, cused = c_oracle.pairing_batch(0, pkxy[:mv], None, hxy[:mv], None, host_threads())
        rhs, _ = c_oracle.pairing_batch(0, gen1, None, sgxy[:mv], None, host_threads())
        cvt = time.perf_counter() - t1
        want_v = (lhs == rhs).all(axis=1)
        ver["gpu_result_matches"] = bool(np.array_equal(verd[:mv] == 1, want_v) and want_v.sum() == mv - 1)          # entry 3 is the forged one
        ver["cpu_baseline"] = {"value": mv / cvt, "unit": "signatures/s", "cores": cused, "kind": "port",
                               "sample": f"first 2^10 signatures, the PAIRING stage only (two pairings each, C port, OpenMP x{cused}); decoding and hash_to_curve are not in the CPU figure"}
        if not ver["gpu_result_matches"]:
            raise SystemExit("bench: bulk verification disagrees with the CPU oracle's pairings on the sample")
    extras["bls_verify_from_bytes"] = ver
    del d_pk, d_sg, d_vm, d_vo, d_vv
    # Fr transform of the MSM's scalar vector
    if n & (n - 1) == 0:
        log_n = int(np.log2(n))
        d_fr = d_scalars.clone()
        nms = median_ms(lambda: ctx.fr_ntt_device(d_fr.data_ptr(), log_n, False), sync, warm=1, reps=10)
        # the transform is bound by the integer VALU like everything else here: n/2 log2 n butterflies of one multiplication
        # (8 x 32-bit limbs: 136 MAC32 in SURVEY.md 8d's counting) -- its own bytes (every element read and written once, 64 B)
        # are 5 % of the HBM peak at this rate, and a one-pass-per-ten-stages variant that moved a third of the bytes ran at the
        # same speed
        ntt_bytes = n * 64
        ntt_mac = (n // 2) * log_n * 136
        extras["fr_ntt"] = {"log_n": log_n, "ms": nms, "elements_per_s": n / (nms * 1e-3),
                            "note": "radix-2 NTT over the scalar field, in place, natural order; 5 radix-4 passes over the data + one LDS pass at 2^20",
                            "roofline": {"bound": "int-valu", "kernel": "k_fr_stage2 x 5 + k_fr_tile", "mac32_per_unit": (log_n * 136) // 2,
                                         "achieved": ntt_mac / (nms * 1e-3) / 1e12, "peak": peak / 1e12, "unit": "TMAC32/s", "frac": ntt_mac / (nms * 1e-3) / peak,
                                         "algorithmic_bytes": ntt_bytes, "hbm_frac_of_8TBs": ntt_bytes / (nms * 1e-3) / 8e12,
                                         # what the butterflies EXECUTE: one 9 x 29-bit lazy Montgomery product = 171 v_mad_u64_u32 and ~375 VALU instructions in all (fr.hip.h;
                                         # instruction mix from the ISA, tools/isa_stats.py); against the issue rate of ALL VALU instructions (= the v_mad peak: they issue alike)
                                         "executed_mac_per_unit": (log_n * 171) // 2, "executed_valu_instructions_per_unit": (log_n * 375) // 2,
                                         "executed_valu_frac_of_issue_peak": (n // 2) * log_n * 375 / (nms * 1e-3) / peak,
                                         "traffic": static_traffic("ntt")[0], "traffic_source": static_traffic("ntt")[1]}}
        del d_fr
    # hash-to-curve in front of the pairings: 2^16 32-byte messages -> G2
    rs = np.random.RandomState(99)
    hm = torch.from_numpy(rs.randint(0, 256, size=np_ * 32, dtype=np.uint8)).to(dev)
    ho = torch.arange(0, (np_ + 1) * 32, 32, dtype=torch.int64, device=dev)
    hdst = b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_"
    hd = torch.from_numpy(np.frombuffer(hdst, dtype=np.uint8).copy()).to(dev)
    hout = torch.zeros((np_, 36), dtype=torch.int64, device=dev)

    def h2c():
        bls._lib.check(ctx.lib.blsgpu_hash_to_curve_device(ctx.h, 2, hm.data_ptr(), ho.data_ptr(), np_, hd.data_ptr(), len(hdst), 0, hout.data_ptr()), "hash_to_curve_device")
    hms = median_ms(h2c, sync, warm=1, reps=3)
    # work per hash in the units of SURVEY.md 8d (one Fp multiplication = 300 MAC32; an Fp2 product counts 3, an Fp2 square 2): two
    # square-root-ratio powers a^((p^2-9)/16) of the simplified SWU map (762 squarings + 204 products in Fp2 each: 2 x 2 136), the two
    # multiplications by |x| of the cofactor clearing (63 doublings + 5 additions on the twist each: 2 x ~1 900), isogeny, point
    # additions, psi maps, sgn0 / exceptional-case selects ~600: ~8 700 field multiplications (an estimate of the algorithm's count,
    # not an instruction count of the kernel)
    mac_h2c = 8700 * 300
    extras["hash_to_g2"] = {"n": np_, "ms": hms, "hashes_per_s": np_ / (hms * 1e-3), "note": "hash_to_curve (XMD:SHA-256, SSWU, RO) of 32-byte messages to G2",
                            "roofline": {"bound": "int-valu", "kernel": "k_hash_to_curve<G2, lane pair>", "mac32_per_unit": mac_h2c, "mac32_per_unit_is": "estimate (see bench.py)",
                                         "achieved": np_ * mac_h2c / (hms * 1e-3) / 1e12, "peak": peak / 1e12, "unit": "TMAC32/s", "frac": np_ * mac_h2c / (hms * 1e-3) / peak,
                                         "algorithmic_bytes": np_ * (32 + 288), "traffic": static_traffic("hash_to_g2")[0], "traffic_source": static_traffic("hash_to_g2")[1]}}
    # ... and to G1 (the min-sig placement), same messages
    hout1 = torch.zeros((np_, 18), dtype=torch.int64, device=dev)
    hdst1 = b"BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_NUL_"
    hd1 = torch.from_numpy(np.frombuffer(hdst1, dtype=np.uint8).copy()).to(dev)
    h1ms = median_ms(lambda: bls._lib.check(ctx.lib.blsgpu_hash_to_curve_device(ctx.h, 1, hm.data_ptr(), ho.data_ptr(), np_, hd1.data_ptr(), len(hdst1), 0, hout1.data_ptr()), "hash_to_curve_device"),
                     sync, warm=1, reps=3)
    mac_h2c1 = 2400 * 300        # two SSWU maps with one square-root-ratio power each (~2 x 560), the 11-isogeny, the addition, the cofactor power by 1 - x (~64 doublings + adds): ~2 400 mul (estimate)
    extras["hash_to_g1"] = {"n": np_, "ms": h1ms, "hashes_per_s": np_ / (h1ms * 1e-3),
                            "roofline": {"bound": "int-valu", "kernel": "k_hash_to_curve<G1>", "mac32_per_unit": mac_h2c1, "mac32_per_unit_is": "estimate (see bench.py)",
                                         "achieved": np_ * mac_h2c1 / (h1ms * 1e-3) / 1e12, "peak": peak / 1e12, "unit": "TMAC32/s", "frac": np_ * mac_h2c1 / (h1ms * 1e-3) / peak,
                                         "algorithmic_bytes": np_ * (32 + 144), "traffic": static_traffic("hash_to_g1")[0], "traffic_source": static_traffic("hash_to_g1")[1]}}
    del hm, ho, hout, hout1
    # point codecs with device pointers: checked decoding of 2^16 compressed points (square root + subgroup test) and encoding
    cod = {}
    for grp, xy_, cb in ((1, g1xy, 48), (2, g2xy, 96)):
        enc = ctx.points_to_bytes(grp, xy_, None, compressed=True)
        d_enc = torch.from_numpy(enc).to(dev)
        d_cx = torch.zeros((np_, 12 * grp), dtype=torch.int64, device=dev); d_ci = torch.zeros(np_, dtype=torch.uint8, device=dev); d_ck = torch.zeros(np_, dtype=torch.uint8, device=dev)
        dms = median_ms(lambda: ctx.points_from_bytes_device(grp, d_enc.data_ptr(), np_, d_cx.data_ptr(), d_ci.data_ptr(), d_ck.data_ptr(), compressed=True, checked=True), sync, warm=1, reps=3)
        ok_all = bool(d_ck.all().item()) and bool(torch.equal(d_cx, (d_g1 if grp == 1 else d_g2)))
        d_eo = torch.zeros((np_, cb), dtype=torch.uint8, device=dev)
        ems = median_ms(lambda: ctx.points_to_bytes_device(grp, d_cx.data_ptr(), d_ci.data_ptr(), np_, d_eo.data_ptr(), compressed=True), sync, warm=1, reps=3)
        # square root + subgroup test, estimates of the multiplication counts of the algorithms the kernels RUN: G1 ~490 (x^((p+1)/4), 4-bit windows)
        # + two multiplications by x (~2 000); G2 ~1 100 (two base-field exponentiations + an inversion, codec.hip.h::fe2_sqrt; the reference's
        # Fp2::sqrt is ~2 700) + one multiplication by x and psi (~1 500)
        mac_dec = (2500 if grp == 1 else 2600) * 300
        cod["g%d" % grp] = {"n": np_, "decode_checked_ms": dms, "decoded_per_s": np_ / (dms * 1e-3), "encode_ms": ems, "roundtrip_ok": ok_all and bool(torch.equal(d_eo, d_enc)),
                            "roofline": {"bound": "int-valu", "kernel": "k_point_decode<G%d>" % grp, "mac32_per_unit": mac_dec, "mac32_per_unit_is": "estimate (see bench.py)",
                                         "achieved": np_ * mac_dec / (dms * 1e-3) / 1e12, "peak": peak / 1e12, "unit": "TMAC32/s", "frac": np_ * mac_dec / (dms * 1e-3) / peak,
                                         "algorithmic_bytes": np_ * (cb + 2 * cb + 2), "traffic": static_traffic("decode_g%d" % grp)[0], "traffic_source": static_traffic("decode_g%d" % grp)[1]}}
        if not cod["g%d" % grp]["roundtrip_ok"]:
            raise SystemExit("bench: codec round trip failed")
        del d_enc, d_cx, d_eo
    extras["codec"] = cod
    # 2^20-point G2 MSM (BASELINE configs[2]), pipelined like the headline
    n2 = min(1 << 20, n)
    k2 = synthetic.scalars(n2, 101)
    b2 = ctx.bases_from_scalars(2, k2)
    d_s2 = torch.from_numpy(sb[:n2].copy()).to(dev)
    d_o2 = [torch.zeros(36, dtype=torch.int64, device=dev) for _ in range(4)]
    ctx.set_pipelining(True)
    for i in range(2):
        ctx.msm_device(b2, d_s2.data_ptr(), n2, d_o2[i].data_ptr())
    ctx.join(0); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(24):
        ctx.msm_device(b2, d_s2.data_ptr(), n2, d_o2[i & 3].data_ptr())
    ctx.join(0); torch.cuda.synchronize()
    g2dt = (time.perf_counter() - t1) / 24
    ctx.set_pipelining(False)
    g2single = median_ms(lambda: ctx.msm_device(b2, d_s2.data_ptr(), n2, d_o2[0].data_ptr()), sync, warm=1, reps=5)
    if not args.no_cpu_baseline:
        from oracle import c_oracle
        m2 = 1 << 13                                  # bounded sample: a few seconds of CPU work
        xy2, inf2 = b2.download(0, m2)
        c_oracle.g2_msm(xy2[:512], inf2[:512], sb[:512], host_threads())
        t1 = time.perf_counter()
        ref2, used2 = c_oracle.g2_msm(xy2, inf2, sb[:m2], host_threads())
        c2dt = time.perf_counter() - t1
        t1 = time.perf_counter(); c_oracle.g2_msm(xy2[:1024], inf2[:1024], sb[:1024], 1); one2 = 1024 / (time.perf_counter() - t1)
        got2 = ctx.msm(b2, sb[:m2])
        same2 = bool(np.array_equal(ctx.batch_normalize(2, got2[None, :])[0][0], c_oracle.g2_to_affine(ref2)[0]))
        extras["cpu_baseline_g2_msm"] = {"value": m2 / c2dt, "unit": "scalar-muls/s", "cores": used2, "kind": "port",
                                         "sample": "first 2^13 (point, scalar) pairs: sum(P_i*s_i) over G2 by 255-step double-and-add + Sum "
                                                   f"(oracle/bls_oracle.c), OpenMP over {used2} threads after a warm-up; single thread (1024-pair sample): {one2:.0f}/s",
                                         "single_thread_value": one2, "parallel_speedup": (m2 / c2dt) / one2, "gpu_result_matches": same2,
                                         "per_op_ns": {"g2_scalar_mul_ns": 1e9 / one2}}
        if not same2:
            raise SystemExit("bench: GPU G2 MSM differs from the CPU oracle on the sample")
    extras["g2_msm_scalar_muls_per_s"] = n2 / g2dt
    extras["g2_msm"] = {"n": n2, "ms": 1e3 * g2dt, "single_call_ms": g2single,
                        "roofline": {"bound": "int-valu", "kernel": "whole G2 MSM (k_msm_accumulate_g2pair dominant)", "mac32_per_unit": MAC32_G2_MSM_2_20,
                                     "achieved": n2 * MAC32_G2_MSM_2_20 / g2dt / 1e12, "peak": peak / 1e12, "unit": "TMAC32/s", "frac": n2 * MAC32_G2_MSM_2_20 / g2dt / peak,
                                     "algorithmic_bytes": n2 * (192 + 32) + 288, "traffic": static_traffic("g2_msm")[0], "traffic_source": static_traffic("g2_msm")[1]}}
    # batched variable-base scalar multiplication, N in -> N out (the reference's unit operation `&G1Affine * &Scalar`)
    xy1, _ = bases.download(0, n)
    d_xy1 = torch.from_numpy(xy1.view(np.int64)).to(dev)
    d_mo = torch.zeros((n, 18), dtype=torch.int64, device=dev)
    mbms = median_ms(lambda: ctx.mul_batch_device(1, d_xy1.data_ptr(), 0, d_scalars.data_ptr(), n, d_mo.data_ptr()), sync, warm=1, reps=3)
    mb = {"n": n, "ms": mbms, "scalar_muls_per_s": n / (mbms * 1e-3),
          "note": "n independent P_i * s_i -> n projective points (blsgpu_g1_mul_batch_device), signed 4-bit windows over the complete formulas: 2 852 field "
                  "multiplications per unit (what the roofline counts, 300 MAC32 each) against the 5 100 of the reference's double-and-add",
          "roofline": {"bound": "int-valu", "kernel": "k_mul_batch<G1>", "mac32_per_unit": MAC32_G1_MUL, "achieved": n * MAC32_G1_MUL / (mbms * 1e-3) / 1e12,
                       "peak": peak / 1e12, "unit": "TMAC32/s", "frac": n * MAC32_G1_MUL / (mbms * 1e-3) / peak,
                       "executed_mac_per_unit": 2852 * 406, "executed_frac_of_peak": n * 2852 * 406 / (mbms * 1e-3) / peak,
                       "reference_algorithm_mac32_per_unit": MAC32_G1_MUL_REF, "algorithmic_bytes": n * (96 + 32 + 144), "traffic": static_traffic("g1_mul_batch")[0],
                       "traffic_source": static_traffic("g1_mul_batch")[1]}}
    if not args.no_cpu_baseline:
        from oracle import c_oracle
        mm = 1 << 12
        want_xy, want_inf = c_oracle.mul_batch_affine(1, xy1[:mm], None, sb[:mm], host_threads())
        got_xy, got_inf = ctx.batch_normalize(1, d_mo[:mm].cpu().numpy().view(np.uint64))
        mb["gpu_result_matches"] = bool(np.array_equal(got_xy, want_xy) and np.array_equal(got_inf, want_inf))
        if not mb["gpu_result_matches"]:
            raise SystemExit("bench: GPU mul_batch differs from the CPU oracle on the sample")
    # the same pairs with the caller vouching for the subgroup (blsgpu_set_assume_subgroup): endomorphism split, 32 windows of two additions
    fctx = bls.Context(torch.cuda.current_device())
    fctx.set_stream(torch.cuda.current_stream().cuda_stream)
    fctx.set_assume_subgroup(True)
    d_mo_f = torch.zeros((n, 18), dtype=torch.int64, device=dev)
    mbf = median_ms(lambda: fctx.mul_batch_device(1, d_xy1.data_ptr(), 0, d_scalars.data_ptr(), n, d_mo_f.data_ptr()), sync, warm=1, reps=3)
    MAC32_G1_MUL_GLV = 1860 * 300
    mb["vouched_subgroup"] = {"ms": mbf, "scalar_muls_per_s": n / (mbf * 1e-3),
                              "note": "k_mul_batch_glv: scalars split with the endomorphism, 128 doublings x 8 + 67 additions x 12 + 32 = 1 860 field multiplications per unit",
                              "roofline": {"bound": "int-valu", "kernel": "k_mul_batch_glv", "mac32_per_unit": MAC32_G1_MUL_GLV, "achieved": n * MAC32_G1_MUL_GLV / (mbf * 1e-3) / 1e12,
                                           "peak": peak / 1e12, "unit": "TMAC32/s", "frac": n * MAC32_G1_MUL_GLV / (mbf * 1e-3) / peak}}
    if not args.no_cpu_baseline:
        fa_xy, fa_inf = ctx.batch_normalize(1, d_mo_f[:mm].cpu().numpy().view(np.uint64))
        mb["vouched_subgroup"]["gpu_result_matches"] = bool(np.array_equal(fa_xy, want_xy) and np.array_equal(fa_inf, want_inf))
        if not mb["vouched_subgroup"]["gpu_result_matches"]:
            raise SystemExit("bench: GPU mul_batch (endomorphism path) differs from the CPU oracle on the sample")
    fctx.close()
    del d_mo_f
    extras["g1_mul_batch"] = mb
    n2m = min(1 << 18, n2)
    xy2m, _ = b2.download(0, n2m)
    d_xy2 = torch.from_numpy(xy2m.view(np.int64)).to(dev)
    d_mo2 = torch.zeros((n2m, 36), dtype=torch.int64, device=dev)
    mb2ms = median_ms(lambda: ctx.mul_batch_device(2, d_xy2.data_ptr(), 0, d_s2.data_ptr(), n2m, d_mo2.data_ptr()), sync, warm=1, reps=3)
    extras["g2_mul_batch"] = {"n": n2m, "ms": mb2ms, "scalar_muls_per_s": n2m / (mb2ms * 1e-3),
                              "roofline": {"bound": "int-valu", "kernel": "k_mul_batch<G2, lane pair>", "mac32_per_unit": MAC32_G2_MUL, "achieved": n2m * MAC32_G2_MUL / (mb2ms * 1e-3) / 1e12,
                                           "peak": peak / 1e12, "unit": "TMAC32/s", "frac": n2m * MAC32_G2_MUL / (mb2ms * 1e-3) / peak,
                                           "reference_algorithm_mac32_per_unit": MAC32_G2_MUL_REF, "algorithmic_bytes": n2m * (192 + 32 + 288),
                                           "traffic": static_traffic("g2_mul_batch")[0], "traffic_source": static_traffic("g2_mul_batch")[1]}}
    if not args.no_cpu_baseline:
        from oracle import c_oracle
        mm2 = 1 << 10
        want_xy2, want_inf2 = c_oracle.mul_batch_affine(2, xy2m[:mm2], None, sb[:mm2], host_threads())
        got_xy2, got_inf2 = ctx.batch_normalize(2, d_mo2[:mm2].cpu().numpy().view(np.uint64))
        extras["g2_mul_batch"]["gpu_result_matches"] = bool(np.array_equal(got_xy2, want_xy2) and np.array_equal(got_inf2, want_inf2))
        if not extras["g2_mul_batch"]["gpu_result_matches"]:
            raise SystemExit("bench: GPU G2 mul_batch differs from the CPU oracle on the sample")
    fctx2 = bls.Context(torch.cuda.current_device())
    fctx2.set_stream(torch.cuda.current_stream().cuda_stream)
    fctx2.set_assume_subgroup(True)
    d_mo2f = torch.zeros((n2m, 36), dtype=torch.int64, device=dev)
    mb2f = median_ms(lambda: fctx2.mul_batch_device(2, d_xy2.data_ptr(), 0, d_s2.data_ptr(), n2m, d_mo2f.data_ptr()), sync, warm=1, reps=3)
    MAC32_G2_MUL_GLS = int(MAC32_G2_MUL * (64 * 8 + 71 * 12 + 48) / 2852)      # 64 doublings + 71 additions + the psi images, in the units of MAC32_G2_MUL
    extras["g2_mul_batch"]["vouched_subgroup"] = {
        "ms": mb2f, "scalar_muls_per_s": n2m / (mb2f * 1e-3),
        "note": "k_mul_batch_gls: four 63-bit digits over psi, 16 windows of four additions (64 doublings + 71 additions per unit)",
        "roofline": {"bound": "int-valu", "kernel": "k_mul_batch_gls", "mac32_per_unit": MAC32_G2_MUL_GLS, "achieved": n2m * MAC32_G2_MUL_GLS / (mb2f * 1e-3) / 1e12,
                     "peak": peak / 1e12, "unit": "TMAC32/s", "frac": n2m * MAC32_G2_MUL_GLS / (mb2f * 1e-3) / peak}}
    if not args.no_cpu_baseline:
        fa2_xy, fa2_inf = ctx.batch_normalize(2, d_mo2f[:mm2].cpu().numpy().view(np.uint64))
        extras["g2_mul_batch"]["vouched_subgroup"]["gpu_result_matches"] = bool(np.array_equal(fa2_xy, want_xy2) and np.array_equal(fa2_inf, want_inf2))
        if not extras["g2_mul_batch"]["vouched_subgroup"]["gpu_result_matches"]:
            raise SystemExit("bench: GPU G2 mul_batch (psi path) differs from the CPU oracle on the sample")
    fctx2.close()
    del d_xy1, d_mo, d_xy2, d_mo2, d_mo2f
    # fixed-base mode: resident window-shifted tables (13 windows of 20 bits, one bucket set, no window combine)
    t1 = time.perf_counter()
    bases.precompute(0)
    pre_s = time.perf_counter() - t1
    d_o1 = [torch.zeros(18, dtype=torch.int64, device=dev) for _ in range(4)]
    ctx.set_pipelining(True)
    for i in range(8):
        ctx.msm_device(bases, d_scalars.data_ptr(), n, d_o1[i & 3].data_ptr())
    ctx.join(0); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(60):                      # (20 steps carried ~0.2 ms per step of pipeline fill and drain; rounds 2-4 quoted that)
        ctx.msm_device(bases, d_scalars.data_ptr(), n, d_o1[i & 3].data_ptr())
    ctx.join(0); torch.cuda.synchronize()
    pdt2 = (time.perf_counter() - t1) / 60
    ctx.set_pipelining(False)
    tab_single = median_ms(lambda: ctx.msm_device(bases, d_scalars.data_ptr(), n, d_o1[0].data_ptr()), sync)      # ONE call alone on the tables (SURVEY.md 8d protocol)
    same = bool(np.array_equal(ctx.batch_normalize(1, d_o1[3].cpu().numpy().view(np.uint64)[None, :])[0],
                               ctx.batch_normalize(1, d_out.cpu().numpy().view(np.uint64)[None, :])[0]))
    extras["g1_msm_precomputed_tables"] = {"scalar_muls_per_s": n / pdt2, "ms": 1e3 * pdt2, "single_call_ms": tab_single,
                                           "whole_msm_frac_single_call": n * MAC32_G1_MSM_2_20 / (tab_single * 1e-3) / peak, "table_build_s": pre_s, "window_bits": 20,
                                           "resident_bytes": 13 * n * 128, "matches_plain_path": same,
                                           "note": "optional mode for reused bases (blsgpu_bases_precompute); NOT the headline value"}
    # measured instruction-issue utilisation next to every canonical fraction (valu_issue)
    def _kms(block, name):
        v = ((block.get("timing") or {}).get("kernel_ms") or {}).get(name)
        return (v.get("total_ms") if isinstance(v, dict) else v)
    for blk, tag, ms in ((extras["pairing_batch"], "pairing", pms), (extras["multi_miller_loop"], "mml", _kms(extras["multi_miller_loop"], "k_multi_miller_shared") or mms),
                         (extras["verification_equations"], "equations", eqms),
                         (extras["bls_verify_from_bytes"], "bls_verify", vms), (extras.get("fr_ntt") or {}, "ntt_leg", (extras.get("fr_ntt") or {}).get("ms")),
                         (extras["hash_to_g2"], "hash_to_g2", hms), (extras["hash_to_g1"], "hash_to_g1", h1ms),
                         (extras["codec"]["g1"], "decode_g1", extras["codec"]["g1"]["decode_checked_ms"]), (extras["codec"]["g2"], "decode_g2", extras["codec"]["g2"]["decode_checked_ms"]),
                         (extras["g2_msm"], "g2_msm", 1e3 * g2dt), (extras["g1_mul_batch"], "g1_mul_batch", mbms), (extras["g2_mul_batch"], "g2_mul_batch", mb2ms)):
        if blk and ms and "roofline" in blk:
            blk["roofline"].update(valu_issue(tag, ms * 1e-3, peak))
    extras["gpu_state"] = {"before_extras": state_before, "after_extras": gpu_state(e.local_rank), "runtime_env": runtime_env(),
                           "note": "sysfs readings of the device the extras ran on (idle clocks before / after; clocks UNDER LOAD are in pairing_batch.clocks_under_load)"}
    return extras


# =====================================================================================================================
# workload "mixed" (BASELINE configs[4])
# =====================================================================================================================
class MixedJobs:
    """The three jobs of BASELINE configs[4] on ONE rank's shard, each on its own library context (own stream set), so the
    G1 MSM, the G2 MSM and the multi-Miller loop overlap on the GPU.  Also used by the single-GPU logical-rank test."""

    def __init__(self, bls, torch, device_index, sizes, rank, world, seed_base=None):
        from bls12_381_amd import synthetic
        from bls12_381_amd.distributed import shard_range
        self.bls, self.torch = bls, torch
        dev = torch.device("cuda", device_index)
        seed = synthetic.SEED if seed_base is None else seed_base
        self.ctx = [bls.Context(device_index) for _ in range(3)]
        # All three contexts take torch's current stream as THEIR stream: the MSM phases still run on each context's internal
        # streams (pipelining on), the Miller kernel on the shared stream beside them, and `join` orders the shared stream behind
        # the MSM tails -- so the collectives, the folds and the final exponentiation that follow are stream-ordered and a step
        # needs no host synchronisation at all (the buffers of step i+1 wait for the readers of step i through the same stream).
        st = torch.cuda.current_stream().cuda_stream
        for c in self.ctx:
            c.set_pipelining(True)
            c.set_stream(st)
        n1 = shard_range(1 << sizes[0], rank, world); n2 = shard_range(1 << sizes[1], rank, world); nm = shard_range(1 << sizes[2], rank, world)
        self.n = (n1[1] - n1[0], n2[1] - n2[0], nm[1] - nm[0])
        self.kb1 = synthetic.scalars(self.n[0], seed + 100 + rank); self.sb1 = synthetic.scalars(self.n[0], seed + 200 + rank)
        self.kb2 = synthetic.scalars(self.n[1], seed + 300 + rank); self.sb2 = synthetic.scalars(self.n[1], seed + 400 + rank)
        self.ka = synthetic.scalars(self.n[2], seed + 500 + rank); self.kq = synthetic.scalars(self.n[2], seed + 600 + rank)
        self.b1 = self.ctx[0].bases_from_scalars(1, self.kb1)
        self.b2 = self.ctx[1].bases_from_scalars(2, self.kb2)
        g1xy, _ = self.ctx[2].bases_from_scalars(1, self.ka).download()
        g2xy, _ = self.ctx[2].bases_from_scalars(2, self.kq).download()
        self.d_s1 = torch.from_numpy(self.sb1).to(dev); self.d_s2 = torch.from_numpy(self.sb2).to(dev)
        self.d_g1 = torch.from_numpy(g1xy.view(np.int64)).to(dev); self.d_g2 = torch.from_numpy(g2xy.view(np.int64)).to(dev)
        self.o1 = torch.zeros(18, dtype=torch.int64, device=dev)
        self.o2 = torch.zeros(36, dtype=torch.int64, device=dev)
        self.om = torch.zeros(72, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()

    def launch(self, which=(0, 1, 2)):
        """enqueue the rank-local part of the selected jobs; returns immediately (each context has its own streams)"""
        if 0 in which:
            self.ctx[0].msm_device(self.b1, self.d_s1.data_ptr(), self.n[0], self.o1.data_ptr())
        if 1 in which:
            self.ctx[1].msm_device(self.b2, self.d_s2.data_ptr(), self.n[1], self.o2.data_ptr())
        if 2 in which:
            self.ctx[2].multi_miller_loop_device(self.d_g1.data_ptr(), self.d_g2.data_ptr(), self.n[2], self.om.data_ptr())

    def join(self):
        for c in self.ctx[:2]:
            c.join(0)

    def sync(self):
        for c in self.ctx:
            c.synchronize()

    def partials(self):
        self.sync()
        return [t.cpu().numpy().view(np.uint64).copy() for t in (self.o1, self.o2, self.om)]

    def expected_scalars(self):
        """discrete-log side of the three identities, this rank's share: (sum s k over G1, over G2, sum a b)"""
        from bls12_381_amd import synthetic
        return (synthetic.dot_mod_r(self.kb1, self.sb1), synthetic.dot_mod_r(self.kb2, self.sb2), synthetic.dot_mod_r(self.ka, self.kq))


def run_mixed(args, e):
    torch, dist, bls = e.torch, e.dist, e.bls
    from bls12_381_amd.distributed import all_gather_rows
    rank, world, dev = e.rank, e.world, e.dev
    multi = e.multi
    steps = args.steps if args.steps is not None else 5
    warmup = args.warmup if args.warmup is not None else 1
    jobs = MixedJobs(bls, torch, e.local_rank, args.mixed_log, rank, world)
    fold_ctx = jobs.ctx[2]
    gath = [torch.zeros((world, w), dtype=torch.int64, device=e.xdev) for w in (18, 36, 72)] if multi else None
    f1 = torch.zeros(18, dtype=torch.int64, device=dev); f2 = torch.zeros(36, dtype=torch.int64, device=dev)
    fm = torch.zeros(72, dtype=torch.int64, device=dev); gt = torch.zeros(72, dtype=torch.int64, device=dev)

    def step(which=(0, 1, 2)):
        jobs.launch(which)
        jobs.join()                        # stream-level: the shared stream now follows the two MSM tails (no host round trip)
        srcm = jobs.om
        if multi:
            # the three tiny exchanges (144 B, 288 B, 576 B per rank) + folds on every rank
            for k, (buf, g, out, grp) in enumerate(((jobs.o1, gath[0], f1, 1), (jobs.o2, gath[1], f2, 2))):
                if k in which:
                    all_gather_rows(g, buf.to(e.xdev), dist)
                    gg = g if g.device == dev else g.to(dev)
                    fold_ctx.point_sum_device(grp, gg.data_ptr(), world, out.data_ptr())
            if 2 in which:
                all_gather_rows(gath[2], jobs.om.to(e.xdev), dist)
                gg = gath[2] if gath[2].device == dev else gath[2].to(dev)
                fold_ctx.fp12_product_device(gg.data_ptr(), world, fm.data_ptr())
                srcm = fm
        if 2 in which:
            fold_ctx.final_exponentiation_device(srcm.data_ptr(), 1, gt.data_ptr())     # ONE final exponentiation per product
        if e.xdev != dev:
            fold_ctx.synchronize()         # CPU collectives (gloo test path) read host copies: keep the steps apart

    def timed(which, k):
        fence(e)
        t0 = time.perf_counter()
        for _ in range(k):
            step(which)
        fence(e)
        return max_over_ranks(e, (time.perf_counter() - t0) / k)

    for _ in range(warmup):
        step()
    dt = timed((0, 1, 2), steps)
    alone = [timed((k,), max(1, steps // 2)) for k in range(3)]
    ok = True
    if multi:
        ok = ranks_agree(e, gt.cpu().numpy().view(np.uint64)) and ranks_agree(e, f1.cpu().numpy().view(np.uint64))
    if rank == 0:
        s = [1 << x for x in args.mixed_log]
        line = {
            "metric": "mixed workload wall time per instance (BASELINE configs[4])", "value": 1e3 * dt, "unit": "ms", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * dt, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 (14x28-bit limbs, 64-bit accumulators)", "data": "synthetic",
            "config": {"workload": "2^%d-point G1 MSM + 2^%d-point G2 MSM + 2^%d-term multi_miller_loop with its final exponentiation, sharded over %d MI355X, "
                                   "the three jobs overlapped on separate stream sets, one all-gather (144 / 288 / 576 B per rank) + fold each"
                                   % (args.mixed_log[0], args.mixed_log[1], args.mixed_log[2], world), "parallelism": "shard%d" % world},
            "jobs_alone_ms": {"g1_msm": 1e3 * alone[0], "g2_msm": 1e3 * alone[1], "multi_miller_loop+final_exp": 1e3 * alone[2]},
            "sum_of_jobs_alone_ms": 1e3 * sum(alone), "overlap_gain": sum(alone) / dt,
            "rates_overlapped": {"g1_scalar_muls_per_s": s[0] / dt, "g2_scalar_muls_per_s": s[1] / dt, "miller_terms_per_s": s[2] / dt},
            "ranks_agree": ok,
        }
        print(json.dumps(line))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


# =====================================================================================================================
# the same workload from ONE process: a device group of the C library (what a Rust host without torch.distributed uses)
# =====================================================================================================================
def group_measure(bls, torch, devices, total, steps, warmup, weak_n=None, check=True, expect_affine=None):
    """K pipelined sharded MSMs through blsgpu_g1_msm_sharded_device + blsgpu_g1_partials_fold ("enqueue MSM i, fold MSM i - 3"); returns the
    record of the run.  total = points of the ONE MSM split over the members (strong), or weak_n points per member."""
    import ctypes
    from bls12_381_amd import synthetic
    N = len(devices)
    g = bls.Group(devices)
    g.set_pipelining(True)
    n_all = weak_n * N if weak_n else total
    sizes = g.shard_sizes(n_all)
    kbs = [synthetic.scalars(sizes[k], synthetic.SEED + 2 * k + 1) for k in range(N)]
    sbs = [synthetic.scalars(sizes[k], synthetic.SEED + 2 * k) for k in range(N)]
    bases = g.bases_from_scalars(1, np.concatenate(kbs))
    d_s = [torch.from_numpy(sbs[k]).to(torch.device("cuda", devices[k])) for k in range(N)]
    d_o = [[torch.zeros(18, dtype=torch.int64, device=torch.device("cuda", devices[k])) for _ in range(8)] for k in range(N)]      # eight: see blsgpu_g1_partials_fold_device
    for d in set(devices):
        torch.cuda.synchronize(d)
    sp = [t.data_ptr() for t in d_s]
    d_fold = [torch.zeros(18, dtype=torch.int64, device=torch.device("cuda", devices[0])) for _ in range(8)]
    state = {"i": 0, "last": None}

    def step():
        i = state["i"]; state["i"] = i + 1
        g.msm_sharded_device(bases, sp, [d_o[k][i & 7].data_ptr() for k in range(N)])
        if i >= 3:           # the fold of MSM i - 3 (whose pipeline slot MSM i + 1 reuses)
            fold(i - 3, 3)

    # members on ONE device: the fold is queued behind the MSM on the members' streams (no host synchronisation in the step).  Members on
    # DIFFERENT devices: the synchronous fold through pinned host memory -- it waits only for MSM i - 3, which has long finished -- because the
    # cross-device form (peer copies + events between devices) has never run on real multi-GPU hardware (the build box has one GPU)
    device_fold = len(set(devices)) == 1

    def fold(j, lag):
        ptrs = [d_o[k][j & 7].data_ptr() for k in range(N)]
        if device_fold:
            g.partials_fold_device(1, ptrs, d_fold[j & 7].data_ptr(), lag=lag)
        else:
            state["last"] = g.partials_fold(1, ptrs, lag=lag)

    def drain():
        i = state["i"]
        for j in range(max(0, i - 3), i):
            fold(j, i - 1 - j)
        g.synchronize()
        if i and device_fold:
            state["last"] = d_fold[(i - 1) & 7].cpu().numpy().view(np.uint64).copy()
        state["i"] = 0

    for _ in range(warmup):
        step()
    drain()
    lib = g.lib
    avg, cnt = ctypes.c_double(), ctypes.c_uint()
    bls._lib.check(lib.blsgpu_msm_accumulate_stats(g.member_ctx(0), 3, ctypes.byref(avg), ctypes.byref(cnt)), "msm_accumulate_stats")
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    drain()
    dt = time.perf_counter() - t0
    bls._lib.check(lib.blsgpu_msm_accumulate_stats(g.member_ctx(0), 0, ctypes.byref(avg), ctypes.byref(cnt)), "msm_accumulate_stats")
    rec = {"members": N, "devices": list(devices), "fold": "device (queued on the members' streams)" if device_fold else "host (pinned staging, lag 3)", "distinct_gpus": len(set(devices)), "total_points": n_all, "points_per_member": sizes[0], "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * dt / steps, "value": float(n_all) * steps / dt, "member0_accumulate_launch_ms": avg.value, "member0_launches_timed": int(cnt.value)}
    if expect_affine is not None:
        # the same shards, seeds and sizes as the one-process-per-GPU run that precedes this measurement: the two folded sums must agree
        c0 = bls.Context(devices[0])
        got = c0.batch_normalize(1, state["last"][None, :])
        rec["result_matches"] = bool(np.array_equal(got[0][0], expect_affine))
        rec["checked_against"] = "the folded result of the RCCL path"
        c0.close()
    elif check:
        # discrete-log identity: sum_i s_i [k_i]G = [sum_i s_i k_i] G  (Python big integers: about a minute for 2^24 terms)
        tot = sum(synthetic.dot_mod_r(kbs[k], sbs[k]) for k in range(N)) % synthetic.R_ORDER
        c0 = bls.Context(devices[0])
        want = c0.bases_from_scalars(1, [tot]).download()
        got = c0.batch_normalize(1, state["last"][None, :])
        rec["result_matches"] = bool(np.array_equal(got[0][0], want[0][0]) and got[1][0] == want[1][0])
        rec["checked_against"] = "[sum s_i k_i] G"
        c0.close()
    bases.free(); g.close()
    return rec


def group_child(args, world, strong, expect_affine, limit_s=420):
    """`bench.py --group <world>` as a child process with a time limit; returns its group_path record (or an error record)"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--group", str(world), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-extras", "--log-total", str(args.log_total)]
    if not strong:
        cmd += ["--weak"] + (["--log-n", str(args.log_n)] if args.log_n is not None else [])
    elif world == 1:                                        # --dist-single: the ONE "shard" is the whole MSM (a group of one is never "strong")
        cmd += ["--weak", "--log-n", str(args.log_total)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
           and not k.startswith("TORCHELASTIC_")}
    if expect_affine is not None:
        env["BENCH_EXPECT_AFFINE"] = np.ascontiguousarray(expect_affine, dtype=np.uint64).tobytes().hex()
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit_s)
    except subprocess.TimeoutExpired:
        return {"error": "bench.py --group %d did not finish within %d s" % (world, limit_s)}
    out = r.stdout.decode(errors="replace").strip().splitlines()
    if r.returncode != 0 or not out:
        return {"error": ("bench.py --group %d exited with %d: " % (world, r.returncode)) + r.stderr.decode(errors="replace")[-160:]}
    try:
        rec = json.loads(out[-1]).get("group_path") or {"error": "no group_path in the child's line"}
    except ValueError:
        rec = {"error": "unparsable child line: " + out[-1][:120]}
    rec["run_as"] = "child process of rank 0 (bench.py --group %d), the other ranks idle at a CPU-side barrier" % world
    return rec


def run_group(args):
    import torch
    import bls12_381_amd as bls
    N = args.group
    ndev = torch.cuda.device_count()
    devices = list(range(N)) if (ndev >= N and not args.same_device) else [0] * N
    steps = args.steps if args.steps is not None else (100 if N == 1 else 20)
    warmup = args.warmup if args.warmup is not None else 5
    strong = N > 1 and not args.weak
    log_n = args.log_n if args.log_n is not None else 20
    total = (1 << args.log_total) if strong else (1 << log_n) * N
    expect = os.environ.get("BENCH_EXPECT_AFFINE")          # set by group_child(): the folded result of the RCCL run on the same shards
    expect = np.frombuffer(bytes.fromhex(expect), dtype=np.uint64).copy() if expect else None
    rec = group_measure(bls, torch, devices, total, steps, warmup, weak_n=None if strong else (1 << log_n), expect_affine=expect)
    if not rec.get("result_matches", True):
        raise SystemExit("bench --group: the folded MSM result is wrong")
    ctx = bls.Context(devices[0])
    peak = max(ctx.mad_throughput(2000) for _ in range(3))
    windows = (256 + WINDOW_BITS - 1) // WINDOW_BITS
    mac = float(rec["points_per_member"]) * windows * MAC32_G1_ADD
    dur = rec["member0_accumulate_launch_ms"] * 1e-3
    roof = {"bound": "int-valu", "kernel": "k_msm_accumulate<G1> (member 0)", "achieved": mac / dur / 1e12 if dur else None, "peak": peak / 1e12, "unit": "TMAC32/s",
            "frac": mac / dur / peak if dur else None, "traffic": None, "launch_ms": rec["member0_accumulate_launch_ms"]}
    workload = ("ONE 2^%d-point G1 MSM sharded over %d members of a blsgpu_group in ONE process (%d distinct GPUs; %d points per member), bases and scalars resident in HBM; "
                "one folded result per step" % (args.log_total, N, rec["distinct_gpus"], rec["points_per_member"])) if strong else (
                "2^%d-point G1 MSM per member of a blsgpu_group of %d in ONE process (%d distinct GPUs), bases and scalars resident in HBM; one folded result per step"
                % (log_n, N, rec["distinct_gpus"]))
    line = {"metric": "G1 MSM throughput (scalar-muls/sec) at 2^%d points" % (args.log_total if strong else log_n) + ("" if strong or N == 1 else " per member"),
            "value": rec["value"], "unit": "scalar-muls/s", "n_gpus": rec["distinct_gpus"], "steps": steps, "warmup": warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u32 (14x28-bit limbs, 64-bit accumulators)", "data": "synthetic",
            "config": {"workload": workload, "points_per_gpu": rec["points_per_member"], "total_points": rec["total_points"], "parallelism": "group%d" % N, "members": N,
                       "devices": rec["devices"]},
            "roofline": roof, "cpu_baseline": None, "group_path": rec}
    emit(line)


def main():
    args = parse()
    if args.group:
        return run_group(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    e = setup(args)
    if args.workload == "mixed":
        run_mixed(args, e)
    else:
        run_msm(args, e)


if __name__ == "__main__":
    main()
