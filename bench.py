#!/usr/bin/env python3
"""bench.py -- G1 MSM throughput (scalar-muls/s) on MI355X, the headline metric of BASELINE.json.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step is ONE multi-scalar multiplication over this rank's shard of synthetic input that is already
resident in HBM: 2^20 affine G1 bases (library-resident, internal form) and 2^20 32-byte scalars per GPU
(BASELINE configs[1]; weak scaling: N GPUs compute one N*2^20-point MSM).  For N > 1 every step ends with
the path's single exchange: an RCCL all-gather of the per-rank partial sums (144 B each) followed by the
fold on every rank.  K steps are timed between barrier + synchronize pairs; the reported
time is the max over ranks; `value` = total scalar-muls of all ranks / that time.

The JSON line also carries
  roofline      the dominant kernel (bucket accumulation) against the integer-VALU roofline: canonical
                MAC32 per launch / HIP-event launch time, peak = v_mad_u64_u32 rate measured live
  cpu_baseline  the C restatement of the reference's own `sum(P_i * s_i)` (oracle/bls_oracle.c) timed on the
                host cores over a bounded sample of the same workload, and checked against the GPU result
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 20
WINDOW_BITS = 16          # canonical c of SURVEY.md 8d; the library picks its own c


def synth_inputs(n, seed):
    """Seeded synthetic input: k_i (base = [k_i]G1) and s_i, 254-bit uniform (top two bits cleared, so < r)."""
    rs = np.random.RandomState(seed)
    kb = rs.randint(0, 256, size=(n, 32), dtype=np.uint8)
    kb[:, 31] &= 0x3F
    sb = rs.randint(0, 256, size=(n, 32), dtype=np.uint8)
    sb[:, 31] &= 0x3F
    return kb, sb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-n", type=int, default=LOG_N, help="log2 of points per GPU (default 20 = BASELINE configs[1])")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for single-GPU plumbing tests)")
    ap.add_argument("--same-device", action="store_true", help="testing aid: all ranks use GPU 0 (needs --backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (batched pairings, G2 MSM)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import bls12_381_amd as bls

    ctx = bls.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 1 << args.log_n
    kb, sb = synth_inputs(n, 0xB1512381 + rank)
    bases = ctx.bases_from_scalars(1, kb)                       # resident bases: [k_i] G1 built on the device
    d_scalars = torch.from_numpy(sb).to(dev)
    d_out = [torch.zeros(18, dtype=torch.int64, device=dev) for _ in range(4)]      # up to four calls may be in flight
    xdev = dev if args.backend == "nccl" else torch.device("cpu")       # where the exchanged partials live
    gathered = torch.zeros((world, 18), dtype=torch.int64, device=xdev) if world > 1 else None
    d_fold = torch.zeros(18, dtype=torch.int64, device=dev)
    ctx.set_pipelining(True)         # the latency-bound tail of MSM i overlaps the chip-filling phases of MSM i+1
    state = {"i": 0}

    def exchange(buf):
        """the path's single exchange step: all-gather the per-rank partial sums, fold on every rank"""
        dist.all_gather(list(gathered.unbind(0)), buf.to(xdev))     # rows of one (world, 18) tensor
        g = gathered if gathered.device == dev else gathered.to(dev)
        ctx.point_sum_device(1, g.data_ptr(), world, d_fold.data_ptr())     # asynchronous fold on this rank's GPU
        state["g"] = g

    def step():
        i = state["i"]; state["i"] = i + 1
        ctx.msm_device(bases, d_scalars.data_ptr(), n, d_out[i & 3].data_ptr())
        if world > 1 and i >= 2:
            # consume result i-2: by now its tail has long finished, so the wait (and the exchange queued behind it on this
            # stream) does not hold back the front of MSM i+1, whose dependency on this stream is recorded at its launch
            ctx.join(2)
            exchange(d_out[(i - 2) & 3])

    def drain():
        ctx.join(0)
        if world > 1:
            for k in range(max(0, state["i"] - 2), state["i"]):
                exchange(d_out[k & 3])
        state["i"] = 0

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    fence()
    ctx.msm_accumulate_stats(True)             # HIP events around every launch of the dominant kernel inside the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    fence()
    dt = time.perf_counter() - t0
    live_acc_ms, live_acc_n = ctx.msm_accumulate_stats(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # every rank must hold the same folded result: compare canonical affine limbs through a max/min all-reduce
        last = d_fold.cpu().numpy().view(np.uint64)
        aff = torch.from_numpy(ctx.batch_normalize(1, last[None, :])[0][0].view(np.int64).copy()).to(xdev)
        hi, lo = aff.clone(), aff.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX); dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        if not bool(torch.equal(hi, lo)):
            raise SystemExit("bench: ranks disagree on the folded MSM result")
    ctx.set_pipelining(False)
    d_out = d_out[0]

    # ---- roofline of the dominant kernel, measured live with HIP events on the library's stream ----------
    roof = None
    phases = None
    if rank == 0:
        peak = ctx.mad_throughput(2000)                          # v_mad_u64_u32 lane-ops/s = MAC32/s
        fp_rate = ctx.fp_mul_throughput(2000)
        ctx.set_profiling(True)
        acc_ms, tot_ms = [], []
        for _ in range(5):
            ctx.msm_device(bases, d_scalars.data_ptr(), n, d_out.data_ptr())
            ph = ctx.last_msm_phase_ms()
            acc_ms.append(ph["accumulate"]); tot_ms.append(ph["total"]); phases = ph
        ctx.set_profiling(False)
        windows = (256 + WINDOW_BITS - 1) // WINDOW_BITS
        mac32_per_launch = float(n) * windows * 11 * 300          # canonical: one complete mixed add per (point, window)
        # duration of the dominant kernel: average over its launches INSIDE the timed (pipelined) region, HIP events on the
        # stream it runs on; the isolated (one MSM at a time) duration is reported next to it
        dur = (live_acc_ms if live_acc_n else float(np.mean(acc_ms))) * 1e-3
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_msm_pmc.json")
        if args.log_n == 20 and os.path.exists(pmc_path):
            # HBM-side bytes per launch of this kernel on this workload, from separate rocprofv3 --pmc passes
            # (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE); see profiles/r01_msm_pmc.md
            traffic = json.load(open(pmc_path))["hbm_bytes_per_launch_corrected"]
        roof = {
            "bound": "int-valu", "kernel": "k_msm_accumulate<G1>",
            "achieved": mac32_per_launch / dur / 1e12, "peak": peak / 1e12, "unit": "TMAC32/s",
            "frac": mac32_per_launch / dur / peak, "frac_isolated": mac32_per_launch / (float(np.mean(acc_ms)) * 1e-3) / peak, "traffic": traffic,
            "launch_ms": dur * 1e3, "launches_timed": int(live_acc_n), "launch_ms_isolated": float(np.mean(acc_ms)), "mac32_per_launch": mac32_per_launch,
            "fp_mul_per_s_chain": fp_rate,
            "whole_msm_frac": (float(n) * 188 * 300) / (float(np.mean(tot_ms)) * 1e-3) / peak,
            "note": "integer-VALU bound (no MFMA, HBM traffic ~13% of peak, see traffic): canonical 300 MAC32 per Fp mul, 11 Fp mul per mixed add, "
                    "16 windows (SURVEY.md 8d); peak = v_mad_u64_u32 issue rate measured in this run",
        }

    # ---- CPU baseline: the reference's own definition on the host cores (bounded sample) ------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle
        m = min(n, 1 << 15)
        xy, inf = bases.download(0, m)
        t1 = time.perf_counter()
        ref, used = c_oracle.g1_msm(xy, inf, sb[:m], 0)
        cdt = time.perf_counter() - t1
        got = ctx.msm(bases, sb[:m])
        same = bool(np.array_equal(ctx.batch_normalize(1, got[None, :])[0][0], c_oracle.g1_to_affine(ref)[0]))
        t1 = time.perf_counter()
        c_oracle.g1_msm(xy[:256], inf[:256], sb[:256], 1)
        one = 256 / (time.perf_counter() - t1)
        cpu = {"value": m / cdt, "unit": "scalar-muls/s", "cores": used, "kind": "port",
               "sample": f"first 2^{int(np.log2(m))} (point, scalar) pairs of the same workload: sum(P_i*s_i) by 255-step double-and-add + Sum, "
                         f"C restatement of the reference algorithm (oracle/bls_oracle.c), OpenMP over {used} threads; single thread: {one:.0f}/s",
               "single_thread_value": one, "parallel_speedup": (m / cdt) / one, "gpu_result_matches": same}
        if not same:
            raise SystemExit("bench: GPU MSM over the CPU sample differs from the oracle")

    # ---- secondary measurements of the same path (BASELINE configs[2]); never part of `value` ---------------
    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = {}
        np_ = 1 << 16
        rs = np.random.RandomState(99)
        ka = rs.randint(0, 256, size=(np_, 32), dtype=np.uint8); ka[:, 31] &= 0x3F
        kq = rs.randint(0, 256, size=(np_, 32), dtype=np.uint8); kq[:, 31] &= 0x3F
        g1xy, g1f = ctx.bases_from_scalars(1, ka).download()
        g2xy, g2f = ctx.bases_from_scalars(2, kq).download()
        d_g1 = torch.from_numpy(g1xy.view(np.int64)).to(dev); d_g2 = torch.from_numpy(g2xy.view(np.int64)).to(dev)
        d_gt = torch.zeros((np_, 72), dtype=torch.int64, device=dev)
        def pair():
            bls._lib.check(ctx.lib.blsgpu_pairing_batch_device(ctx.h, d_g1.data_ptr(), None, d_g2.data_ptr(), None, np_, d_gt.data_ptr()), "pairing_batch_device")
        pair(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            pair()
        torch.cuda.synchronize()
        pdt = (time.perf_counter() - t1) / 3
        extras["pairings_per_s"] = np_ / pdt
        if not args.no_cpu_baseline:
            # the reference's pairing on the host cores (C restatement, oracle/bls_oracle.c) on a bounded sample, and an exact
            # comparison of the GPU results over that sample
            from oracle import c_oracle
            mp = 1 << 13
            t1 = time.perf_counter()
            cref, cused = c_oracle.pairing_batch(0, g1xy[:mp], g1f[:mp], g2xy[:mp], g2f[:mp], 0)
            cpdt = time.perf_counter() - t1
            t1 = time.perf_counter()
            c_oracle.pairing_batch(0, g1xy[:16], g1f[:16], g2xy[:16], g2f[:16], 1)
            one_p = 16 / (time.perf_counter() - t1)
            same_p = bool(np.array_equal(d_gt[:mp].cpu().numpy().view(np.uint64), cref))
            extras["cpu_baseline_pairing"] = {"value": mp / cpdt, "unit": "pairings/s", "cores": cused, "kind": "port",
                                              "sample": f"first 2^13 of the same pairs, C restatement of pairings.rs (Miller loop + final exponentiation), "
                                                        f"OpenMP over {cused} threads; single thread: {one_p:.0f}/s",
                                              "single_thread_value": one_p, "parallel_speedup": (mp / cpdt) / one_p, "gpu_result_matches": same_p}
            if not same_p:
                raise SystemExit("bench: GPU pairings differ from the CPU oracle on the sample")
        extras["pairing_batch"] = {"n": np_, "ms": 1e3 * pdt, "note": "2^16 independent pairing(P_i, Q_i), inputs and outputs in HBM",
                                   "frac_of_fp_mul_chain_rate": (np_ * 16000 / pdt) / fp_rate}
        # multi_miller_loop at BASELINE configs[4]'s size (2^18 terms, the 2^16 pairs tiled four times): one shared accumulator
        # per four terms, partial products multiplied up; no final exponentiation in the timed region
        nm = 4 * np_
        d_g1m, d_g2m = d_g1.repeat(4, 1), d_g2.repeat(4, 1)
        def mml():
            bls._lib.check(ctx.lib.blsgpu_multi_miller_loop_device(ctx.h, d_g1m.data_ptr(), None, d_g2m.data_ptr(), None, nm, d_gt.data_ptr()), "multi_miller_loop_device")
        mml(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            mml()
        torch.cuda.synchronize()
        mdt = (time.perf_counter() - t1) / 3
        extras["multi_miller_loop_terms_per_s"] = nm / mdt
        extras["multi_miller_loop"] = {"n": nm, "ms": 1e3 * mdt, "note": "one product of 2^18 Miller values (no final exponentiation)"}
        del d_g1m, d_g2m
        # Fr transform of the MSM's scalar vector
        d_fr = d_scalars.clone()
        ctx.fr_ntt_device(d_fr.data_ptr(), args.log_n, False); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            ctx.fr_ntt_device(d_fr.data_ptr(), args.log_n, False)
        torch.cuda.synchronize()
        ndt = (time.perf_counter() - t1) / 10
        extras["fr_ntt"] = {"log_n": args.log_n, "ms": 1e3 * ndt, "elements_per_s": n / ndt,
                            "note": "radix-2 NTT over the scalar field, in place, natural order; 5 radix-4 passes over the data + one LDS pass at 2^20"}
        del d_fr
        # hash-to-curve in front of the pairings: 2^16 32-byte messages -> G2
        hm = torch.from_numpy(rs.randint(0, 256, size=np_ * 32, dtype=np.uint8)).to(dev)
        ho = torch.arange(0, (np_ + 1) * 32, 32, dtype=torch.int64, device=dev)
        hdst = b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_"
        hd = torch.from_numpy(np.frombuffer(hdst, dtype=np.uint8).copy()).to(dev)
        hout = torch.zeros((np_, 36), dtype=torch.int64, device=dev)
        def h2c():
            bls._lib.check(ctx.lib.blsgpu_hash_to_curve_device(ctx.h, 2, hm.data_ptr(), ho.data_ptr(), np_, hd.data_ptr(), len(hdst), 0, hout.data_ptr()), "hash_to_curve_device")
        h2c(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            h2c()
        torch.cuda.synchronize()
        hdt = (time.perf_counter() - t1) / 3
        extras["hash_to_g2"] = {"n": np_, "ms": 1e3 * hdt, "hashes_per_s": np_ / hdt, "note": "hash_to_curve (XMD:SHA-256, SSWU, RO) of 32-byte messages to G2"}
        del hm, ho, hout
        n2 = min(1 << 20, n)
        k2 = rs.randint(0, 256, size=(n2, 32), dtype=np.uint8); k2[:, 31] &= 0x3F
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
        if not args.no_cpu_baseline:
            from oracle import c_oracle
            m2 = 1 << 12                                  # bounded sample: ~10 s of CPU work
            xy2, inf2 = b2.download(0, m2)
            t1 = time.perf_counter()
            ref2, used2 = c_oracle.g2_msm(xy2, inf2, sb[:m2], 0)
            c2dt = time.perf_counter() - t1
            got2 = ctx.msm(b2, sb[:m2])
            same2 = bool(np.array_equal(ctx.batch_normalize(2, got2[None, :])[0][0], c_oracle.g2_to_affine(ref2)[0]))
            extras["cpu_baseline_g2_msm"] = {"value": m2 / c2dt, "unit": "scalar-muls/s", "cores": used2, "kind": "port",
                                             "sample": "first 2^12 (point, scalar) pairs: sum(P_i*s_i) over G2 by 255-step double-and-add + Sum "
                                                       f"(oracle/bls_oracle.c), OpenMP over {used2} threads", "gpu_result_matches": same2}
            if not same2:
                raise SystemExit("bench: GPU G2 MSM differs from the CPU oracle on the sample")
        extras["g2_msm_scalar_muls_per_s"] = n2 / g2dt
        extras["g2_msm"] = {"n": n2, "ms": 1e3 * g2dt}
        # fixed-base mode: resident window-shifted tables (13 windows of 20 bits, one bucket set, no window combine)
        t1 = time.perf_counter()
        bases.precompute(0)
        pre_s = time.perf_counter() - t1
        d_o1 = [torch.zeros(18, dtype=torch.int64, device=dev) for _ in range(4)]
        ctx.set_pipelining(True)
        for i in range(3):
            ctx.msm_device(bases, d_scalars.data_ptr(), n, d_o1[i].data_ptr())
        ctx.join(0); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(20):
            ctx.msm_device(bases, d_scalars.data_ptr(), n, d_o1[i & 3].data_ptr())
        ctx.join(0); torch.cuda.synchronize()
        pdt2 = (time.perf_counter() - t1) / 20
        ctx.set_pipelining(False)
        same = bool(np.array_equal(ctx.batch_normalize(1, d_o1[3].cpu().numpy().view(np.uint64)[None, :])[0],
                                   ctx.batch_normalize(1, d_out.cpu().numpy().view(np.uint64)[None, :])[0]))
        extras["g1_msm_precomputed_tables"] = {"scalar_muls_per_s": n / pdt2, "ms": 1e3 * pdt2, "table_build_s": pre_s, "window_bits": 20,
                                               "resident_bytes": 13 * n * 128, "matches_plain_path": same,
                                               "note": "optional mode for reused bases (blsgpu_bases_precompute); NOT the headline value"}

    if rank == 0:
        total = float(n) * world * args.steps
        line = {
            "metric": "G1 MSM throughput (scalar-muls/sec) at 2^%d points per GPU" % args.log_n,
            "value": total / dt, "unit": "scalar-muls/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (14x28-bit limbs, 64-bit accumulators)", "data": "synthetic",
            "config": {"workload": "2^%d-point G1 MSM per MI355X, bases resident in HBM, scalars in HBM; one result per step "
                                   "(N>1: RCCL all-gather of N partial sums + fold)" % args.log_n,
                       "points_per_gpu": n, "total_points": n * world, "parallelism": "shard%d" % world},
            "roofline": roof, "cpu_baseline": cpu, "msm_phase_ms": phases, "extras": extras,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
