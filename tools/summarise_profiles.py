#!/usr/bin/env python3
"""gpurun_out/prof (tools/collect_profiles.sh) -> profiles/r01_msm_kernel_stats.md, r01_msm_pmc.{md,json}, r01_pairing_pmc.md"""
import csv, glob, json, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")
OUT = os.path.join(ROOT, "profiles")

def one(pattern):
    f = glob.glob(os.path.join(P, pattern), recursive=True)
    return f[0] if f else None

def stats_table(path, title, out):
    rows = list(csv.DictReader(open(path)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows) or 1
    with open(out, "w") as fh:
        fh.write(f"# {title}\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for r in rows:
            fh.write(f"| `{r['Name'][:100]}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {int(r['MinNs'])/1e3:.1f} | {int(r['MaxNs'])/1e3:.1f} | {100*int(r['TotalDurationNs'])/tot:.1f} |\n")

def counters(dirpat, kernel):
    acc = collections.defaultdict(list)
    f = one(dirpat + "/**/*counter_collection.csv")
    if not f: return {}
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: sum(v) / len(v) for k, v in acc.items()}
    out["_launches"] = max([len(v) for v in acc.values()] or [0])
    return out

def duration(dirpat, kernel):
    f = one(dirpat + "/**/*kernel_trace.csv")
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
    return sum(d) / len(d), len(d)

stats_table(one("stats/**/*kernel_stats.csv"), "r01: rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --no-extras\n\nMI355X, 2^20-point G1 MSM per step, pipelined (4 slots); 5 warm-up + 100 timed MSMs + 5 isolated roofline/phase probes = 110 launches of each per-MSM kernel.", os.path.join(OUT, "r01_msm_kernel_stats.md"))
K = "k_msm_accumulate<bls::FpPolicy>"
dur, nl = duration("stats", K)
def durations_in_order(dirpat, kernel):
    f = one(dirpat + "/**/*kernel_trace.csv")
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"])
    return [d for _, d in rows]
_d = durations_in_order("stats", K)
dur_pipe = sum(_d[5:-5]) / max(1, len(_d[5:-5]))      # the timed steps of the bench (after 5 warm-up launches): pipelined, overlapped with sort / tails
dur_iso = sum(_d[-5:]) / max(1, len(_d[-5:]))         # the five isolated roofline / phase probes at the end
try:
    _b = json.loads(open(os.path.join(P, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
    bench_note = f"the bench line printed by this very run: launch_ms = {_b['roofline']['launch_ms']:.3f}, launch_ms_isolated = {_b['roofline']['launch_ms_isolated']:.3f}, frac = {_b['roofline']['frac']:.3f}"
except Exception:
    bench_note = "bench line of the traced run not available"
fs, ws, sq = counters("pmc_FETCH_SIZE", K), counters("pmc_WRITE_SIZE", K), counters("pmc_SQ_INSTS_VALU", K)
fetch_kb, write_kb = fs.get("FETCH_SIZE", 0), ws.get("WRITE_SIZE", 0)
hbm = (2 * fetch_kb + write_kb) * 1024
n, W = 1 << 20, 16
alg = W * n * (128 + 4) + (1 << 19) * 176
mads = W * n * (7 * 406 + 2 * 315 + 602)          # 7 mul + 2 sqr + one 2-product sum per mixed addition
j = {"kernel": K, "workload": "2^20-point G1 MSM, c=16", "launch_avg_ns": dur, "launches": nl, "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb,
     "hbm_bytes_per_launch_corrected": hbm,
     "correction": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 16-byte-per-lane reads at half); Infinity-Cache hits are included",
     "algorithmic_bytes_per_launch": alg, "sq": {k: v for k, v in sq.items() if not k.startswith("_")}, "valu_wave_insts": sq.get("SQ_INSTS_VALU"), "mad_wave_insts_expected": mads / 64}
json.dump(j, open(os.path.join(OUT, "r01_msm_pmc.json"), "w"), indent=1)
ghz = sq.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur * 1e-9) / 1e9 if dur else 0
with open(os.path.join(OUT, "r01_msm_pmc.md"), "w") as fh:
    fh.write(f"""# r01: PMC counters of the dominant kernel ({K}, 2^20 points, c = 16)

Separate rocprofv3 passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`, each with `--kernel-trace` only) over `python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras` (tools/collect_profiles.sh); counter averages over the {fs.get('_launches', 0)} launches of those passes.

* launch duration (kernel trace of the default bench command): {dur/1e3:.0f} us over all {nl} launches = {dur_pipe/1e3:.0f} us for the {len(_d) - 10} timed pipelined launches (what `bench.py` reports as `roofline.launch_ms`) and {dur_iso/1e3:.0f} us for the 5 isolated probes at the end (`launch_ms_isolated`); {bench_note}
* FETCH_SIZE = {fetch_kb:.0f} KB raw -> x2 (gfx950 half-count of 16-byte-per-lane reads) = {2*fetch_kb*1024/1e9:.2f} GB;  WRITE_SIZE = {write_kb:.0f} KB = {write_kb*1024/1e9:.2f} GB
* HBM-side traffic per launch = {hbm/1e9:.2f} GB (algorithmic: 16 windows x 2^20 gathers x (128 B record + 4 B index) + 2^19 x 176 B bucket records = {alg/1e9:.2f} GB); at {dur/1e3:.0f} us that is {hbm/dur:.2f} GB/ms = {hbm/dur/1e3:.2f} TB/s = {100*hbm/dur/1e3/8:.0f}% of the 8 TB/s HBM peak: not memory-bound (the 128 MB of resident bases also fit the 256 MB Infinity Cache).
* SQ_INSTS_VALU = {sq.get('SQ_INSTS_VALU',0):.3e} wave-instructions, of which {mads/64:.3e} are the v_mad_u64_u32 of the 7 mul + 2 sqr + one two-product sum per mixed addition ({100*mads/64/max(sq.get('SQ_INSTS_VALU',1),1):.0f}%).
* GRBM_GUI_ACTIVE = {sq.get('GRBM_GUI_ACTIVE',0):.3e} (summed over 8 XCDs) -> {ghz:.2f} GHz effective clock under profiling.
""")
# pairing
KP = "k_pairing"
pd, pn = duration("pair_stats", KP)
pf, pw, ps = counters("pair_pmc_FETCH_SIZE", KP), counters("pair_pmc_WRITE_SIZE", KP), counters("pair_pmc_SQ_INSTS_VALU", KP)
stats_table(one("pair_stats/**/*kernel_stats.csv"), "r01: rocprofv3 --kernel-trace --stats -- python tools/quick_pair3.py  (3 launches of 2^16 pairings)", os.path.join(OUT, "r01_pairing_kernel_stats.md"))
with open(os.path.join(OUT, "r01_pairing_pmc.md"), "w") as fh:
    fh.write(f"""# r01: PMC counters of k_pairing (2^16 pairings per launch, lane-pair layout, 2 wavefronts/SIMD)

`tools/quick_pair3.py` under rocprofv3 (`--kernel-trace` + one `--pmc` group per pass; tools/collect_profiles.sh); averages over {pn} launches.

* launch duration: {pd/1e6:.2f} ms -> {65536/(pd*1e-9):.3e} pairings/s
* FETCH_SIZE = {pf.get('FETCH_SIZE',0)/1e6:.2f} GB raw (x2 if counted as 16-byte reads: {2*pf.get('FETCH_SIZE',0)*1024/1e9:.1f} GB), WRITE_SIZE = {pw.get('WRITE_SIZE',0)*1024/1e9:.1f} GB: the per-lane scratch traffic of the out-of-line Fp12 routines ({(2*pf.get('FETCH_SIZE',0)+pw.get('WRITE_SIZE',0))*1024/pd/1e3:.2f} TB/s).
* SQ_INSTS_VALU = {ps.get('SQ_INSTS_VALU',0):.3e} wave-instructions = {ps.get('SQ_INSTS_VALU',0)/2048:.3e} per wavefront; ~2.97e6 of them are v_mad_u64_u32 (16 k field multiplications per pairing over two lanes).
* SQ_WAVE_CYCLES = {ps.get('SQ_WAVE_CYCLES',0):.3e}, SQ_BUSY_CYCLES = {ps.get('SQ_BUSY_CYCLES',0):.3e}, GRBM_GUI_ACTIVE = {ps.get('GRBM_GUI_ACTIVE',0):.3e}
""")
print("profiles written")
