#!/usr/bin/env python3
"""gpurun_out/prof_{msm,pair,mml} (tools/collect_profiles.sh) -> profiles/<round>_*.md / .json

    python tools/summarise_profiles.py r03
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_meta import kernel_meta, find as meta_find      # registers / scratch / LDS from the code object, not from rocprof's VGPR_Count column
META = kernel_meta()
G = os.path.join(ROOT, "gpurun_out")
OUT = os.path.join(ROOT, "profiles")
RND = sys.argv[1] if len(sys.argv) > 1 else "r03"
WRITTEN = []                    # tags of the *_pmc.json files this run writes (only those get a source digest)


def one(pattern):
    f = glob.glob(os.path.join(G, pattern), recursive=True)
    return max(f, key=os.path.getmtime) if f else None        # gpurun merges every run into gpurun_out/: take the newest


def stats_table(path, title, out):
    rows = list(csv.DictReader(open(path)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows) or 1
    with open(out, "w") as fh:
        fh.write(f"# {title}\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for r in rows:
            fh.write(f"| `{r['Name'][:100]}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {int(r['MinNs'])/1e3:.1f} | {int(r['MaxNs'])/1e3:.1f} | {100*int(r['TotalDurationNs'])/tot:.1f} |\n")


def counters(d, kernel):
    """averages per launch of every counter collected for `kernel` under gpurun_out/<d>/pmc_*"""
    res = {}
    files = {}
    for f in glob.glob(os.path.join(G, d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        k = f[len(os.path.join(G, d)) + 1:].split(os.sep)[0]               # one file per pmc_<i> directory: the newest
        if k not in files or os.path.getmtime(f) > os.path.getmtime(files[k]):
            files[k] = f
    for f in sorted(files.values()):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k] = sum(v) / len(v)
            res["_launches"] = max(res.get("_launches", 0), len(v))
    return res


def durations(d, kernel):
    f = one(d + "/stats/**/*kernel_trace.csv")
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r) for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"])
    return [x[1] for x in rows], (rows[0][2] if rows else {})


def msm():
    K = "k_msm_accumulate<bls::FpPolicy>"
    stats_table(one("prof_msm/stats/**/*kernel_stats.csv"),
                f"{RND}: rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --no-extras\n\nMI355X, 2^20-point G1 MSM per step (GLV: 2^21 half-width terms, 8 windows of 16 bits), pipelined (4 slots); 5 warm-up + 100 timed MSMs + 5 isolated roofline/phase probes.",
                os.path.join(OUT, f"{RND}_msm_kernel_stats.md"))
    d, row = durations("prof_msm", K)
    pipe = d[5:-5] if len(d) > 20 else d
    iso = d[-5:]
    dur = sum(d) / len(d)
    try:
        b = json.loads(open(os.path.join(G, "prof_msm", "bench_under_rocprof.json")).read().strip().splitlines()[-1])
        note = f"the bench line printed by this very run: value = {b['value']:.4g} scalar-muls/s, ms_per_step = {b['ms_per_step']:.3f}, roofline.launch_ms = {b['roofline']['launch_ms']:.3f}, launch_ms_isolated = {b['roofline']['launch_ms_isolated']:.3f}, frac = {b['roofline']['frac']:.3f}"
    except Exception:
        note = "bench line of the traced run not available"
    c = counters("prof_msm", K)
    fetch_kb, write_kb = c.get("FETCH_SIZE", 0), c.get("WRITE_SIZE", 0)
    hbm = (2 * fetch_kb + write_kb) * 1024
    n, W = 1 << 20, 16
    alg = W * n * (128 + 4) + (1 << 18) * 176
    mads = W * n * (7 * 406 + 2 * 315 + 602)
    j = {"kernel": K, "workload": "2^20-point G1 MSM (GLV: 2^21 terms x 8 windows, c=16)", "launch_avg_ns": dur, "launches": len(d),
         "launch_pipelined_avg_ns": sum(pipe) / max(1, len(pipe)), "launch_isolated_avg_ns": sum(iso) / max(1, len(iso)),
         "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb, "hbm_bytes_per_launch_corrected": hbm,
         "correction": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 16-byte-per-lane reads at half); Infinity-Cache hits are included",
         "algorithmic_bytes_per_launch": alg, "counters": {k: v for k, v in c.items() if not k.startswith("_")},
         "mad_wave_insts_expected": mads / 64, "vgpr": meta_find(META, "k_msm_accumulate FpPolicy").get("vgpr_count"),
         "scratch": meta_find(META, "k_msm_accumulate FpPolicy").get("private_segment_fixed_size")}
    json.dump(j, open(os.path.join(OUT, f"{RND}_msm_pmc.json"), "w"), indent=1)
    WRITTEN.append("msm")
    ghz = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur * 1e-9) / 1e9 if dur else 0
    wc = c.get("SQ_WAVE_CYCLES", 1)
    with open(os.path.join(OUT, f"{RND}_msm_pmc.md"), "w") as fh:
        fh.write(f"""# {RND}: PMC counters of the dominant kernel ({K}, 2^20 points, GLV, c = 16)

Separate rocprofv3 passes (one `--pmc` group each, `--kernel-trace` only; tools/pmc.sh, tools/collect_profiles.sh) over `python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras`; counter averages over the {c.get('_launches', 0)} launches of a pass.

* launch duration (kernel trace of the DEFAULT bench command): {dur/1e3:.0f} us over all {len(d)} launches = {sum(pipe)/max(1,len(pipe))/1e3:.0f} us for the timed pipelined launches (what `bench.py` reports as `roofline.launch_ms`) and {sum(iso)/max(1,len(iso))/1e3:.0f} us for the 5 isolated probes at the end (`launch_ms_isolated`); {note}
* registers / scratch per lane (code-object metadata): {j['vgpr']} VGPR, {j['scratch']} B scratch
* FETCH_SIZE = {fetch_kb:.0f} KB raw -> x2 (gfx950 half-count of 16-byte-per-lane reads) = {2*fetch_kb*1024/1e9:.2f} GB;  WRITE_SIZE = {write_kb:.0f} KB = {write_kb*1024/1e9:.2f} GB
* HBM-side traffic per launch = {hbm/1e9:.2f} GB (algorithmic: 8 windows x 2^21 gathers x (128 B record + 4 B index) + 2^18 x 176 B bucket records = {alg/1e9:.2f} GB); at {dur/1e3:.0f} us that is {hbm/dur/1e3:.2f} TB/s = {100*hbm/dur/1e3/8:.0f}% of the 8 TB/s HBM peak: not memory-bound.  With GLV the gathers range over 256 MB (bases + their endomorphism images), the size of the Infinity Cache.
* SQ_INSTS_VALU = {c.get('SQ_INSTS_VALU',0):.3e} wave-instructions, of which {mads/64:.3e} are the v_mad_u64_u32 of the 7 mul + 2 sqr + one two-product sum per mixed addition ({100*mads/64/max(c.get('SQ_INSTS_VALU',1),1):.0f}%).
* wave-cycle split: SQ_ACTIVE_INST_ANY {100*c.get('SQ_ACTIVE_INST_ANY',0)/wc:.0f}%, SQ_WAIT_INST_ANY (issue stalls) {100*c.get('SQ_WAIT_INST_ANY',0)/wc:.0f}%, SQ_WAIT_ANY (s_waitcnt) {100*c.get('SQ_WAIT_ANY',0)/wc:.0f}% of SQ_WAVE_CYCLES = {wc:.3e}
* instruction cache: {c.get('SQC_ICACHE_MISSES',0):.3e} misses of {c.get('SQC_ICACHE_REQ',0):.3e} requests
* GRBM_GUI_ACTIVE = {c.get('GRBM_GUI_ACTIVE',0):.3e} (summed over 8 XCDs) -> {ghz:.2f} GHz effective clock under profiling.
""")


WIDE_NOTES = """* what the traffic is: every workgroup streams the two generated programs (bls12_381_amd/wide_prog.bin, ~4 MB) once, 256 workgroups per launch = ~1 GB of descriptor reads, of which a few per cent reach HBM (the rest hits L2 / Infinity Cache); the pairing's own inputs and outputs are 864 B per item.
* why three quarters of the wave cycles are `s_waitcnt`: twelve to fifteen of the sixteen wavefronts of a workgroup wait at the round's barrier while the one wavefront that holds the reducer lanes runs the Montgomery reduction (~55 % of a round, DESIGN.md 4.4) -- the path is built for the latency of ONE pairing, not for occupancy.
"""


def pairing(d, K, units, unit_name, mac32, tag, what, layout, alg_bytes_per_unit, meta_key=None, notes=""):
    dd, row = durations(d, K)
    if not dd:
        return
    dur = sum(dd) / len(dd)
    c = counters(d, K)
    m = meta_find(META, meta_key or K)
    stats_table(one(d + "/stats/**/*kernel_stats.csv"), f"{RND}: rocprofv3 --kernel-trace --stats -- python tools/run_pairing.py {what}", os.path.join(OUT, f"{RND}_{tag}_kernel_stats.md"))
    wc = c.get("SQ_WAVE_CYCLES", 1)
    fetch_kb, write_kb = c.get("FETCH_SIZE", 0), c.get("WRITE_SIZE", 0)
    hbm = (2 * fetch_kb + write_kb) * 1024
    alg = units * alg_bytes_per_unit
    j = {"kernel": K, "layout": layout, "units": units, "unit": unit_name, "launch_avg_ns": dur, "launches": len(dd), "mac32_per_unit": mac32,
         "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb, "hbm_bytes_per_launch_corrected": hbm, "algorithmic_bytes_per_launch": alg,
         "traffic_over_algorithmic": hbm / alg if alg else None,
         "correction": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 16-byte-per-lane reads at half); WRITE_SIZE as reported; Infinity-Cache hits are included",
         "vgpr": m.get("vgpr_count"), "scratch_frame_bytes": m.get("private_segment_fixed_size"), "lds_bytes_per_block": m.get("group_segment_fixed_size"),
         "counters": {k: v for k, v in c.items() if not k.startswith("_")}}
    json.dump(j, open(os.path.join(OUT, f"{RND}_{tag}_pmc.json"), "w"), indent=1)
    WRITTEN.append(tag)
    with open(os.path.join(OUT, f"{RND}_{tag}_pmc.md"), "w") as fh:
        fh.write(f"""# {RND}: PMC counters of {K} ({units} {unit_name} per launch, {layout})

`tools/run_pairing.py {what}` under rocprofv3 (`--kernel-trace` + one `--pmc` group per pass; tools/pmc.sh); averages over {len(dd)} launches.

* launch duration: {dur/1e6:.2f} ms -> {units/(dur*1e-9):.3e} {unit_name}/s; canonical work {mac32/1e6:.2f} M MAC32 per unit (SURVEY.md 8d) -> {units*mac32/(dur*1e-9)/1e12:.2f} TMAC32/s
* registers / scratch / LDS (code-object metadata): {m.get('vgpr_count')} VGPR, {m.get('private_segment_fixed_size')} B scratch frame per lane (the deepest call path of the final exponentiation; the hot loops touch none of it), {m.get('group_segment_fixed_size')} B LDS per block
* HBM-side traffic per launch: FETCH_SIZE = {fetch_kb/1e6:.3f} GB raw -> x2 = {2*fetch_kb*1024/1e9:.2f} GB, WRITE_SIZE = {write_kb*1024/1e9:.2f} GB, together {hbm/1e9:.2f} GB against {alg/1e6:.2f} MB of inputs and outputs ({alg_bytes_per_unit} B per unit) = {hbm/alg if alg else 0:.1f}x
* SQ_INSTS_VALU = {c.get('SQ_INSTS_VALU',0):.3e} wave-instructions; SQ_INSTS_VMEM_RD / WR = {c.get('SQ_INSTS_VMEM_RD',0):.3e} / {c.get('SQ_INSTS_VMEM_WR',0):.3e}; SQ_INSTS_LDS = {c.get('SQ_INSTS_LDS',0):.3e}
* wave-cycle split: SQ_ACTIVE_INST_ANY {100*c.get('SQ_ACTIVE_INST_ANY',0)/wc:.0f}%, SQ_WAIT_INST_ANY (issue stalls) {100*c.get('SQ_WAIT_INST_ANY',0)/wc:.0f}%, SQ_WAIT_ANY (s_waitcnt) {100*c.get('SQ_WAIT_ANY',0)/wc:.0f}% of SQ_WAVE_CYCLES = {wc:.3e}
* instruction cache: {c.get('SQC_ICACHE_MISSES',0):.3e} misses of {c.get('SQC_ICACHE_REQ',0):.3e} requests
* GRBM_GUI_ACTIVE = {c.get('GRBM_GUI_ACTIVE',0):.3e} (GRBM_GUI_ACTIVE / launch duration; the counter is summed over the chip's engines, so only its ratio between variants is meaningful)
{notes}""")


os.makedirs(OUT, exist_ok=True)
if one("prof_msm/stats/**/*kernel_stats.csv"):
    msm()
if one("prof_pair/stats/**/*kernel_trace.csv"):
    pairing("prof_pair", "k_pairing_quad", 65536, "pairings", 4.8e6, "pairing", "pairing 16 3", "quad layout: one pairing per four lanes, 2 wavefronts/SIMD", 864)
if one("prof_pair_lp/stats/**/*kernel_trace.csv"):
    pairing("prof_pair_lp", "k_pairing(", 65536, "pairings", 4.8e6, "pairing_lanepair", "pairing 16 3 (BLSGPU_PAIRING_LAYOUT=pair)", "lane-pair layout of rounds 1-2, 2 wavefronts/SIMD", 864, meta_key="9k_pairingEi")
if one("prof_mml/stats/**/*kernel_trace.csv"):
    pairing("prof_mml", "k_multi_miller_shared", 262144, "terms", 2.07e6, "mml", "mml 18 3", "lane-pair layout, four terms per shared accumulator", 288)
if one("prof_wide/stats/**/*kernel_trace.csv"):
    pairing("prof_wide", "k_pairing_wide", 256, "pairings", 4.8e6, "pairing_wide", "pairing 8 20",
            "wide layout: one pairing per 1024-lane workgroup, one workgroup per CU (the small-batch latency path)", 864, notes=WIDE_NOTES)
if one("prof_mmlp/stats/**/*kernel_trace.csv"):
    pairing("prof_mmlp", "k_mml_prep_quad", 262144, "terms", 2900 * 300, "mml_prepared", "mmlp 18 3",
            "quad layout, eight PREPARED terms per shared accumulator (lines loaded from a resident G2Prepared table of four points)", 96, meta_key="k_mml_prep_quad")
if one("prof_eqp/stats/**/*kernel_trace.csv"):
    pairing("prof_eqp", "k_mml_prep_quad", 65536, "equations", (6900 + 2 * 2900) * 300, "equations_prepared", "eqp 16 3",
            "quad layout, one accumulator per three-term equation: one unprepared term (running point in the coalesced work area) + two prepared terms", 3 * 96 + 192 + 576, meta_key="k_mml_prep_quad")
if one("prof_mmlq/stats/**/*kernel_trace.csv"):
    pairing("prof_mmlq", "k_mml_prep_quad", 262144, "terms", 2.07e6, "mml_quad_explicit", "mml 18 3 (BLSGPU_MML_IMPL=4)",
            "quad layout, four UNPREPARED terms per shared accumulator, per-term state placed by hand (coalesced work area + LDS parking)", 288, meta_key="k_mml_prep_quad")
if one("prof_eq/stats/**/*kernel_trace.csv"):
    # the whole call: Miller quads + segmented product + final exponentiations; the counters below are those of the Miller kernel, the call's
    # traffic (all kernels) is in the json as hbm_bytes_per_call
    pairing("prof_eq", "k_pairing_quad", 49152, "Miller loops (2^14 three-term equations)", 2.07e6, "equations", "equations 14 3",
            "quad layout, one Miller loop per term (per-term path of blsgpu_multi_miller_loop_many)", 288)
    try:
        import csv as _csv
        tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "SQ_INSTS_VALU": 0.0}
        names = ("k_pairing_quad", "k_multi_miller_seg", "k_fp12_prod_seg_quad", "k_final_exp_quad", "k_mml_prep_quad")
        newest = []
        for pd in sorted(glob.glob(os.path.join(G, "prof_eq", "pmc_*"))):
            fs = glob.glob(os.path.join(pd, "**", "*counter_collection.csv"), recursive=True)
            if fs:
                newest.append(max(fs, key=os.path.getmtime))          # gpurun merges every run into gpurun_out/: the newest file of each pass
        for f in newest:
            for r in _csv.DictReader(open(f)):
                if r["Counter_Name"] in tot and any(k in r["Kernel_Name"] for k in names):
                    tot[r["Counter_Name"]] += float(r["Counter_Value"])
        calls = 3                                                        # tools/run_pairing.py equations 14 3
        jp = os.path.join(OUT, f"{RND}_equations_pmc.json")
        j = json.load(open(jp)); j["hbm_bytes_per_launch_miller_kernel"] = j["hbm_bytes_per_launch_corrected"]
        j["hbm_bytes_per_launch_corrected"] = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / calls
        j["valu_wave_instructions_per_call"] = tot["SQ_INSTS_VALU"] / calls          # every kernel of the call (bench.py::valu_issue reads this first)
        j["algorithmic_bytes_per_call"] = 16384 * (3 * 288 + 576)
        j["note"] = "hbm_bytes_per_launch_corrected = every kernel of ONE blsgpu_multi_miller_loop_many call (2^14 three-term equations) together"
        json.dump(j, open(jp, "w"), indent=1)
    except Exception as ex:
        print("equations: call-level traffic not written:", ex)
for mode, what in ((0, "keys in G1, signatures in G2"), (1, "keys in G2, signatures in G1")):
    f = one("prof_verify/mode%d/**/*kernel_stats.csv" % mode)
    if f:
        stats_table(f, f"{RND}: rocprofv3 --kernel-trace --stats -- python tools/run_verify.py 14 {mode} 3  (bulk verification of 2^14 signatures from bytes, {what}; "
                       "three calls + the construction of the synthetic signatures)", os.path.join(OUT, f"{RND}_verify_chain_mode{mode}_kernel_stats.md"))
from srcdigest import stamp
stamp(RND, WRITTEN)            # the counters were collected from this checkout's sources (bench.py marks a figure stale when they differ)
print("profiles written for", RND)
