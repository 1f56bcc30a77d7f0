#!/usr/bin/env python3
"""HBM-side traffic of the secondary bench legs whose roofline block had `traffic: null`.

    python tools/leg_traffic.py collect [leg ...]      on the GPU box: per leg two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel trace only,
                                                       as /opt/skills/guides/MI355X_MICROARCH.md prescribes) of tools/run_leg.py -> gpurun_out/prof_leg_<leg>
    python tools/leg_traffic.py summarise r06          -> profiles/r06_<tag>_pmc.json (the file bench.py::static_traffic reads) + one table, profiles/r06_leg_traffic.md

A leg may launch several kernels per call (the G2 MSM launches ~40): the figure is the sum over every kernel the leg's calls launched, divided by
the number of calls; kernels that only build the inputs are excluded by name.  FETCH_SIZE / WRITE_SIZE are KiB; gfx950 counts 64-byte fetches
as 32-byte ones, so FETCH_SIZE is doubled (the same correction as tools/summarise_profiles.py).  A third pass counts SQ_INSTS_VALU: the VALU wave-instructions a call EXECUTES -- bench.py turns it into
`valu_issue_frac`, the measured counterpart of the canonical (and, for the decoding / hashing legs, estimated) MAC32 fractions.  The bulk-verification chain is covered the
same way through tools/run_verify.py."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
# leg -> (bench tag, command, algorithmic bytes per call, kernels that only prepare inputs)
N16 = 1 << 16
LEGS = {
    "h2c_g2": ("hash_to_g2", ["tools/run_leg.py", "h2c_g2", "3"], N16 * (32 + 288), ()),
    "h2c_g1": ("hash_to_g1", ["tools/run_leg.py", "h2c_g1", "3"], N16 * (32 + 144), ()),
    "decode_g1": ("decode_g1", ["tools/run_leg.py", "decode_g1", "3"], N16 * (48 + 96 + 2), ("k_fixed_base", "k_bases_from_scalars", "k_bases_export", "k_bases_endo", "k_point_encode", "k_bases_subgroup")),
    "decode_g2": ("decode_g2", ["tools/run_leg.py", "decode_g2", "3"], N16 * (96 + 192 + 2), ("k_fixed_base", "k_bases_from_scalars", "k_bases_export", "k_bases_endo", "k_point_encode", "k_bases_subgroup")),
    "g2_msm": ("g2_msm", ["tools/run_leg.py", "g2_msm", "3"], (1 << 20) * (192 + 32) + 288, ("k_fixed_base", "k_bases_from_scalars", "k_bases_endo", "k_bases_subgroup")),
    "mul_g1": ("g1_mul_batch", ["tools/run_leg.py", "mul_g1", "3"], (1 << 20) * (96 + 32 + 144), ("k_fixed_base", "k_bases_from_scalars", "k_bases_export", "k_bases_endo", "k_bases_subgroup")),
    "mul_g2": ("g2_mul_batch", ["tools/run_leg.py", "mul_g2", "3"], (1 << 18) * (192 + 32 + 288), ("k_fixed_base", "k_bases_from_scalars", "k_bases_export", "k_bases_endo", "k_bases_subgroup")),
    "ntt": ("ntt_leg", ["tools/run_leg.py", "ntt", "3"], (1 << 20) * 64, ()),
    "verify": ("bls_verify", ["tools/run_verify.py", "14", "0", "3"], (1 << 14) * (48 + 96 + 32 + 1),
               ("k_fixed_base", "k_bases_from_scalars", "k_bases_export", "k_bases_endo", "k_bases_subgroup", "k_mul_batch", "k_point_encode")),
}
CALLS = {"verify": 3}          # run_verify makes exactly `reps` calls (its set-up uses other entry points, excluded by name where they overlap)


def collect(legs):
    for leg in legs:
        tag, cmd, _, _ = LEGS[leg]
        out = os.path.join(G, "prof_leg_" + leg)
        subprocess.run(["rm", "-rf", out]); os.makedirs(out, exist_ok=True)
        for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU")):
            log = open(os.path.join(out, "pmc_%d.log" % i), "w")
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", os.path.join(out, "pmc_%d" % i), "--", sys.executable] + cmd,
                           cwd=ROOT, stdout=log, stderr=subprocess.STDOUT, env=dict(os.environ, TMPDIR="/tmp"))
        for f in glob.glob(os.path.join(out, "**", "*agent_info.csv"), recursive=True):
            os.remove(f)
        print("collected", leg, flush=True)


def summarise(rnd):
    rows = []
    for leg, (tag, cmd, alg, setup) in LEGS.items():
        d = os.path.join(G, "prof_leg_" + leg)
        if not os.path.isdir(d):
            continue
        calls = CALLS.get(leg)
        if calls is None:
            for line in open(os.path.join(d, "pmc_0.log"), errors="replace"):
                if line.startswith("LEG_KERNEL_WINDOW"):
                    calls = int(line.split()[1])
        if not calls:
            print("no call count for", leg); continue
        tot = collections.Counter(); per_kernel = collections.defaultdict(collections.Counter)
        # gpurun MERGES every run's files into gpurun_out/ (the file names carry the process id): per pass directory only the newest file counts
        newest = []
        for pd in sorted(glob.glob(os.path.join(d, "pmc_*"))):
            fs = glob.glob(os.path.join(pd, "**", "*counter_collection.csv"), recursive=True)
            if fs:
                newest.append(max(fs, key=os.path.getmtime))
        for f in newest:
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if any(s in name for s in setup) or name.startswith("__amd_rocclr") or "at::native" in name or "k_fr_tw" in name:
                    continue
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                per_kernel[name.split("(")[0][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
        fetch, write = tot["FETCH_SIZE"] * 1024 / calls, tot["WRITE_SIZE"] * 1024 / calls
        hbm = 2 * fetch + write
        j = {"leg": leg, "calls_in_run": calls, "FETCH_SIZE_bytes_raw_per_call": fetch, "WRITE_SIZE_bytes_per_call": write, "hbm_bytes_per_launch_corrected": hbm,
             "valu_wave_instructions_per_call": (tot["SQ_INSTS_VALU"] / calls) if tot["SQ_INSTS_VALU"] else None,
             "algorithmic_bytes_per_launch": alg, "ratio": hbm / alg,
             "correction": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64-byte fetches as 32-byte ones); WRITE_SIZE as reported; sum over every kernel of a call",
             "kernels": {k: {"fetch_bytes_raw_per_call": v["FETCH_SIZE"] * 1024 / calls, "write_bytes_per_call": v["WRITE_SIZE"] * 1024 / calls} for k, v in per_kernel.items()}}
        with open(os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (rnd, tag)), "w") as fh:
            json.dump(j, fh, indent=1)
        rows.append((leg, tag, calls, fetch, write, hbm, alg))
    with open(os.path.join(ROOT, "profiles", "%s_leg_traffic.md" % rnd), "w") as fh:
        fh.write("# %s: HBM-side traffic of the secondary bench legs (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; tools/leg_traffic.py)\n\n" % rnd)
        fh.write("Per call of the leg, summed over every kernel the call launches; FETCH_SIZE doubled (gfx950 correction), Infinity-Cache hits included.\n\n")
        fh.write("| leg (bench key) | calls in run | fetch raw | write | fetch x 2 + write | algorithmic | ratio |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for leg, tag, calls, fetch, write, hbm, alg in rows:
            fh.write("| %s (`%s`) | %d | %.3e | %.3e | %.3e | %.3e | %.1fx |\n" % (leg, tag, calls, fetch, write, hbm, alg, hbm / alg))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from srcdigest import stamp
    stamp(rnd, [tag for _, tag, *_ in rows])
    print("wrote", len(rows), "legs")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "collect":
        collect(sys.argv[2:] or list(LEGS))
    else:
        summarise(sys.argv[2] if len(sys.argv) > 2 else "r06")
