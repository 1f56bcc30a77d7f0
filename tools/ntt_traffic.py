#!/usr/bin/env python3
"""HBM-side traffic of ONE 2^20-point Fr transform from rocprofv3 counter passes (run on the GPU box):

    python tools/ntt_traffic.py collect      # two --pmc passes (FETCH_SIZE, WRITE_SIZE) + a kernel trace of tools/ntt_time.py 20 -> gpurun_out/prof_ntt
    python tools/ntt_traffic.py summarise r04   # -> profiles/r04_ntt_pmc.json / .md

A transform is several launches (column-tile passes since round 5, radix-4 passes over global memory before, + the LDS tile kernel); the counters are summed over the launches
of one transform: total over the run / number of transforms in it (twiddle-table kernels excluded).  FETCH_SIZE / WRITE_SIZE are in
KiB... on gfx950 FETCH_SIZE counts 64-byte units as 32-byte ones (MI355X_MICROARCH.md): the summary applies the same correction
as tools/summarise_profiles.py (fetch bytes doubled)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "gpurun_out", "prof_ntt")
KERNELS = ("k_fr_stage2", "k_fr_stage1", "k_fr_cols", "k_fr_tile")


def collect():
    env = dict(os.environ, TMPDIR="/tmp", NTT_TIME_ONLY_DEFAULT="1")
    subprocess.run(["rm", "-rf", D]); os.makedirs(D)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "ntt_time.py"), "20"]
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(D, "stats"), "--"] + cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", os.path.join(D, "pmc_%d" % i), "--"] + cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in glob.glob(os.path.join(D, "**", "*agent_info.csv"), recursive=True):
        os.remove(f)


def summarise(rnd):
    tot = collections.defaultdict(float)
    tiles = 0
    for f in glob.glob(os.path.join(D, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in KERNELS):
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                if "k_fr_tile" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                    tiles += 1
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(D, "stats", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for k in KERNELS:
                if k in r["Kernel_Name"]:
                    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    n = 1 << 20
    fetch = tot["FETCH_SIZE"] * 1024 / max(tiles, 1)
    write = tot["WRITE_SIZE"] * 1024 / max(tiles, 1)
    res = {"workload": "one in-place 2^20-point Fr transform (tools/ntt_time.py 20)", "transforms_profiled": tiles,
           "fetch_bytes_per_transform_raw": fetch, "write_bytes_per_transform": write,
           "hbm_bytes_per_launch_corrected": 2 * fetch + write, "algorithmic_bytes": n * 64,
           "ratio_to_algorithmic": (2 * fetch + write) / (n * 64),
           "kernel_avg_us": {k: sum(v) / len(v) for k, v in dur.items() if v}, "launches_per_transform": {k: len(v) / max(tiles, 1) for k, v in dur.items() if v}}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "profiles", "%s_ntt_pmc.json" % rnd), "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", "%s_ntt_pmc.md" % rnd), "w") as fh:
        fh.write("# %s: HBM-side traffic of one 2^20-point Fr transform (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)\n\n" % rnd)
        fh.write("| | bytes per transform |\n|---|---:|\n")
        fh.write("| FETCH_SIZE (raw, KiB units -> bytes) | %.3e |\n| WRITE_SIZE | %.3e |\n| fetch x 2 (gfx950 correction) + write | %.3e |\n| algorithmic (32 B in + 32 B out per element) | %.3e |\n| ratio | %.2f |\n\n"
                 % (fetch, write, 2 * fetch + write, n * 64, res["ratio_to_algorithmic"]))
        fh.write("Kernels of one transform (kernel trace, average per launch): " + ", ".join("%s %.1f us x %.1f" % (k, res["kernel_avg_us"][k], res["launches_per_transform"][k]) for k in res["kernel_avg_us"]) + "\n")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from srcdigest import stamp
    stamp(rnd, ["ntt"])
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "collect":
        collect()
    else:
        summarise(sys.argv[2] if len(sys.argv) > 2 else "r04")
