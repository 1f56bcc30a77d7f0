#!/usr/bin/env python3
"""Per-kernel / per-function ISA statistics from a device-only assembly dump (hipcc --offload-device-only -S).

    python tools/isa_stats.py build/api.s [name-filter ...]

For every symbol: VGPRs, scratch bytes, LDS bytes (kernels), and the instruction mix of its body
(v_mad_u64_u32, other VALU, scratch loads/stores, global loads/stores, DPP moves, s_waitcnt)."""
import re
import sys
from collections import Counter


def parse(path):
    funcs = {}
    cur = None
    meta = {}
    with open(path) as f:
        for line in f:
            s = line.strip()
            m = re.match(r"^([A-Za-z_][\w.$]*):\s*(;.*)?$", s)
            if m and not s.startswith(".L") and not m.group(1).startswith("BB"):
                cur = m.group(1)
                funcs[cur] = Counter()
                continue
            if s.startswith(".amdhsa_kernel "):
                meta_name = s.split()[1]
                meta[meta_name] = {}
                cur_meta = meta[meta_name]
                continue
            if s.startswith(".amdhsa_") and meta:
                parts = s.split()
                if len(parts) == 2:
                    cur_meta[parts[0][8:]] = parts[1]
                continue
            if s.startswith("; ") and cur:
                m2 = re.match(r"; (NumVgprs|ScratchSize|NumSgprs|Occupancy|LDSByteSize|codeLenInByte): (\d+)", s)
                if m2:
                    funcs[cur]["_" + m2.group(1)] = int(m2.group(2))
                continue
            if cur is None or not s or s.startswith(".") or s.startswith(";"):
                continue
            op = s.split()[0]
            c = funcs[cur]
            c["insts"] += 1
            if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
                c["mad64"] += 1
            elif op.startswith("v_"):
                c["valu_other"] += 1
                if "dpp" in s or "quad_perm" in s or "row_" in s:
                    c["dpp"] += 1
            elif op.startswith("scratch_load"):
                c["scr_ld"] += 1
            elif op.startswith("scratch_store"):
                c["scr_st"] += 1
            elif op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
                c["g_ld"] += 1
            elif op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"):
                c["g_st"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith("s_swappc") or op.startswith("s_setpc"):
                c["calls"] += 1
    return funcs, meta


def main():
    path = sys.argv[1]
    filt = sys.argv[2:]
    funcs, meta = parse(path)
    print(f"{'symbol':70s} {'vgpr':>5s} {'scr B':>6s} {'insts':>7s} {'mad64':>7s} {'valu':>7s} {'dpp':>5s} {'scrld':>6s} {'scrst':>6s} {'gld':>5s} {'gst':>5s} {'lds':>5s} {'call':>5s}")
    for name, c in funcs.items():
        if c["insts"] < 20:
            continue
        if filt and not any(x in name for x in filt):
            continue
        print(f"{name[:70]:70s} {c.get('_NumVgprs', 0):5d} {c.get('_ScratchSize', 0):6d} {c['insts']:7d} {c['mad64']:7d} {c['valu_other']:7d} {c['dpp']:5d} "
              f"{c['scr_ld']:6d} {c['scr_st']:6d} {c['g_ld']:5d} {c['g_st']:5d} {c['lds']:5d} {c['calls']:5d}")


if __name__ == "__main__":
    main()
