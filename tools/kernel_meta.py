#!/usr/bin/env python3
"""Register / scratch / LDS figures of the kernels in libblsgpu.so, read from the code object's own metadata
(llvm-readelf --notes of the gfx950 bundle) -- what the hardware allocates, unlike rocprofv3's VGPR_Count column, which
reports the arch-VGPR half on gfx950.

    python tools/kernel_meta.py [name-substring ...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_meta(lib=None):
    lib = lib or os.path.join(ROOT, "bls12_381_amd", "libblsgpu.so")
    tmp = tempfile.mkdtemp()
    try:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        objs = [os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f]
        if not objs:
            return {}
        # one code object per translation unit of the library (bls12_381_amd/csrc/host.h)
        notes = "\n".join(subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", o], capture_output=True, text=True).stdout for o in sorted(objs))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and line.lstrip().startswith("-"):
            cur = {"agpr_count": int(v)}
        elif cur is not None:
            if k == "name":
                out[v] = cur
                cur["name"] = v
            elif k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
                cur[k] = int(v)
    return out


def find(meta, needle):
    """metadata of the first kernel whose (mangled) name contains every '<'-free piece of `needle`"""
    pieces = [p for p in re.split(r"[<>:, ]+", needle) if p and p != "bls"]
    for name, m in meta.items():
        if all(p in name for p in pieces):
            return m
    return {}


if __name__ == "__main__":
    meta = kernel_meta()
    for name, m in sorted(meta.items()):
        if len(sys.argv) > 1 and not any(a in name for a in sys.argv[1:]):
            continue
        print(f"{name[:90]:90s} vgpr {m.get('vgpr_count', 0):4d}  scratch {m.get('private_segment_fixed_size', 0):6d} B  lds {m.get('group_segment_fixed_size', 0):6d} B  "
              f"vgpr spills {m.get('vgpr_spill_count', 0)}")
