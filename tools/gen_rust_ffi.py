#!/usr/bin/env python3
"""include/bls12_381_hip.h -> rust/bls12_381-hip/src/ffi.rs (the `extern "C"` block of the Rust binding).

    python tools/gen_rust_ffi.py            # prints the Rust source
    python tools/gen_rust_ffi.py --write    # rewrites rust/bls12_381-hip/src/ffi.rs

tests/test_host_cpu.py::test_rust_ffi_matches_header re-parses BOTH files independently and compares every signature, so a
hand edit of either side that drifts from the other fails the CPU suite."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bls12_381_hip.h")
OUT = os.path.join(ROOT, "rust", "bls12_381-hip", "src", "ffi.rs")

C_TO_RUST = {
    "int": "c_int", "size_t": "usize", "void": "()", "unsigned": "c_uint",
    "blsgpu_ctx*": "*mut BlsgpuCtx", "blsgpu_ctx**": "*mut *mut BlsgpuCtx",
    "blsgpu_bases*": "*mut BlsgpuBases", "const blsgpu_bases*": "*const BlsgpuBases", "blsgpu_bases**": "*mut *mut BlsgpuBases",
    "const uint64_t*": "*const u64", "uint64_t*": "*mut u64", "const uint8_t*": "*const u8", "uint8_t*": "*mut u8",
    "const void*": "*const c_void", "void*": "*mut c_void", "double*": "*mut f64", "float*": "*mut f32", "unsigned*": "*mut c_uint",
    "const char*": "*const c_char", "char*": "*mut c_char", "size_t*": "*mut usize", "const int*": "*const c_int",
    "blsgpu_group*": "*mut BlsgpuGroup", "const blsgpu_group*": "*const BlsgpuGroup", "blsgpu_group**": "*mut *mut BlsgpuGroup",
    "blsgpu_g2_prepared*": "*mut BlsgpuG2Prepared", "const blsgpu_g2_prepared*": "*const BlsgpuG2Prepared", "blsgpu_g2_prepared**": "*mut *mut BlsgpuG2Prepared",
    "const uint32_t*": "*const u32", "uint32_t*": "*mut u32", "const void*const*": "*const *const c_void", "void*const*": "*const *mut c_void", "const size_t*": "*const usize",
    "blsgpu_group_g2_prepared*": "*mut BlsgpuGroupG2Prepared", "const blsgpu_group_g2_prepared*": "*const BlsgpuGroupG2Prepared", "blsgpu_group_g2_prepared**": "*mut *mut BlsgpuGroupG2Prepared",
    "blsgpu_group_bases*": "*mut BlsgpuGroupBases", "const blsgpu_group_bases*": "*const BlsgpuGroupBases", "blsgpu_group_bases**": "*mut *mut BlsgpuGroupBases",
}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def parse_header(path=HEADER):
    """-> list of (name, return C type, [(C type, param name)])"""
    text = strip_comments(open(path).read())
    out = []
    for m in re.finditer(r"\b((?:const\s+)?[A-Za-z_][\w]*\s*\**)\s*\b(blsgpu_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        ret = re.sub(r"\s+", " ", ret).replace(" *", "*").strip()
        ps = []
        params = params.strip()
        if params and params != "void":
            for p in params.split(","):
                p = re.sub(r"\s+", " ", p).strip()
                arr = re.match(r"^(.*?)(\w+)\s*\[\s*\d+\s*\]$", p)
                if arr:                                   # `uint64_t out[18]` decays to a pointer
                    ctype, pname = arr.group(1).strip() + "*", arr.group(2)
                else:
                    mm = re.match(r"^(.*?)(\w+)$", p)
                    ctype, pname = mm.group(1).strip(), mm.group(2)
                ctype = ctype.replace(" *", "*").replace("* ", "*")
                ps.append((ctype, pname))
        out.append((name, ret, ps))
    return out


def rust_source():
    decls = parse_header()
    lines = ["//! `extern \"C\"` declarations of libblsgpu.so -- GENERATED from include/bls12_381_hip.h by tools/gen_rust_ffi.py; do not edit.",
             "//! One entry per symbol of the C ABI; the header documents each one and cites the reference lines it replaces.",
             "#![allow(non_camel_case_types, dead_code)]",
             "use core::ffi::{c_char, c_int, c_uint, c_void};",
             "",
             "/// opaque: one device + its streams and scratch (blsgpu_create / blsgpu_destroy)",
             "#[repr(C)] pub struct BlsgpuCtx { _private: [u8; 0] }",
             "/// opaque: bases resident in HBM (blsgpu_g1_bases_upload & co. / blsgpu_bases_free)",
             "#[repr(C)] pub struct BlsgpuBases { _private: [u8; 0] }",
             "/// opaque: one context per listed device (blsgpu_group_create / blsgpu_group_destroy)",
             "#[repr(C)] pub struct BlsgpuGroup { _private: [u8; 0] }",
             "/// opaque: bases sharded over the members of a group (blsgpu_group_bases_upload / blsgpu_group_bases_free)",
             "#[repr(C)] pub struct BlsgpuGroupBases { _private: [u8; 0] }",
             "/// opaque: `G2Prepared` values resident in HBM (blsgpu_g2_prepare / blsgpu_g2_prepared_free)",
             "#[repr(C)] pub struct BlsgpuG2Prepared { _private: [u8; 0] }",
             "/// opaque: the same `G2Prepared` table on every member of a group (blsgpu_group_g2_prepare / blsgpu_group_g2_prepared_free)",
             "#[repr(C)] pub struct BlsgpuGroupG2Prepared { _private: [u8; 0] }",
             "",
             "pub const BLSGPU_OK: c_int = 0;",
             "/// scalar arguments hold `Scalar::to_bytes()` output (default) / the Montgomery limbs of `Scalar([u64; 4])` (blsgpu_set_scalar_form)",
             "pub const BLSGPU_SCALAR_BYTES: c_int = 0;",
             "pub const BLSGPU_SCALAR_MONT: c_int = 1;",
             "",
             "#[link(name = \"blsgpu\")]",
             "extern \"C\" {"]
    for name, ret, ps in decls:
        args = ", ".join("%s: %s" % ("r#type" if n == "type" else n, C_TO_RUST[t]) for t, n in ps)
        r = "" if ret == "void" else " -> " + C_TO_RUST[ret]
        lines.append("    pub fn %s(%s)%s;" % (name, args, r))
    lines.append("}")
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    src = rust_source()
    if "--write" in sys.argv:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        open(OUT, "w").write(src)
        print("wrote", OUT)
    else:
        sys.stdout.write(src)
