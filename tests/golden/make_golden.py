#!/usr/bin/env python3
"""Extract the reference's own known-answer material into committed fixtures.

Run in the build container (needs /root/reference, which does not exist on the GPU box):

    python3 tests/golden/make_golden.py

Outputs (all under tests/golden/):
  ref_kats.json     every 6-limb (Fp) / 4-limb (Scalar) hex literal of the reference's #[test] functions,
                    grouped per test in source order, plus the named module constants and the RELIC
                    pairing constant (Gt::generator) -- /root/reference/src/{fp,fp2,fp6,fp12,g1,g2,scalar,
                    pairings}.rs.  Only numbers are extracted; no reference code is copied.
  h2c_vectors.json  the RFC 9380 (draft-16) vectors the reference's integration tests hold for hash_to_curve / encode_to_curve
                    (XMD:SHA-256, G1 and G2) and expand_message_xmd(SHA-256): /root/reference/tests/{hash_to_curve_g1,
                    hash_to_curve_g2,expand_msg}.rs -- (dst, msg, expected output) as hex.
  g1_uncompressed_valid_test_vectors.dat, g1_compressed_..., g2_uncompressed_..., g2_compressed_...
                    byte copies of /root/reference/src/tests/*.dat (k*generator, k = 0..999;
                    src/tests/mod.rs:3-76) -- binary golden vectors, not source.
"""
import json, os, re, shutil

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))
HEX6 = re.compile(r"\[\s*((?:0x[0-9a-f_]+\s*,\s*){5}0x[0-9a-f_]+)\s*,?\s*\]")
HEX4 = re.compile(r"\[\s*((?:0x[0-9a-f_]+\s*,\s*){3}0x[0-9a-f_]+)\s*,?\s*\]")


def limbs(m):
    return [int(x.strip().replace("_", ""), 16) for x in m.split(",") if x.strip()]


def arrays(text, rx):
    return [limbs(m) for m in rx.findall(text)]


def lines(path, a, b):
    with open(path) as fh:
        return "".join(fh.readlines()[a - 1:b])


kats = {"tests": {}, "consts": {}}
for mod in ["fp", "fp2", "fp6", "fp12", "g1", "g2", "scalar", "pairings"]:
    src = open(f"{REF}/{mod}.rs").read()
    for part in src.split("#[test]")[1:]:
        name = re.search(r"fn (\w+)\(", part).group(1)
        a6, a4 = arrays(part, HEX6), arrays(part, HEX4)
        if a6 or a4:
            kats["tests"][f"{mod}.{name}"] = {"fp": a6, "scalar": a4}

C = kats["consts"]
C["fp.MODULUS"] = arrays(lines(f"{REF}/fp.rs", 70, 77), HEX6)[0]
C["fp.INV"] = int(re.search(r"const INV: u64 = (0x[0-9a-f_]+);", open(f"{REF}/fp.rs").read()).group(1).replace("_", ""), 16)
C["fp.R"] = arrays(lines(f"{REF}/fp.rs", 83, 90), HEX6)[0]
C["fp.R2"] = arrays(lines(f"{REF}/fp.rs", 93, 100), HEX6)[0]
C["fp.R3"] = arrays(lines(f"{REF}/fp.rs", 103, 110), HEX6)[0]
C["g1.B"] = arrays(lines(f"{REF}/g1.rs", 176, 183), HEX6)[0]
C["g1.GENERATOR"] = arrays(lines(f"{REF}/g1.rs", 197, 217), HEX6)
C["g1.BETA"] = arrays(lines(f"{REF}/g1.rs", 421, 428), HEX6)[0]
C["g2.B"] = arrays(lines(f"{REF}/g2.rs", 177, 194), HEX6)
C["g2.GENERATOR"] = arrays(lines(f"{REF}/g2.rs", 210, 250), HEX6)
C["fp6.FROBENIUS_C1"] = arrays(lines(f"{REF}/fp6.rs", 159, 171), HEX6)
C["fp6.FROBENIUS_C2"] = arrays(lines(f"{REF}/fp6.rs", 173, 186), HEX6)
C["fp12.FROBENIUS_C1"] = arrays(lines(f"{REF}/fp12.rs", 149, 168), HEX6)
C["scalar.MODULUS"] = arrays(lines(f"{REF}/scalar.rs", 76, 81), HEX4)[0]
C["scalar.GENERATOR"] = arrays(lines(f"{REF}/scalar.rs", 99, 105), HEX4)[0]
C["scalar.INV"] = int(re.search(r"const INV: u64 = (0x[0-9a-f_]+);", open(f"{REF}/scalar.rs").read()).group(1).replace("_", ""), 16)
C["scalar.R"] = arrays(lines(f"{REF}/scalar.rs", 159, 164), HEX4)[0]
C["scalar.R2"] = arrays(lines(f"{REF}/scalar.rs", 167, 172), HEX4)[0]
C["scalar.R3"] = arrays(lines(f"{REF}/scalar.rs", 175, 180), HEX4)[0]
C["scalar.TWO_INV"] = arrays(lines(f"{REF}/scalar.rs", 183, 188), HEX4)[0]
C["scalar.S"] = int(re.search(r"const S: u32 = (\d+);", open(f"{REF}/scalar.rs").read()).group(1))
C["scalar.ROOT_OF_UNITY"] = arrays(lines(f"{REF}/scalar.rs", 200, 205), HEX4)[0]
C["scalar.ROOT_OF_UNITY_INV"] = arrays(lines(f"{REF}/scalar.rs", 208, 213), HEX4)[0]
C["scalar.DELTA"] = arrays(lines(f"{REF}/scalar.rs", 217, 222), HEX4)[0]
C["scalar.LARGEST"] = arrays(lines(f"{REF}/scalar.rs", 1051, 1056), HEX4)[0]
C["scalar.FROM_BYTES_WIDE_MAXIMUM"] = arrays(lines(f"{REF}/scalar.rs", 1029, 1040), HEX4)[0]      # from_bytes_wide(&[0xff; 64])
gt = arrays(lines(f"{REF}/pairings.rs", 359, 475), HEX6)
assert len(gt) == 12, len(gt)
C["pairings.GT_GENERATOR"] = gt
relic = arrays(lines(f"{REF}/tests/mod.rs", 78, 231), HEX6)
assert len(relic) == 12 and relic == gt, "RELIC constant in src/tests/mod.rs must equal Gt::generator"
C["lib.BLS_X"] = int(re.search(r"const BLS_X: u64 = (0x[0-9a-f_]+);", open(f"{REF}/lib.rs").read()).group(1).replace("_", ""), 16)

# ---- hash-to-curve: map constants and the RFC 9380 (draft-16) vectors of the reference's tests ----
H2C = f"{REF}/hash_to_curve"
g1src, g2src = open(f"{H2C}/map_g1.rs").read(), open(f"{H2C}/map_g2.rs").read()


def const_block(src, name):
    """the hex literals between `const NAME` and the next top-level `const`/`impl`/`fn`"""
    m = re.search(r"^(?:pub )?const " + name + r"\b.*?(?=^(?:pub )?const |^impl|^fn |^#\[)", src, re.S | re.M)
    assert m, name
    out = []
    for t in re.finditer(r"Fp::zero\(\)|Fp::one\(\)|Fp2::zero\(\)|Fp2::one\(\)|" + HEX6.pattern, m.group(0)):
        tok = t.group(0)
        if tok.startswith("Fp::zero"): out.append([0] * 6)
        elif tok.startswith("Fp::one"): out.append(list(C["fp.R"]))
        elif tok.startswith("Fp2::zero"): out += [[0] * 6, [0] * 6]
        elif tok.startswith("Fp2::one"): out += [list(C["fp.R"]), [0] * 6]
        else: out.append(limbs(t.group(1)))
    return out


for name, cnt in [("ISO11_XNUM", 12), ("ISO11_XDEN", 11), ("ISO11_YNUM", 16), ("ISO11_YDEN", 16), ("SSWU_ELLP_A", 1), ("SSWU_ELLP_B", 1),
                  ("SSWU_XI", 1), ("SQRT_M_XI_CUBED", 1)]:
    v = const_block(g1src, name)
    assert len(v) == cnt, (name, len(v))
    C["h2c_g1." + name] = v
C["h2c_g1.F_2_256"] = arrays(lines(f"{H2C}/map_g1.rs", 513, 522), HEX6)[0]
for name, cnt in [("ISO3_XNUM", 8), ("ISO3_XDEN", 6), ("ISO3_YNUM", 8), ("ISO3_YDEN", 8), ("SSWU_ELLP_A", 2), ("SSWU_ELLP_B", 2), ("SSWU_XI", 2),
                  ("SSWU_ETAS", 8), ("SSWU_RV1", 2)]:
    v = const_block(g2src, name)
    assert len(v) == cnt, (name, len(v))                      # Fp2 constants: c0, c1 alternating
    C["h2c_g2." + name] = v
kats["tests"]["h2c_g1.test_simple_swu_expected"] = {"fp": arrays(lines(f"{H2C}/map_g1.rs", 655, 760), HEX6), "scalar": []}


def rust_bytes(lit):
    """body of a Rust byte-string literal b"..." (with backslash-newline continuations) -> bytes"""
    lit = re.sub(r"\\\n\s*", "", lit)
    return lit.encode("latin-1").decode("unicode_escape").encode("latin-1")


def vector_file(path, out_key):
    src = open(path).read()
    vecs = []
    for fn in re.split(r"#\[test\]", src)[1:]:
        name = re.search(r"fn (\w+)\(", fn).group(1)
        d = re.search(r'let dst = b"((?:[^"\\]|\\.|\\\n)*)";', fn, re.S)
        if not d:
            continue
        dst = rust_bytes(d.group(1))
        for m in re.finditer(r'msg: b"((?:[^"\\]|\\.|\\\n)*)",(.*?)(?:expected|uniform_bytes): &hex!\(\s*"([^"]*)"', fn, re.S):
            ln = re.search(r"len_in_bytes: (0x[0-9a-fA-F]+|\d+)", m.group(2))
            vecs.append({"test": name, "dst": dst.hex(), "msg": rust_bytes(m.group(1)).hex(), "out": re.sub(r"\s", "", m.group(3)),
                         "len_in_bytes": int(ln.group(1), 0) if ln else None})
    return vecs


T = "/root/reference/tests"
h2c = {"g1": vector_file(f"{T}/hash_to_curve_g1.rs", "g1"), "g2": vector_file(f"{T}/hash_to_curve_g2.rs", "g2"),
       # every expander the reference tests: XMD over SHA-256 (+ long DST) and SHA-512, XOF over SHAKE128 (+ long DST) and SHAKE256
       "expand_msg": vector_file(f"{T}/expand_msg.rs", "x")}
# `HashToField for Scalar` (hash_to_curve/map_scalar.rs:24-45): 48 input bytes -> the Debug form of the Scalar (big-endian hex of the canonical integer)
ms = open(f"{H2C}/map_scalar.rs").read()
h2c["hash_to_scalar"] = [{"okm": (bytes(48) if m.group(1) else rust_bytes(m.group(2))).hex(), "out": m.group(3)}
                         for m in re.finditer(r'(?:(&\[0u8; 48\])|b"([^"]*)"),\s*"0x([0-9a-f]{64})"', ms)]
assert len(h2c["hash_to_scalar"]) == 3, h2c["hash_to_scalar"]
assert len(h2c["g1"]) == 10 and len(h2c["g2"]) == 10 and len(h2c["expand_msg"]) == 60, {k: len(v) for k, v in h2c.items()}
with open(os.path.join(OUT, "h2c_vectors.json"), "w") as fh:
    json.dump(h2c, fh, indent=0, separators=(",", ":"))

with open(os.path.join(OUT, "ref_kats.json"), "w") as fh:
    json.dump(kats, fh, indent=0, separators=(",", ":"))
for g in ["g1_uncompressed", "g1_compressed", "g2_uncompressed", "g2_compressed"]:
    shutil.copyfile(f"{REF}/tests/{g}_valid_test_vectors.dat", os.path.join(OUT, f"{g}_valid_test_vectors.dat"))
print("tests:", len(kats["tests"]), "consts:", len(C))
