"""Digest of the kernel sources behind a bench leg, stored next to its committed counter traffic (profiles/<round>_<tag>_pmc.json: "source_digest") and
compared by bench.py::static_traffic: a `traffic` figure measured on other kernel sources than the ones being timed is marked `traffic_stale`."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bls12_381_amd", "csrc")
_COMMON = ["fe.hip.h", "fp2.hip.h", "curve.hip.h", "convert.hip.h", "scalar.hip.h", "pairlane.hip.h"]
_FAMILY = {
    "msm": ["msm.hip.h", "team.hip.h", "abi_kernels.hip.h", "api_msm.hip"],
    "mul": ["mulbatch.hip.h", "msm.hip.h", "api_msm.hip"],
    "pairing": ["pairing.hip.h", "quad.hip.h", "prep.hip.h", "api_pairing.hip"],
    "aux": ["codec.hip.h", "h2c.hip.h", "expand.hip.h", "fr.hip.h", "api_aux.hip"],
}
_TAG = {"msm": "msm", "g2_msm": "msm", "g1_mul_batch": "mul", "g2_mul_batch": "mul", "pairing": "pairing", "mml": "pairing", "equations": "pairing",
        "equations_prepared": "pairing", "mml_prepared": "pairing", "pairing_wide": "pairing", "ntt": "aux", "hash_to_g1": "aux", "hash_to_g2": "aux",
        "decode_g1": "aux", "decode_g2": "aux", "ntt_leg": "aux", "bls_verify": None}          # the verification chain spans the pairing and the aux families


def source_digest(tag):
    fam = _TAG.get(tag, "pairing")
    files = _COMMON + (_FAMILY["pairing"] + _FAMILY["aux"] if fam is None else _FAMILY[fam])
    h = hashlib.sha256()
    for f in sorted(set(files)):
        p = os.path.join(CSRC, f)
        h.update(f.encode() + b"\0" + (open(p, "rb").read() if os.path.exists(p) else b"") + b"\0")
    return h.hexdigest()[:16]


def stamp(rnd, tags=None):
    """write "source_digest" into profiles/<rnd>_<tag>_pmc.json for the tags the CALLER has just written (a summarising tool stamps only the files whose
    counters it took from this checkout's run; stamping every file of the round would vouch for counters that were not re-collected)"""
    import glob
    import json
    alias = {"pairing_lanepair": "pairing", "mml_quad_explicit": "mml"}
    for f in glob.glob(os.path.join(ROOT, "profiles", "%s_*_pmc.json" % rnd)):
        tag = os.path.basename(f)[len(rnd) + 1:-len("_pmc.json")]
        if tags is not None and tag not in tags:
            continue
        j = json.load(open(f))
        j["source_digest"] = source_digest(alias.get(tag, tag))
        json.dump(j, open(f, "w"), indent=1)
