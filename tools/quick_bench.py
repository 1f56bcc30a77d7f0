import time, numpy as np, sys
sys.path.insert(0,'.')
import bls12_381_amd as b
ctx = b.default_context()
print("fp_mul/s", ctx.fp_mul_throughput(2000), "mad/s", ctx.mad_throughput(2000))
for logn in (16, 18, 20):
    n = 1 << logn
    rs = np.random.RandomState(logn)
    kb = rs.randint(0, 256, size=(n, 32), dtype=np.uint8); kb[:, 31] &= 0x3F
    sb = rs.randint(0, 256, size=(n, 32), dtype=np.uint8); sb[:, 31] &= 0x3F
    t0=time.time(); bases = ctx.bases_from_scalars(1, kb); t1=time.time()
    print(f"n=2^{logn} bases_from_scalars {t1-t0:.3f}s")
    ctx.set_profiling(True)
    for c in ([0] if logn<20 else [0, 13, 14, 15, 16]):
        ctx.set_msm_window(c)
        out = ctx.msm(bases, sb)
        t0=time.time(); out = ctx.msm(bases, sb); t1=time.time()
        print(f"  c={c} msm wall {1e3*(t1-t0):.2f} ms ->", {k: round(v,3) for k,v in ctx.last_msm_phase_ms().items()})
    ctx.set_profiling(False)
n=4096
r = np.random.RandomState(1)
a = r.randint(0,256,size=(n,32),dtype=np.uint8); a[:,31]&=0x3f
bb = r.randint(0,256,size=(n,32),dtype=np.uint8); bb[:,31]&=0x3f
g1,f1 = ctx.bases_from_scalars(1,a).download(); g2,f2 = ctx.bases_from_scalars(2,bb).download()
for m in (256, 4096):
    t0=time.time(); gt = ctx.pairing_batch(g1[:m],f1[:m],g2[:m],f2[:m]); t1=time.time()
    print(f"pairing_batch n={m}: {1e3*(t1-t0):.1f} ms -> {m/(t1-t0):.0f} pairings/s")
    t0=time.time(); ml = ctx.miller_loop_batch(g1[:m],f1[:m],g2[:m],f2[:m]); t1=time.time()
    print(f"miller_loop_batch n={m}: {1e3*(t1-t0):.1f} ms")
