import time, numpy as np, sys
sys.path.insert(0,'.')
import bls12_381_amd as b
ctx = b.default_context()
n=1<<16
r = np.random.RandomState(1)
a = r.randint(0,256,size=(n,32),dtype=np.uint8); a[:,31]&=0x3f
bb = r.randint(0,256,size=(n,32),dtype=np.uint8); bb[:,31]&=0x3f
g1,f1 = ctx.bases_from_scalars(1,a).download(); g2,f2 = ctx.bases_from_scalars(2,bb).download()
for m in (1<<12, 1<<14, 1<<16):
    ctx.pairing_batch(g1[:256],f1[:256],g2[:256],f2[:256])
    t0=time.time(); gt = ctx.pairing_batch(g1[:m],f1[:m],g2[:m],f2[:m]); t1=time.time()
    print(f"pairing_batch n={m}: {1e3*(t1-t0):.1f} ms -> {m/(t1-t0):.0f} pairings/s (incl H2D/D2H)")
    t0=time.time(); ml = ctx.miller_loop_batch(g1[:m],f1[:m],g2[:m],f2[:m]); t1=time.time()
    print(f"miller_loop_batch n={m}: {1e3*(t1-t0):.1f} ms -> {m/(t1-t0):.0f}/s")
    t0=time.time(); fe = ctx.final_exponentiation_batch(ml); t1=time.time()
    print(f"final_exp_batch n={m}: {1e3*(t1-t0):.1f} ms -> {m/(t1-t0):.0f}/s")
    t0=time.time(); mm = ctx.multi_miller_loop(g1[:m],f1[:m],g2[:m],f2[:m]); t1=time.time()
    print(f"multi_miller_loop n={m}: {1e3*(t1-t0):.1f} ms -> {m/(t1-t0):.0f} terms/s")
# G2 msm
for logn in (16,18,20):
    n=1<<logn
    kb = r.randint(0,256,size=(n,32),dtype=np.uint8); kb[:,31]&=0x3f
    sb = r.randint(0,256,size=(n,32),dtype=np.uint8); sb[:,31]&=0x3f
    t0=time.time(); bases = ctx.bases_from_scalars(2,kb); t1=time.time()
    ctx.set_profiling(True)
    ctx.msm(bases,sb); t2=time.time(); ctx.msm(bases,sb); t3=time.time()
    print(f"G2 msm 2^{logn}: bases {t1-t0:.2f}s msm {1e3*(t3-t2):.1f} ms", {k: round(v,2) for k,v in ctx.last_msm_phase_ms().items()})
    ctx.set_profiling(False)
