#!/bin/bash
# A/B on one box: run the short bench twice per library variant under bls12_381_amd/variants/
cp bls12_381_amd/libblsgpu.so /tmp/lib_orig.so
for rep in 1 2; do
for v in bls12_381_amd/variants/*.so; do
  cp $v bls12_381_amd/libblsgpu.so
  echo -n "$(basename $v) : "
  timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['ms_per_step'],3), round(d['roofline']['launch_ms'],3), {k: round(v,2) for k,v in d['msm_phase_ms'].items()})"
done; done
cp /tmp/lib_orig.so bls12_381_amd/libblsgpu.so
