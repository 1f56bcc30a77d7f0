#!/bin/bash
# A/B on one box: run a script twice per library variant under bls12_381_amd/variants/   usage: tools/ab_run.sh tools/quick_pair2.py
cp bls12_381_amd/libblsgpu.so /tmp/lib_orig.so
for rep in 1 2; do
for v in bls12_381_amd/variants/*.so; do
  cp $v bls12_381_amd/libblsgpu.so
  echo "== $(basename $v)"; timeout 300 python $1 2>&1 | grep -v amdgpu.ids
done; done
cp /tmp/lib_orig.so bls12_381_amd/libblsgpu.so
