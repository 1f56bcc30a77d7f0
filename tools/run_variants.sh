#!/bin/bash
# usage: tools/run_variants.sh <script.py>   -- runs the script once per bls12_381_amd/variants/*.so (copied over the library)
cp bls12_381_amd/libblsgpu.so /tmp/lib_orig.so
echo "== default"; python $1 2>&1 | grep -v amdgpu.ids
for v in bls12_381_amd/variants/*.so; do
  echo "== $v"; cp $v bls12_381_amd/libblsgpu.so; timeout 300 python $1 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_orig.so bls12_381_amd/libblsgpu.so
