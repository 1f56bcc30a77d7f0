mkdir -p gpurun_out/r06b
for cfg in "X=1" "HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0" "HSA_SCRATCH_SINGLE_LIMIT_ASYNC=134217728" "HSA_SCRATCH_SINGLE_LIMIT=4294967296"; do
  tag=$(echo $cfg | tr '=' '_')
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06b/$tag.json 2> gpurun_out/r06b/$tag.err
  echo "$cfg rc=$?"
  python - <<PY
import json
l=open("gpurun_out/r06b/$tag.json").read().strip().splitlines()[-1]
d=json.loads(l)
print({k:d.get(k) for k in ["ms_per_step","pairing_ms","mml_ms","equations_per_s","prepared_equations_per_s","prepared_equations_speedup"]}, d["extras"]["bls_verify_from_bytes"]["ms"])
PY
done
