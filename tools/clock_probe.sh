#!/bin/bash
# Samples GPU clock / power / utilisation while the headline bench runs (is the VALU-bound kernel clock- or power-limited?).
#   bash tools/clock_probe.sh > gpurun_out/clock_probe.log
python bench.py --steps 4000 --warmup 5 --no-cpu-baseline --no-extras > /tmp/clock_probe_bench.log 2>&1 &
BPID=$!
n=0
while kill -0 $BPID 2>/dev/null && [ $n -lt 400 ]; do
  out=$(rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power \(W\)|GPU use" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';')
  case "$out" in *"GPU use (%): 0"*) ;; *) echo "t=$n $out";; esac
  n=$((n+1)); sleep 0.3
done
wait $BPID
tail -1 /tmp/clock_probe_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench value', d['value'], 'ms_per_step', d['ms_per_step'], 'acc launch_ms', d['roofline']['launch_ms'], 'peak TMAC32/s', d['roofline']['peak'])"
echo "--- idle"
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | head -4
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
