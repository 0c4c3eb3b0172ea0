#!/bin/bash
# Same-box A/B of two builds of the library on the default bench workload: alternates VK_LIB=A / VK_LIB=B (boxes differ by +-1.5 %,
# so only pairs measured in one call are compared).   usage: bash tests/diag/ab_bench.sh LIB_A LIB_B [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq 1 $R); do
  for L in "$A" "$B"; do
    VK_LIB=$L timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --api-steps 0 > /tmp/ab_line.json 2>/tmp/ab_err.log || { echo "bench failed for $L"; tail -5 /tmp/ab_err.log; exit 1; }
    python - "$L" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab_line.json').read().strip().splitlines()[-1])
r = d.get('roofline') or {}
print(f"{sys.argv[1].split('/')[-1]:<28} {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} img/s  frac {r.get('frac')}")
PY
  done
done
