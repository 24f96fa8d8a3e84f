#!/bin/bash
# A/B of the B = 32 (and 17 / 24) frame step: library $1 (default: sesameai-tts_amd/lib/ab/libcsm_hip_base.so) against the shipped one, alternating.
cd "$(dirname "$0")/../.."
BASE=${1:-sesameai-tts_amd/lib/ab/libcsm_hip_base.so}
run() { CSM_HIP_LIB=$1 timeout 300 python bench.py --batch $2 --steps 60 --warmup 5 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", [k["avg_us"] for k in (d["roofline"]["dominant_kernels"] or [])])'; }
for rep in 1 2 3; do
  for b in ${2:-32}; do
    echo "base B=$b $(run $BASE $b)"
    echo "new  B=$b $(run "" $b)"
  done
done
