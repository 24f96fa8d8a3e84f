#!/bin/bash
# A/B of the B = 1 frame step: library $1 (default sesameai-tts_amd/lib/ab/libcsm_hip_base.so) against the shipped one, alternating on one box.
cd "$(dirname "$0")/../.."
BASE=${1:-sesameai-tts_amd/lib/ab/libcsm_hip_base.so}; N=${2:-4}
run() { CSM_HIP_LIB=$1 timeout 300 python bench.py --steps 125 --warmup 10 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/frame", [k["avg_us"] for k in (d["roofline"]["dominant_kernels"] or [])])'; }
for rep in $(seq $N); do
  echo "base $(run $BASE)"
  echo "new  $(run "")"
done
