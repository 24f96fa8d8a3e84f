#!/bin/bash
# A/B of the B = 1 frame step between two settings of ONE environment switch, alternating on one box:  tools/dbg/ab_env_b1.sh CSM_PERSIST_WARM 0 1 [reps]
cd "$(dirname "$0")/../.."
V=$1; A=$2; B=$3; N=${4:-4}
run() { env $V=$1 timeout 300 python bench.py --steps 125 --warmup 10 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/frame", [k["avg_us"] for k in (d["roofline"]["dominant_kernels"] or [])])'; }
for rep in $(seq $N); do
  echo "$V=$A $(run $A)"
  echo "$V=$B $(run $B)"
done
