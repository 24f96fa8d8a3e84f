#!/bin/bash
# B = 1 frame step: the 16-launch backbone against the one-launch form (CSM_BB_STACK=1) at several request credits, alternating on one box
cd "$(dirname "$0")/../.."
run() { env CSM_BB_STACK=$1 CSM_BB_STACK_CREDIT=$2 timeout 300 python bench.py --steps 125 --warmup 10 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/frame", [k["avg_us"] for k in (d["roofline"]["dominant_kernels"] or [])])'; }
for rep in 1 2; do
  echo "layers   $(run 0 0)"
  for c in ${@:-0 8 16}; do echo "stack c=$c $(run 1 $c)"; done
done
