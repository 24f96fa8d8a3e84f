#!/bin/bash
# the batched persistent decoder's two nap knobs (CSM_PERSIST_TRICKLE x CSM_PERSIST_POLL) at one batch size: ms/step.  Usage: tools/sweep_pm_knobs.sh 32
cd "$(dirname "$0")/.."
B=${1:-32}
for t in 2 4 8 16; do for p in 0 1 2 4; do
  out=$(env CSM_PERSIST_TRICKLE=$t CSM_PERSIST_POLL=$p timeout 300 python bench.py --batch $B --steps 40 --warmup 5 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1)
  echo "B=$B trickle=$t poll=$p $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")')"
done; done
