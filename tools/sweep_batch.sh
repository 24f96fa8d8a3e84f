#!/bin/bash
# frames/s vs utterances per GPU, batched persistent decoder on / off.  Usage: tools/sweep_batch.sh "2 4 8 32" > gpurun_out/sweep.txt
cd "$(dirname "$0")/.."
run() { # label, env, batch
  local out
  out=$(env $2 timeout 300 python bench.py --batch $3 --steps 40 --warmup 5 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1)
  echo "$1 B=$3 $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "frames/s", d["ms_per_step"], "ms/step", "prefill+f0", d["prefill_plus_frame0_ms"], "ms")')"
}
for b in ${1:-1 2 3 4 8 16 32 64}; do
  run persistent "CSM_PERSIST_M=1" $b
  if [ "$b" -ge 2 ] && [ "$b" -le 32 ]; then run chain "CSM_PERSIST_M=0" $b; fi
done
