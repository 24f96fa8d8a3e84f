cd $GRAFT_REPO_ROOT
for b in 32 8 2; do for v in 1 0 1 0; do
  echo -n "B=$b CSM_ATTN_MERGE=$v: "
  CSM_ATTN_MERGE=$v timeout 300 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")'
done; done
python -m pytest tests/test_frame_gpu.py -m gpu -x -q -k "batched or refill or continuous or config5 or golden" 2>&1 | tail -3
