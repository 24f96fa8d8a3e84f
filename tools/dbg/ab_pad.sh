cd $GRAFT_REPO_ROOT
CSM_DBG=8 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mimi --no-extras 2>&1 | grep "^addr"
for v in 0 64 256 1024 1536 1984 4096 0; do
  echo -n "PAD=$v KB: "
  CSM_DBG_PAD=$v timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['dominant_kernels']; print(d['ms_per_step'], [(x['kernel'], x['avg_us']) for x in k])"
done
