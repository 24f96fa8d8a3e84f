# B = 1 frame step under the CSM_DBG bisect knobs (temporary)
cd $GRAFT_REPO_ROOT
for v in 0 1 2 4 7 0 7; do
  echo -n "CSM_DBG=$v: "
  CSM_DBG=$v timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['dominant_kernels']; print(d['ms_per_step'], [(x['kernel'], x['avg_us']) for x in k])"
done
echo -n ".r2tree: "; (cd .r2tree && timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
