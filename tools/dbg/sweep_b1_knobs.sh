# B = 1 frame step against the persistent launch's nap lengths (CSM_PERSIST_TRICKLE x CSM_PERSIST_POLL), alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for tp in "${@:-8:1 8:0 10:0}"; do
  t=${tp%%:*}; p=${tp##*:}
  echo -n "trickle=$t poll=$p: "
  CSM_PERSIST_TRICKLE=$t CSM_PERSIST_POLL=$p timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['dominant_kernels']; print(d['ms_per_step'], [(x['kernel'], x['avg_us']) for x in k])"
done; done
