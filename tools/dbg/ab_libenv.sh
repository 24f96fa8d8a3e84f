# usage: ab_libenv.sh lib:ENV=VAL ...
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  lib=${spec%%:*}; ev=${spec#*:}
  echo -n "$lib $ev: "
  env CSM_HIP_LIB=$GRAFT_REPO_ROOT/sesameai-tts_amd/lib/$lib $ev timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['dominant_kernels']; print(d['ms_per_step'], [(x['kernel'], x['avg_us']) for x in k])"
done
