cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  for w in bf16 fp8; do
  echo -n "$lib $w: "
  env CSM_HIP_LIB=$GRAFT_REPO_ROOT/sesameai-tts_amd/lib/$lib timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-mimi --no-extras --weights $w 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
