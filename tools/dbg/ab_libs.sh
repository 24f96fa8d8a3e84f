# ms per frame step (B = 1) and the two dominant kernels' live timings for alternative builds of the library (CSM_HIP_LIB)
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  echo -n "$lib: "
  CSM_HIP_LIB=$GRAFT_REPO_ROOT/sesameai-tts_amd/lib/$lib timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['dominant_kernels']; print(d['ms_per_step'], [(x['kernel'], x['avg_us']) for x in k])"
done
