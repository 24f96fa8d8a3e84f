# ms per frame step at batch $1 for alternative builds of the library (CSM_HIP_LIB), alternating:  bash tools/dbg/ab_libs_b.sh 32 libA.so libB.so libA.so libB.so
cd $GRAFT_REPO_ROOT
B=$1; shift
for lib in "$@"; do
  echo -n "B=$B $lib: "
  CSM_HIP_LIB=$GRAFT_REPO_ROOT/sesameai-tts_amd/lib/$lib timeout 300 python bench.py --batch $B --steps 40 --warmup 5 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['dominant_kernels']; print(d['ms_per_step'], [(x['kernel'], x['avg_us']) for x in (k or [])])"
done
