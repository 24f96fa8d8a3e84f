# prefill of one prompt (190 / 1334 rows) and of 32 x 190 rows for alternative builds of the library
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in "$@"; do
  export CSM_HIP_LIB=$GRAFT_REPO_ROOT/sesameai-tts_amd/lib/$lib
  echo "== $lib"
  python tools/prefill_prof.py 1334 20 2>&1 | tail -1
  python tools/prefill_prof.py 190 20 2>&1 | tail -1
  python bench.py --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=32 x 190 rows: prefill + frame 0', d['prefill_plus_frame0_ms'], 'ms')"
done; done
