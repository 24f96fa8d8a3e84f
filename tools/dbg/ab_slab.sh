cd $GRAFT_REPO_ROOT
run() { echo -n "$*: "; env "$@" timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['dominant_kernels']; print(d['ms_per_step'], [(x['kernel'], x['avg_us']) for x in k])"; }
run CSM_XSLAB=0
for off in 0 4 8 16 32 64 128 512; do run CSM_XSLAB_OFF=$off; done
for al in 1 16 64 256; do run CSM_XSLAB_ALIGN=$al; done
run CSM_XSLAB=0
