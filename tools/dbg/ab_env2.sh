cd $GRAFT_REPO_ROOT
run() { echo -n "$*: "; env "$@" timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['dominant_kernels']; print(d['ms_per_step'], [(x['kernel'], x['avg_us']) for x in k])"; }
run CSM_X=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run CSM_BB_R2=0 HIP_FORCE_DEV_KERNARG=1
run CSM_BB_R2=0 HIP_FORCE_DEV_KERNARG=0
