# B = 1 frame step of several checked-out trees (each with its own built library), alternating on one box
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  echo -n "$t: "
  (cd $t && timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-mimi --no-extras 2>/tmp/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" || tail -3 /tmp/err.txt)
done
