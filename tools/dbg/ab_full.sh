cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in .r2tree . .r2tree .; do
  (cd $t && rm -rf /tmp/st && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 40 --warmup 4 > /tmp/bench_out.txt 2>&1; tail -1 /tmp/bench_out.txt | python3 -c "import json,sys; print('ms_per_step', json.loads(sys.stdin.read())['ms_per_step'])"; f=$(find /tmp/st -name "*kernel_stats.csv" | head -1); echo "== $t"; python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print(f"{r['Name'][:60]:60s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.2f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
)
done
