cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ic
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|SQ_INST_CYCLES|SQ_WAIT_INST|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_ACTIVE_INST|SQ_INSTS_VALU\b|SQ_INSTS_SALU\b|INST_LEVEL|FIFO" | head -60 > gpurun_out/r02ic/avail.txt
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --kernel-trace --output-format csv -d gpurun_out/r02ic/pmc1 -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 6 --warmup 2 > gpurun_out/r02ic/b1.json 2> gpurun_out/r02ic/e1.txt
python - <<'PY'
import csv, glob, collections
for d in ["gpurun_out/r02ic/pmc1"]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            if r["Counter_Name"] == "SQC_ICACHE_REQ": n[k] += 1
    for k in sorted(agg, key=lambda k: -agg[k].get("SQC_ICACHE_MISSES", 0))[:8]:
        print(k, n[k], {c: round(v / max(n[k], 1)) for c, v in agg[k].items()})
PY
