#!/bin/bash
# per-kernel times of an S-row prefill under rocprofv3 for the environment given in front of the call:
#   gpurun -- 'CSM_G128_2D_TILES=9999 bash tools/dbg/prefill_kernels.sh 1334 tag'
S=${1:-1334}; TAG=${2:-x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pfk_${TAG}_$S; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/prefill_prof.py $S 6 > $O/run.txt 2>&1
cd $R
F=$(find $O -name "*kernel_stats.csv" | head -1)
echo "== S=$S $TAG: $(grep 'prefill S' $O/run.txt)"
python3 - "$F" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Name"] for k in ("k_gemm128", "k_attn_flash", "k_resid_norm<", "k_mmt", "k_mm32", "k_mmq"))]
tot = 0.0
for r in rows:
    n = int(r["Calls"]); avg = float(r["AverageNs"]) / 1e3
    if n < 16: continue
    per_layer = avg * n / (n // 96 * 96 if n >= 96 else n) if False else avg
    print(f"   {avg:8.1f} us x {n:4d}  {r['Name'][:70]}")
PY
