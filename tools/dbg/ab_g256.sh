# k_gemm128 at 1,334 rows (rocprofv3 per-kernel times) with 128 x 128 blocks and, with CSM_G256_MIN_ROWS=1024, the opt-in 256 x 128 form;
# prefill + frame 0 of 32 x 190 rows; the bit-identity tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for thr in ${1:-1000000 1024}; do
  export CSM_G256_MIN_ROWS=$thr
  echo "== CSM_G256_MIN_ROWS=$thr"
  rm -rf /tmp/pfs; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfs -- python3 tools/prefill_prof.py 1334 10 > /tmp/pf_out.txt 2>&1
  f=$(find /tmp/pfs -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'k_gemm128' in n: print(f\"{n[:44]:44s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:8.1f} us total {float(r['TotalDurationNs'])/1e6:7.2f} ms\")
"
  python bench.py --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-mimi --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=32 x 190 rows: prefill + frame 0', d['prefill_plus_frame0_ms'], 'ms')"
done
unset CSM_G256_MIN_ROWS
python -m pytest tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -2
python -m pytest tests/test_frame_gpu.py -m gpu -x -q -k "prefill or prefix or prompt or golden or config5" 2>&1 | tail -2
