# per-kernel times of a 1334-row prompt prefill (rocprofv3 --kernel-trace --stats) for alternative builds of the library
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  export CSM_HIP_LIB=$GRAFT_REPO_ROOT/sesameai-tts_amd/lib/$lib
  echo "== $lib"
  rm -rf /tmp/pfs; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfs -- python3 tools/prefill_prof.py 1334 10 > /tmp/pf_out.txt 2>&1
  f=$(find /tmp/pfs -name "*kernel_stats.csv" | head -1); python3 -c "
import csv,sys
tot=0
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if any(k in n for k in ('k_gemm128','k_attn_flash','k_resid_norm<','k_mm32','k_mmt','k_mmq')):
        print(f\"{n[:44]:44s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:8.1f} us total {float(r['TotalDurationNs'])/1e6:7.2f} ms\")
"
done
