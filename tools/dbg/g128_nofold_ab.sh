#!/bin/bash
# price the K-quarter fold of k_gemm128 (the fold keeps a prompt row's bits the same in k_mm32 / k_mmt / k_mmq / k_gemm128,
# i.e. whatever the row count of the call -- and doubles the accumulator registers, 128 of 238 VGPRs at the 128 x 128 tile).
# A = the shipped library; B = the same sources with -DG128_NOFOLD (`make -C sesameai-tts_amd/csrc nofold`): one K-ascending accumulation chain,
# 64 accumulator registers, parity with the other prompt kernels OFF.  Alternating on one box: prefill wall time at 1,334 rows and at 32 x 1,334
# = 42,688 rows, then rocprofv3 kernel stats of each (the gate/up launch is k_gemm128<3, ...>: 89.5 GFLOP per launch at 1,334 rows).
cd "$(dirname "$0")/../.."
R=$PWD
NF=$R/sesameai-tts_amd/lib/libcsm_hip_nofold.so
[ -f "$NF" ] || { echo "build it first: make -C sesameai-tts_amd/csrc nofold"; exit 1; }
for rep in 1 2 3; do
  for lib in "" "$NF"; do
    tag=$([ -n "$lib" ] && echo "no fold" || echo "shipped")
    echo -n "[$tag] "; CSM_HIP_LIB=$lib python tools/prefill_prof.py 1334 12 2>&1 | tail -1
    echo -n "[$tag] "; CSM_HIP_LIB=$lib python tools/dbg/prefill_b32.py 1334 2>&1 | tail -1
  done
done
cd /tmp && export TMPDIR=/tmp
for lib in "" "$NF"; do
  tag=$([ -n "$lib" ] && echo nofold || echo shipped)
  export CSM_HIP_LIB=$lib
  rm -rf /tmp/nf_$tag; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nf_$tag -- python3 $R/tools/prefill_prof.py 1334 10 > /dev/null 2>&1
  echo "== kernel stats, one 1,334-row prompt [$tag]: name, calls, average ns"
  python3 $R/tools/pmc_summary.py stats /tmp/nf_$tag /tmp/nf_$tag.csv > /dev/null; python3 -c "import csv,sys; [print(f'   {r[0][:60]:60s} calls {r[1]:>4s}  avg {float(r[3])/1e3:9.1f} us') for r in csv.reader(open(sys.argv[1])) if r and any(k in r[0] for k in ('k_gemm128','k_attn_flash','k_resid_norm'))]" /tmp/nf_$tag.csv
  rm -rf /tmp/nf_$tag; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nf_$tag -- python3 $R/tools/dbg/prefill_b32.py 1334 > /dev/null 2>&1
  echo "== kernel stats, 32 x 1,334 rows [$tag]"
  python3 $R/tools/pmc_summary.py stats /tmp/nf_$tag /tmp/nf_$tag.csv > /dev/null; python3 -c "import csv,sys; [print(f'   {r[0][:60]:60s} calls {r[1]:>4s}  avg {float(r[3])/1e3:9.1f} us') for r in csv.reader(open(sys.argv[1])) if r and any(k in r[0] for k in ('k_gemm128','k_attn_flash'))]" /tmp/nf_$tag.csv
done
