#!/bin/bash
# A/B of the prompt flash attention's key-group split (attn_flash.cuh AF_NG): the shipped library (AF_NG = 1: one wave group per block, blocks
# dispatched longest key walk first) against a build with -DAF_NG=2 (the key range of a block split over two wave groups;
# sesameai-tts_amd/lib/ab/libcsm_hip_ng2.so; hipcc ... -DAF_NG=2), alternating on one box.  Prefill wall time of one prompt (190 / 700 /
# 1334 rows) and of 32 prompts at once (32 x 190, 32 x 1334 rows).
cd "$(dirname "$0")/../.."
AB=sesameai-tts_amd/lib/ab/libcsm_hip_ng2.so
for rep in 1 2; do
  for lib in "$AB" ""; do
    tag=$([ -n "$lib" ] && echo "AF_NG=2" || echo "shipped (AF_NG=1)")
    for S in 190 700 1334; do
      echo -n "[$tag] "; CSM_HIP_LIB=$lib python tools/prefill_prof.py $S 12 2>&1 | tail -1
    done
    echo -n "[$tag] "; CSM_HIP_LIB=$lib python tools/dbg/prefill_b32.py 190 2>&1 | tail -1
    echo -n "[$tag] "; CSM_HIP_LIB=$lib python tools/dbg/prefill_b32.py 1334 2>&1 | tail -1
  done
done
