#!/bin/bash
# register / scratch usage of one kernel template of a .cuh, compiled alone:  tools/kres.sh <header> <kernel-name-pattern> [extra hipcc flags]
H=$1; PAT=$2; shift 2
cd "$(dirname "$0")/../sesameai-tts_amd/csrc"
printf '#include "%s"\ntemplate __global__ void k_dec_persist_m<1>(const DecPersistMArgs);\ntemplate __global__ void k_dec_persist_m<2>(const DecPersistMArgs);\n' "$H" > /tmp/kres_tu.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. "$@" -c -Rpass-analysis=kernel-resource-usage -save-temps=obj -o /tmp/kres_tu.o /tmp/kres_tu.hip 2>&1 \
  | grep -A11 "Function Name: .*$PAT" | grep "Name\|VGPRs\|Scratch\|Spill" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
