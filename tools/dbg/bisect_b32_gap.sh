#!/bin/bash
# config 3 (B = 32) sits at 1.03x the oracle's bf16-vs-fp32 gap where B = 1 / B = 4 hold 1x.  Which op adds the excess?
# Runs tests/test_frame_gpu.py::test_csm1b_config3_batch32_vs_golden with one batched-path switch flipped at a time and prints its
# "config 3 (B=32): max|dlogit| frame 0 .. frame 1 .." line (frame 0 = 6,080-row prefill + depth pass, frame 1 = the 32-row decode step).
#   gpurun -- 'bash tools/dbg/bisect_b32_gap.sh > gpurun_out/r4/bisect_b32_gap.txt 2>&1'
cd "$(dirname "$0")/../.."
run() {
    echo "== $1"
    env $2 python -m pytest tests/test_frame_gpu.py -q -s -k "test_csm1b_config3_batch32_vs_golden" 2>&1 | grep -E "config 3 \(B=32\)|passed|failed|Error" | cut -c1-400
}
run "default (k_gemm128 prefill, operand-order activations, batched persistent decoder, split-key attention merged in kernel)" "X=1"
run "CSM_PERSIST_M=0      (depth decoder as the launch chain)" "CSM_PERSIST_M=0"
run "CSM_G128_MIN_ROWS=1000000   (prefill through k_mm32 32x32 tiles instead of the LDS-tiled 128x128 kernel)" "CSM_G128_MIN_ROWS=1000000"
run "CSM_XPACK=0          (row-major activations on the decode step)" "CSM_XPACK=0"
run "CSM_ATTN_MERGE=0     (split-key attention merged by a second launch)" "CSM_ATTN_MERGE=0"
run "CSM_SLAB_K=100000    (no split-K of the residual projections on the decode step: kg = 1)" "CSM_SLAB_K=100000"
run "CSM_WIDE=0           (GEMV kernels for every batch size: the B = 1 arithmetic, row by row)" "CSM_WIDE=0"
run "CSM_PERSIST_M=0 CSM_WIDE=0" "CSM_PERSIST_M=0 CSM_WIDE=0"
