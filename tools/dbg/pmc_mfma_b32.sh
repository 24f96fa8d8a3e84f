cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcb32; mkdir -p $O
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p -- python3 bench.py --batch 32 --no-cpu-baseline --no-extras --no-mimi --steps 10 --warmup 2 > $O/bench.json 2> $O/err.txt
python tools/pmc_summary.py mfma $O/p $O/pmc_mfma_util_b32.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --batch 32 --no-cpu-baseline --no-extras --no-mimi --steps 10 --warmup 2 (round 3, config 3)"
rm -rf $O/p; head -40 $O/pmc_mfma_util_b32.json
