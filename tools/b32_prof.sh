cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b32
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b32/s -- python3 bench.py --batch 32 --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 > gpurun_out/b32/bench.json 2> gpurun_out/b32/err.txt
python tools/pmc_summary.py stats gpurun_out/b32/s gpurun_out/b32/b32_kernel_stats.csv | cut -c1-150 | head -24
rm -rf gpurun_out/b32/s
cut -c1-200 gpurun_out/b32/bench.json
