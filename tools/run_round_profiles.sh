cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02a/gputests.txt
timeout 600 python bench.py > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02a/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 > gpurun_out/r02a/bench_under_rocprof.json 2>gpurun_out/r02a/rocprof_stats.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02a/pmc -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 > gpurun_out/r02a/bench_under_pmc.json 2>gpurun_out/r02a/rocprof_pmc.err
python tools/pmc_summary.py stats gpurun_out/r02a/stats gpurun_out/r02a/final_kernel_stats.csv > /dev/null
python tools/pmc_summary.py traffic gpurun_out/r02a/pmc gpurun_out/r02a/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 (round 2, persistent decoder)" > /dev/null
rm -rf gpurun_out/r02a/pmc gpurun_out/r02a/stats
cat gpurun_out/r02a/gputests.txt; cat gpurun_out/r02a/bench_default.json; cat gpurun_out/r02a/pmc_traffic.json; head -8 gpurun_out/r02a/final_kernel_stats.csv
