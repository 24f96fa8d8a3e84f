# Round-end evidence run on the GPU box: full GPU suite, default bench line, rocprofv3 kernel stats of the bench (B = 1 and B = 32), PMC
# traffic passes (FETCH_SIZE) for both, batch sweep (batched persistent decoder on / off), timelines of the persistent launches.
# Summaries land in gpurun_out/$1 (default r03a); copy what is to be judged to profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03a}; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/gpu_tests.txt
timeout 900 python bench.py > $O/final_bench.json 2> $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 > $O/final_bench_under_rocprof.json 2>$O/rocprof_stats.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 > $O/bench_under_pmc.json 2>$O/rocprof_pmc.err
python tools/pmc_summary.py stats $O/stats $O/final_kernel_stats.csv > /dev/null
python tools/pmc_summary.py traffic $O/pmc $O/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 (round 3)" > /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s32 -- python3 bench.py --batch 32 --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 > $O/b32_bench_under_rocprof.json 2> $O/b32_err.txt
python tools/pmc_summary.py stats $O/s32 $O/b32_kernel_stats.csv > /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc32 -- python3 bench.py --batch 32 --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 > $O/b32_bench_under_pmc.json 2>$O/b32_pmc.err
python tools/pmc_summary.py traffic $O/pmc32 $O/b32_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --batch 32 --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 (round 3, config 3)" > /dev/null
rm -rf $O/pmc $O/stats $O/s32 $O/pmc32
tools/sweep_batch.sh "1 2 3 4 8 16 24 32 64" > $O/batch_sweep.txt 2>&1
timeout 300 python tools/persist_timeline.py > $O/persist_timeline.txt 2>&1
for b in 32 16 4; do timeout 300 python tools/persist_m_timeline.py $b 2>&1 | grep -v amdgpu.ids > $O/persist_m_timeline_b$b.txt; done
cat $O/gpu_tests.txt; cut -c1-600 $O/final_bench.json; cat $O/pmc_traffic.json | head -20; head -6 $O/final_kernel_stats.csv | cut -c1-160; head -8 $O/b32_kernel_stats.csv | cut -c1-160; cat $O/batch_sweep.txt
