# Round-end evidence run on the GPU box: full GPU suite, default bench line, rocprofv3 kernel stats of the bench, PMC traffic
# pass (FETCH_SIZE) and an MFMA-busy pass over a prompt prefill.  Summaries land in gpurun_out/r02a/ (copy to profiles/).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gputests.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 > $O/bench_under_rocprof.json 2>$O/rocprof_stats.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 > $O/bench_under_pmc.json 2>$O/rocprof_pmc.err
python tools/pmc_summary.py stats $O/stats $O/final_kernel_stats.csv > /dev/null
python tools/pmc_summary.py traffic $O/pmc $O/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 (round 2, persistent decoder)" > /dev/null
for S in 190 1334; do
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma$S -- python3 tools/prefill_prof.py $S 5 > $O/prefill_$S.txt 2>$O/rocprof_mfma$S.err
  python tools/pmc_summary.py mfma $O/mfma$S $O/pmc_mfma_util_S$S.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 tools/prefill_prof.py $S 5 (round 2)" > /dev/null
  timeout 100 python tools/prefill_prof.py $S 20 | tail -1 >> $O/prefill_times.txt
done
rm -rf $O/pmc $O/stats $O/mfma190 $O/mfma1334
cat $O/gputests.txt; cat $O/bench_default.json; cat $O/pmc_traffic.json; head -8 $O/final_kernel_stats.csv; cat $O/prefill_times.txt; head -30 $O/pmc_mfma_util_S190.json
