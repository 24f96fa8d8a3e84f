# Round-end evidence run on the GPU box (rounds 4-6): full GPU suite with the parity prints, default bench line, rocprofv3 kernel stats of the bench
# at B = 1 / B = 32 / B = 1 with the fp8 weight stream, PMC traffic passes (FETCH_SIZE) for the three, kernel stats of the S = 190 and S = 1334
# prefills, batch sweep, timelines of the persistent launches, soaks.  Summaries land in gpurun_out/$1 (default r06a); copy what is to be
# judged to profiles/.     gpurun --timeout 2700 -- 'bash tools/run_round_profiles.sh r06a'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06a}; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -s > $O/gpu_tests_full.txt 2>&1
grep -E "passed|failed" $O/gpu_tests_full.txt | tail -2 > $O/gpu_tests.txt
grep -E "dlogit|excused|identical|\[parity\]|\[decisive|\[possweep\]|\[soak\]|\[8 ranks\]|composed" $O/gpu_tests_full.txt | cut -c1-400 > $O/parity_margins.txt
timeout 900 python bench.py > $O/final_bench.json 2> $O/bench_default.err
# the N > 1 control flow at full size on a one-rank RCCL process group, timed stage by stage (8 x its host-side set-up must fit the driver's 1,800 s)
( TIMEFORMAT='whole process: %R s wall, %U s user, %S s sys'; time BENCH_RCCL_WORLD1=1 python bench.py --steps 20 --warmup 5 --extra-steps 5 > $O/rccl_one_rank_bench.json ) 2> $O/rccl_one_rank_bench.err
grep -E "^\[bench|whole process" $O/rccl_one_rank_bench.err > $O/rccl_one_rank_stages.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
prof() {   # prof <name> <bench args...>: kernel stats + FETCH_SIZE pass of one bench configuration
    local name=$1; shift
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$name -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 "$@" > $O/${name}_bench_under_rocprof.json 2> $O/${name}_rocprof.err
    python tools/pmc_summary.py stats $O/st_$name $O/${name}_kernel_stats.csv > /dev/null
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$name -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 "$@" > $O/${name}_bench_under_pmc.json 2> $O/${name}_pmc.err
    python tools/pmc_summary.py traffic $O/pmc_$name $O/${name}_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 $* (round 6)" > /dev/null
    rm -rf $O/st_$name $O/pmc_$name
}
prof final
prof b32 --batch 32
prof fp8 --weights fp8
cp $O/final_pmc_traffic.json $O/pmc_traffic.json
# matrix-core busy of the B = 32 frame step ("an MFMA-busy pass for B = 32 at HEAD")
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mf_b32 -- python3 bench.py --batch 32 --no-cpu-baseline --no-extras --no-mimi --steps 10 --warmup 2 > /dev/null 2>&1
python tools/pmc_summary.py mfma $O/mf_b32 $O/pmc_mfma_util_b32.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --batch 32 --no-cpu-baseline --no-extras --no-mimi --steps 10 --warmup 2 (round 6, config 3)" > /dev/null
rm -rf $O/mf_b32
# GPU time of ONE csm_create (the table builds): kernel stats of a process that only creates a B = 1 handle
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_create -- python3 tools/dbg/create_only.py > $O/create_only.txt 2>&1
python tools/pmc_summary.py stats $O/st_create $O/create_kernel_stats.csv > /dev/null; rm -rf $O/st_create
for S in 190 1334; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pf_$S -- python3 tools/prefill_prof.py $S 10 > $O/prefill_S$S.txt 2>&1
    python tools/pmc_summary.py stats $O/pf_$S $O/prefill_S${S}_kernel_stats.csv > /dev/null
    timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pfm_$S -- python3 tools/prefill_prof.py $S 4 > /dev/null 2>&1
    python tools/pmc_summary.py mfma $O/pfm_$S $O/pmc_mfma_util_S$S.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/prefill_prof.py $S 4 (round 6)" > /dev/null
    rm -rf $O/pf_$S $O/pfm_$S
done
R=$GRAFT_REPO_ROOT; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/mimi_prof -o mimi -- python3 $R/tools/mimi_prof.py > $R/$O/mimi_prof_under_rocprof.txt 2>&1)
python3 tools/dbg/mimi_chunk_timeline.py $O/mimi_prof/mimi_results.db > $O/mimi_chunk10_timeline.txt 2>&1; rm -rf $O/mimi_prof
timeout 300 python3 tools/mimi_prof.py 2>&1 | grep -v amdgpu.ids > $O/mimi_times.txt
tools/sweep_batch.sh "1 2 4 8 16 32" > $O/batch_sweep.txt 2>&1
timeout 300 python tools/persist_timeline.py > $O/persist_timeline.txt 2>&1
timeout 300 python tools/persist_m_timeline.py 32 2>&1 | grep -v amdgpu.ids > $O/persist_m_timeline_b32.txt
timeout 600 python tools/soak_attn_merge.py 1000 2 32,8 > $O/soak_attn_merge.txt 2>&1
timeout 600 python tools/soak_persist_m.py 200 2 2,17,32 > $O/soak_persist_m.txt 2>&1
timeout 300 python tools/dbg/ref_loop_breakdown.py 60 2>&1 | grep -v amdgpu.ids > $O/ref_loop_breakdown.txt
cat $O/gpu_tests.txt; python tools/dbg/bench_summary.py $O/final_bench.json; cat $O/pmc_traffic.json | head -12; head -5 $O/final_kernel_stats.csv | cut -c1-160; head -6 $O/b32_kernel_stats.csv | cut -c1-160; cat $O/batch_sweep.txt; tail -2 $O/soak_attn_merge.txt
