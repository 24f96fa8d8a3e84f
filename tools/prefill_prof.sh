# per-kernel picture of a prompt prefill (rocprofv3 --kernel-trace --stats): bash tools/prefill_prof.sh [S]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S=${1:-190}
mkdir -p gpurun_out/pf
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf/s -- python3 tools/prefill_prof.py $S 10 > gpurun_out/pf/out.txt 2>&1
tail -1 gpurun_out/pf/out.txt
python tools/pmc_summary.py stats gpurun_out/pf/s gpurun_out/pf/stats_$S.csv | cut -c1-150 | grep -v "k_gemv\|rocclr\|at::native"
rm -rf gpurun_out/pf/s
