cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in .r2tree .; do
  (cd $t && rm -rf /tmp/st && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 40 --warmup 4 > /dev/null 2>&1; f=$(find /tmp/st -name "*kernel_stats.csv" | head -1); echo "== $t"; grep "k_dec_persist\|k_bb_layer\|k_advance\|k_sample\|k_embed\|k_gemv<1, 4\|k_gemv<2" $f | cut -d, -f1-4 | cut -c1-110)
done
