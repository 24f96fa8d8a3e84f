# FETCH_SIZE passes of the bench at B = 1, B = 32 and with the fp8 weight stream -> gpurun_out/$1/{pmc_traffic,b32_pmc_traffic,fp8_pmc_traffic}.json
# (each carries the digest of the kernel sources it measured; bench.py reports profiles/r*/pmc_traffic.json only if that digest matches).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04p}; mkdir -p $O
pass() { local name=$1; shift
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$name -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 "$@" > $O/${name}_bench_under_pmc.json 2> $O/${name}_pmc.err
    python tools/pmc_summary.py traffic $O/pmc_$name $O/$name.json "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --no-mimi --steps 20 --warmup 2 $* (round 4)" > /dev/null
    rm -rf $O/pmc_$name; }
pass pmc_traffic
pass b32_pmc_traffic --batch 32
pass fp8_pmc_traffic --weights fp8
python -c "import bench; print('digest', bench.csrc_digest())"; grep -h "csrc_digest\|traffic_bytes" $O/*.json
