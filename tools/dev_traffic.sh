cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/traffic; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
cd $R; python3 tools/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json | grep -A4 solve_kernel | head -8
python3 bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'])"
python3 tests/dev/dev_variant_check.py 2>&1 | grep -E "^[0-4] variant 0" | cut -c1-150
