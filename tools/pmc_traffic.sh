cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_traffic; rm -rf $O; mkdir -p $O
timeout -s KILL 150 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/fetch -o r -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 > /dev/null 2> $O/fetch.err &&
timeout -s KILL 150 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/write -o r -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 > /dev/null 2> $O/write.err
echo rc=$?
for k in "cs_mlp_split16_kernel<16, 4, 8, 3>" "cs_mlp_split16_kernel<16, 6, 8, 3>" "cs_mlp_split16_kernel<16, 4, 8, 2>" "cs_lair_resample_kernel"; do echo "== $k"; python3 tools/pmc_kernel.py $O "$k"; done
timeout -k 10 100 python3 tools/trace_phases.py | grep "ms per"
