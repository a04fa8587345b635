cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/prof_r3/stats -o r3 -- python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/prof_r3_bench.json 2> gpurun_out/prof_r3_stats.err &&
timeout -s KILL 150 rocprofv3 --output-format csv --pmc FETCH_SIZE -d gpurun_out/prof_r3/fetch -o r3 -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 > /dev/null 2> gpurun_out/prof_r3_fetch.err &&
timeout -s KILL 150 rocprofv3 --output-format csv --pmc WRITE_SIZE -d gpurun_out/prof_r3/write -o r3 -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 > /dev/null 2> gpurun_out/prof_r3_write.err &&
timeout -s KILL 150 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/prof_r3/sq -o r3 -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 > /dev/null 2> gpurun_out/prof_r3_sq.err
echo rc=$?
find gpurun_out/prof_r3 -name "*.csv" | head -20
