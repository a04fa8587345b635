cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_conv
rm -rf $O; mkdir -p $O
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/p1 -o p -- python3 tools/bench_conv_layer.py 2048 4 128,64 > $O/p1.log 2>&1 &&
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_WAIT_ANY SQ_BUSY_CYCLES -d $O/p2 -o p -- python3 tools/bench_conv_layer.py 2048 4 128,64 > $O/p2.log 2>&1
echo rc=$?
python3 tools/pmc_kernel.py $O "halo_split_kernel<4, 1, 8>"
echo ---
python3 tools/pmc_kernel.py $O "halo_split_kernel<2, 2, 8>"
