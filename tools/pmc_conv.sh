# PMC passes over one convolution shape: bash tools/pmc_conv.sh <channels:resolution> <kernel name substring>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
SHAPE=${1:-128:14}
KERNEL=${2:-halo_split_kernel}
O=gpurun_out/pmc_conv
rm -rf $O; mkdir -p $O
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/p1 -o p -- python3 tools/bench_conv_layer.py 2006 4 $SHAPE > $O/p1.log 2>&1 &&
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_BUSY_CYCLES -d $O/p2 -o p -- python3 tools/bench_conv_layer.py 2006 4 $SHAPE > $O/p2.log 2>&1 &&
timeout -s KILL 90 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/p3 -o p -- python3 tools/bench_conv_layer.py 2006 4 $SHAPE > $O/p3.log 2>&1 &&
timeout -s KILL 90 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/p4 -o p -- python3 tools/bench_conv_layer.py 2006 4 $SHAPE > $O/p4.log 2>&1 &&
timeout -s KILL 90 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $O/p5 -o p -- python3 tools/bench_conv_layer.py 2006 4 $SHAPE > $O/p5.log 2>&1
echo rc=$?
python3 tools/pmc_kernel.py $O "$KERNEL"
