# PMC passes over the ResNet layer programs: bash tools/pmc_conv.sh [omniglot|mnist_gmm] [rows]
# (one rocprofv3 run per counter group; summaries: python3 tools/pmc_kernel.py gpurun_out/pmc_conv <kernel substring>)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
NET=${1:-omniglot}
ROWS=${2:-2048}
O=gpurun_out/pmc_conv
rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/sq -o p -- python3 tools/bench_conv.py $NET $ROWS > $O/sq.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/fetch -o p -- python3 tools/bench_conv.py $NET $ROWS > $O/fetch.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/write -o p -- python3 tools/bench_conv.py $NET $ROWS > $O/write.log 2>&1
echo rc=$?
for k in "halo_split_kernel<4, 1, 8, false, 2>" "halo_split_kernel<2, 2, 4, false, 2>" "halo_split_kernel<2, 2, 8, false, 2>" "halo_split_kernel<4, 1, 8, true, 2>" "halo_split_kernel<2, 2, 8, true, 2>" "halo_split_kernel<4, 1, 8, false, 3>" "halo_split_kernel<2, 2, 8, false, 3>" "halo_resident_kernel<8, 2>" "halo_resident_kernel<8, 3>" "conv_mfma_split_kernel<4, 2>" "conv_mfma_split_kernel<2, 2>" "conv_mfma_split_kernel<1, 2>" "conv_mfma_split_kernel<4, 3>" "cs_image_amax_kernel"; do echo "== $k"; python3 tools/pmc_kernel.py $O "$k"; done
