# Kernel statistics of the ResNet encoder / decoder layer programs: bash tools/profile_conv.sh [omniglot|mnist_gmm] [rows] [enc|dec]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
NET=${1:-omniglot}
ROWS=${2:-2048}
O=gpurun_out/prof_conv
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O -o conv -- python3 tools/bench_conv.py $NET $ROWS $3 > $O/bench.txt 2> $O/err.txt
echo rc=$?
cat $O/bench.txt
if [ -n "$3" ]; then python3 tools/conv_trace_layers.py $O/conv_kernel_trace.csv > $O/layers.txt; cat $O/layers.txt; fi
