# Kernel statistics, per-layer traces and PMC summaries of both convolutional network pairs; results under gpurun_out/conv_all/
cd $GRAFT_REPO_ROOT
O=gpurun_out/conv_all
rm -rf $O; mkdir -p $O
for spec in "omniglot 2048" "mnist_gmm 8192"; do
  set -- $spec
  for part in enc dec; do
    bash tools/profile_conv.sh $1 $2 $part > $O/$1_$part.log 2>&1 || exit 1
    cp gpurun_out/prof_conv/conv_kernel_stats.csv $O/$1_${part}_kernel_stats.csv
    cp gpurun_out/prof_conv/layers.txt $O/$1_${part}_layers.txt 2>/dev/null
  done
  timeout -k 10 200 python tools/bench_conv.py $1 $2 > $O/$1_bench.txt 2>&1 || exit 1
  bash tools/pmc_conv.sh $1 $2 > $O/$1_pmc.txt 2>&1 || exit 1
done
echo done
