cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_gmm10; mkdir -p gpurun_out/prof_gmm10
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/prof_gmm10 -o g -- python3 tools/time_whole_dataset.py configs/mnist_gmm/samples/lair_k4_dmis_gr_mult_replenish1_finalresample.yaml test_class.init_args.num_iterations=1000 > gpurun_out/prof_gmm10/out.txt 2>&1
echo rc=$?
tail -1 gpurun_out/prof_gmm10/out.txt
python3 - <<'P'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_gmm10/g_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows); 
# steady-state window: middle 50%
sub=rows[n//4:3*n//4]
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in sub)
span=int(sub[-1]["End_Timestamp"])-int(sub[0]["Start_Timestamp"])
print("launches total",n,"window launches",len(sub),"busy ms",busy/1e6,"span ms",span/1e6,"busy frac",busy/span, "avg kernel us", busy/len(sub)/1e3, "avg period us", span/len(sub)/1e3)
P
