# The YAML driver as 2 processes on a box with ONE GPU (CSVAE_DIST_REHEARSAL=1: both ranks on GPU 0, gloo gathers on
# host copies) against the same configuration as one process: every artefact file must hold the same arrays.
#   bash tools/rehearse_sharded_driver.sh [config.yaml]
CFG=${1:-configs/uci_miniboone/sampling_u50/lair_k4_dmis_gr_mult_1replenish.yaml}
KEY=num_iterations; case $CFG in *acmwg*) KEY=num_mwg_iterations;; esac
O=/tmp/cs_rehearse; rm -rf $O; mkdir -p $O
SET="--set data.synthetic_num_test=400 --set data.num_first_datapoints_test=301 --set data.batch_size=64 --set test_class.init_args.$KEY=6"
timeout -k 10 200 python test.py --config=$CFG $SET --set trainer.default_root_dir=$O/one || exit 1
CSVAE_DIST_REHEARSAL=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 test.py --config=$CFG $SET --set trainer.default_root_dir=$O/two || exit 1
python - <<'P'
import glob, os, sys
import numpy as np
one = sorted(glob.glob("/tmp/cs_rehearse/one/**/*.npz", recursive=True))
two = sorted(glob.glob("/tmp/cs_rehearse/two/**/*.npz", recursive=True))
assert one and [os.path.basename(p) for p in one] == [os.path.basename(p) for p in two], (one, two)
assert len({os.path.dirname(p) for p in two}) == 1, "the sharded run wrote into more than one directory"
for a, b in zip(one, two):
    fa, fb = np.load(a), np.load(b)
    assert sorted(fa.files) == sorted(fb.files), (a, fa.files, fb.files)
    for k in fa.files:
        x, y = fa[k], fb[k]
        assert x.shape == y.shape and x.dtype == y.dtype, (a, k, x.shape, y.shape)
        if os.path.basename(a) == "metrics_summary.npz":
            assert np.allclose(x, y, rtol=1e-5, atol=1e-6, equal_nan=True), (a, k)
        else:
            assert np.array_equal(x, y, equal_nan=True), (a, k)
print(f"sharded (2 ranks) == one process: {len(one)} files identical")
P
