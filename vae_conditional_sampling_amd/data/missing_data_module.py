"""Test-split data module with the reference's batch contract.

Counterpart of ``MissingDataModule`` (irwg/data/missing_data_module.py:187-470) restricted to
what the sampling driver needs: ``setup('test')`` builds the test split, applies the
missingness pattern (``uniform``, image ``quadrants`` or ``nan_miss``), optionally drops
fully-missing rows, appends the data index and keeps the first ``num_first_datapoints_test`` points; ``test_dataloader()`` yields
``(X float32 [B,D], M bool [B,D] (True = observed), I int64 [B])`` in order (no shuffling).

Datasets: a ``.npy`` file ``{data_root}/{dataset}/test.npy`` when present; otherwise, when
``synthetic`` is true, seeded standard-normal data with the dataset's dimensionality (the UCI
sets are standardised, irwg/data/uci_miniboone.py:44-53) -- the real data files are not
distributed with this package.

Sharding: ``rank`` / ``world_size`` split the (filtered, index-ordered) test set into
contiguous blocks, one per process; ``shard_offset`` is the global index of the first local
example and keys the counter RNG.
"""
import math
import os

import numpy as np
import torch

from ..utils.hparams import AttrDict

DATASET_DIMS = {"uci_miniboone": 43, "uci_hepmass": 21, "uci_gas": 8, "uci_power": 6, "toy_grid2dmog4": 2,
                "omniglot28x28_fbin": 784, "mnist_gmm": 196, "mnist_gmm_custtest0": 196,
                "toy_grid2dmog_custtest0": 2}
DATASET_TEST_SIZES = {"uci_miniboone": 3648, "uci_hepmass": 174987, "uci_gas": 105206, "uci_power": 204928,
                      "toy_grid2dmog4": 14, "omniglot28x28_fbin": 8070, "mnist_gmm": 10, "mnist_gmm_custtest0": 10,
                      "toy_grid2dmog_custtest0": 14}
BINARY_DATASETS = ("omniglot28x28_fbin",)  # synthetic stand-in: Bernoulli(0.2) pixels


def toy_custom_test_points():
    """The 14 fixed test points of the toy configuration, one coordinate observed at
    -2, -1, -0.5, 0, 0.5, 1, 2 and the other missing (NaN) (irwg/data/toy.py:605-621)."""
    v = torch.tensor([-2., -1., -.5, 0., .5, 1., 2.])
    nan = torch.full_like(v, float("nan"))
    return torch.cat([torch.stack([nan, v], 1), torch.stack([v, nan], 1)])


def uniform_missingness_mask(shape, total_miss, rng):
    """Exactly ``int(total_miss * N*D)`` missing cells placed by a random permutation of all
    cells (irwg/data/uniform_missingness_provider.py:50-63).  True = observed."""
    total = int(np.prod(shape))
    mask = torch.ones(total, dtype=torch.bool)
    mask[:int(total_miss * total)] = 0
    perm = torch.randperm(total, generator=rng)
    return mask[perm].reshape(*shape)


def quadrant_missingness_mask(n, img_dims, total_miss, rng):
    """``int(total_miss * 4)`` of the four image quadrants missing per example, drawn without
    replacement (irwg/data/quadrant_missingness_provider.py:57-74).  True = observed."""
    assert total_miss in (0., 0.25, 0.5, 0.75), "Missingness fraction can only be 0., 0.25, 0.5, or 0.75!"
    H, W = int(img_dims[0]), int(img_dims[1])
    mask = torch.ones(n, H, W, dtype=torch.bool)
    if total_miss > 0.:
        q = torch.multinomial(torch.ones(n, 4), num_samples=int(total_miss * 4), replacement=False, generator=rng)
        hh, hw = H // 2, W // 2
        for quad, (rows, cols) in enumerate(((slice(0, hh), slice(0, hw)), (slice(0, hh), slice(hw, W)),
                                             (slice(hh, H), slice(0, hw)), (slice(hh, H), slice(hw, W)))):
            sel = (q == quad).any(-1)
            mask[sel, rows, cols] = False
    return mask.reshape(n, -1)


def shard_range(n, rank, world_size):
    """Contiguous block [lo, hi) of rank `rank` when n examples are split over world_size ranks
    (blocks differ by at most one example)."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class MissingDataModule:
    def __init__(self, dataset, batch_size, missingness="uniform", total_miss_test=0.0, setup_seed=0,
                 data_root="./data", num_first_datapoints_test=None, filter_fully_missing_test=False,
                 synthetic=False, synthetic_num_test=None, rank=0, world_size=1, **unused):
        self.hparams = AttrDict(dataset=dataset, batch_size=batch_size, missingness=missingness,
                                total_miss_test=total_miss_test, setup_seed=setup_seed, data_root=data_root,
                                num_first_datapoints_test=num_first_datapoints_test,
                                filter_fully_missing_test=filter_fully_missing_test, synthetic=synthetic,
                                synthetic_num_test=synthetic_num_test, rank=rank, world_size=world_size, **unused)
        self.model = None

    def set_model(self, model):
        self.model = model

    def setup(self, stage="test"):
        if stage != "test":
            raise NotImplementedError(f"{stage=} is not implemented.")
        hp = self.hparams
        rng = torch.Generator().manual_seed(int(hp.setup_seed))
        rng_dataset = torch.Generator().manual_seed(int(hp.setup_seed * 2.33 / 2))
        path = os.path.join(hp.data_root, hp.dataset, "test.npy")
        if os.path.exists(path):
            X = torch.from_numpy(np.load(path, allow_pickle=False).astype(np.float32))
        elif hp.dataset == "toy_grid2dmog_custtest0":
            X = toy_custom_test_points()  # defined in code by the reference too: no data file
        elif hp.synthetic:
            if hp.dataset not in DATASET_DIMS:
                raise NotImplementedError(f"unknown dataset {hp.dataset!r}")
            n = hp.synthetic_num_test or DATASET_TEST_SIZES[hp.dataset]
            if hp.dataset in BINARY_DATASETS:
                X = (torch.rand(n, DATASET_DIMS[hp.dataset], generator=rng_dataset) < 0.2).float()
            else:
                X = torch.randn(n, DATASET_DIMS[hp.dataset], generator=rng_dataset)
        else:
            raise FileNotFoundError(f"{path} not found (set data.synthetic: true for shape-faithful synthetic data)")
        Xf = torch.nan_to_num(X, nan=0.0)
        self.test_data_core = AttrDict(data=X, data_min=Xf.min(0)[0].numpy(), data_max=Xf.max(0)[0].numpy())
        kind = str(hp.missingness).split(".")[-1]
        if kind == "uniform":
            M = uniform_missingness_mask(tuple(X.shape), hp.total_miss_test, rng)
        elif kind == "nan_miss":
            # the data itself marks its missing entries with NaN (custom test points of the toy and
            # MNIST-GMM configurations); the synthetic stand-in hides the lower half of the features
            if hp.synthetic and not torch.isnan(X).any():
                X = X.clone()
                X[:, X.shape[1] // 2:] = float("nan")
            M = ~torch.isnan(X)
        elif kind == "quadrants":
            M = quadrant_missingness_mask(X.shape[0], hp.img_dims, hp.total_miss_test, rng)
        else:
            raise NotImplementedError(f"missingness {hp.missingness!r} (only 'uniform', 'quadrants' and 'nan_miss')")
        if hp.filter_fully_missing_test:
            keep = M.any(dim=1)
            X, M = X[keep], M[keep]
        # DatasetWithIndices wraps the filtered dataset (missing_data_module.py:436-440 of the reference):
        # the data index is the position among the kept rows
        I = torch.arange(X.shape[0], dtype=torch.int64)
        if hp.num_first_datapoints_test is not None:
            n = min(X.shape[0], int(hp.num_first_datapoints_test))
            X, M, I = X[:n], M[:n], I[:n]
        self.num_test_total = X.shape[0]
        lo, hi = shard_range(X.shape[0], hp.rank, hp.world_size)
        self.shard_offset = lo
        self.test_data = (X[lo:hi].contiguous(), M[lo:hi].contiguous(), I[lo:hi].contiguous())
        print("Test data size:", self.num_test_total, f"(this rank: {hi - lo})")

    def __len__(self):
        return math.ceil(self.test_data[0].shape[0] / self.hparams.batch_size)

    def test_dataloader(self):
        X, M, I = self.test_data
        bs = int(self.hparams.batch_size)
        for lo in range(0, X.shape[0], bs):
            yield X[lo:lo + bs], M[lo:lo + bs], I[lo:lo + bs]
