"""cs_row_order_statistics (radix selection) against torch.sort, and the epoch-end summary built on it."""
import ctypes as C

import numpy as np
import pytest
import torch

from vae_conditional_sampling_amd import _lib

pytestmark = pytest.mark.gpu


def _select(x, ranks):
    rows, m = x.shape
    L = _lib.lib()
    nq = len(ranks)
    ws = torch.empty(L.cs_row_order_statistics_workspace(rows, nq), dtype=torch.uint8, device=x.device)
    out = torch.empty(rows, nq, device=x.device, dtype=torch.float32)
    r = (C.c_int64 * nq)(*ranks)
    _lib.check(L.cs_row_order_statistics(_lib.ptr(x), rows, m, r, nq, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
    return out


@pytest.mark.parametrize("rows,m", [(1, 1), (3, 2), (7, 1000), (40, 65537), (2, 1 << 21), (600, 4097), (1500, 300)])
def test_order_statistics_match_a_sort(rows, m):
    g = torch.Generator().manual_seed(rows * 7 + m)
    x = torch.randn(rows, m, generator=g)
    x[0, : m // 2] = 0.25                      # many equal values
    if m > 8:
        x[-1, :4] = torch.tensor([float("inf"), -float("inf"), 0.0, -0.0])
        x[-1, 5:8] = torch.tensor([1e-42, -1e-42, 3e38])       # denormals, large magnitudes
    if rows > 2:
        x[1] = torch.randint(0, 3, (m,), generator=g).float()  # three distinct values
    x = x.cuda().contiguous()
    ranks = sorted(set([0, m - 1, m // 2, (m - 1) // 2, m // 10, (9 * m) // 10, min(m - 1, 1), max(0, m - 2), m // 4, (3 * m) // 4]))
    out = _select(x, ranks)
    ref = x.sort(dim=1).values[:, ranks]
    assert torch.equal(out, ref)               # (-0.0 == 0.0 as floats)
    # a rank repeated, and ranks in any order
    r2 = [ranks[-1], ranks[0], ranks[-1]]
    assert torch.equal(_select(x, r2), x.sort(dim=1).values[:, r2])


def test_too_many_ranks_and_bad_ranks_are_refused():
    x = torch.randn(2, 10).cuda()
    L = _lib.lib()
    assert L.cs_row_order_statistics_workspace(2, 17) == -1 and L.cs_row_order_statistics_workspace(2, 16) > 0
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    out = torch.empty(2, 17, device="cuda")
    r17 = (C.c_int64 * 17)(*range(17))
    assert L.cs_row_order_statistics(_lib.ptr(x), 2, 10, r17, 17, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.current_stream()) != 0
    with pytest.raises(_lib.CsError):
        _select(x, [10])


def test_pointwise_summary_on_the_device_matches_numpy():
    """The epoch-end aggregation (test_step_vae_sampling.py:824-858) through the selection kernel against numpy's
    quantiles, including an iteration that holds a NaN."""
    from vae_conditional_sampling_amd.sampling.imputation_metrics import pointwise_summary
    rs = np.random.RandomState(2)
    values = [rs.rand(9, n, 4).astype(np.float32) for n in (50, 31, 7)]
    values[1][3, 5, 2] = np.nan
    qs = [0.1, 0.25, 0.5, 0.75, 0.9]
    mean, std, quant = pointwise_summary([torch.from_numpy(v).cuda() for v in values], qs, "cuda")
    allv = np.concatenate(values, axis=1)                      # [t, n, k]
    ref_mean = allv.mean(axis=1).mean(axis=1)
    ref_q = np.stack([np.quantile(allv.reshape(9, -1), q, axis=1) for q in qs])
    ok = ~np.isnan(ref_mean)
    assert np.allclose(mean[ok], ref_mean[ok], rtol=1e-5)
    assert np.array_equal(np.isnan(quant), np.isnan(ref_q)) and np.isnan(quant[:, 3]).all()
    assert np.allclose(quant, ref_q, rtol=1e-6, atol=1e-7, equal_nan=True)
