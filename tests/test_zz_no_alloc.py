"""GPU test of cs_final_resample_static, the entry point that allocates nothing: the CDF workspace is a fixed
array in the library's device memory.  It must reproduce cs_final_resample (indices bit for bit, gathered
latents exactly) on two calls in a row, the second with a smaller input (the array keeps the first call's
contents), and reject problems beyond its stated maximum."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(L, _lib, static, logw, Zp, u, n, rng=None):
    TI, B, Kp = logw.shape
    Zd = Zp.shape[-1]
    idx = torch.full((B, n), -1, dtype=torch.int64, device="cuda")
    Zout = torch.full((B, n, Zd), float("nan"), device="cuda")
    r = C.byref(rng) if rng is not None else None
    if static:
        _lib.check(L.cs_final_resample_static(_lib.ptr(logw), _lib.ptr(Zp), TI, B, Kp, Zd, _lib.ptr(u), r, n,
                                              _lib.ptr(idx), _lib.ptr(Zout), _lib.current_stream()))
    else:
        ws = torch.empty(B * TI * Kp, device="cuda")
        _lib.check(L.cs_final_resample(_lib.ptr(logw), _lib.ptr(Zp), TI, B, Kp, Zd, _lib.ptr(u), r, n, _lib.ptr(ws),
                                       _lib.ptr(idx), _lib.ptr(Zout), _lib.current_stream()))
    torch.cuda.synchronize()
    return idx, Zout


def test_static_workspace_final_resampling_matches_the_allocating_entry_point():
    from vae_conditional_sampling_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    for (TI, B, Kp, Zd, n) in ((37, 130, 5, 32, 148), (9, 21, 3, 7, 20)):   # second call: smaller input
        logw = (3.0 * torch.randn(TI, B, Kp, generator=g)).cuda()
        logw[0, 0, :] = float("-inf")                       # proposals of zero weight
        Zp = torch.randn(TI, B, Kp, Zd, generator=g).cuda()
        u = torch.rand(B, n, generator=g, dtype=torch.float64).cuda()
        i0, z0 = _run(L, _lib, False, logw, Zp, u, n)
        i1, z1 = _run(L, _lib, True, logw, Zp, u, n)
        assert int(i0.min()) >= 0 and int(i0.max()) < TI * Kp
        assert torch.equal(i0, i1), "indices differ from cs_final_resample"
        assert torch.equal(z0, z1), "gathered latents differ from cs_final_resample"
        # counter-RNG uniforms instead of injected ones
        rng = _lib.make_rng(77, 3, 1000)
        i0, z0 = _run(L, _lib, False, logw, Zp, None, n, rng)
        i1, z1 = _run(L, _lib, True, logw, Zp, None, n, rng)
        assert torch.equal(i0, i1) and torch.equal(z0, z1)


def test_static_workspace_call_leaves_the_free_device_memory_unchanged():
    """The property the entry point exists for: with inputs and outputs in place, a call takes no device memory
    (hipMemGetInfo's free byte count is the same before and after, on the second call as on later ones)."""
    from vae_conditional_sampling_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(12)
    TI, B, Kp, Zd, n = 11, 40, 5, 8, 30
    logw = torch.randn(TI, B, Kp, generator=g).cuda()
    Zp = torch.randn(TI, B, Kp, Zd, generator=g).cuda()
    u = torch.rand(B, n, generator=g, dtype=torch.float64).cuda()
    idx = torch.empty(B, n, dtype=torch.int64, device="cuda")
    Zout = torch.empty(B, n, Zd, device="cuda")
    args = (_lib.ptr(logw), _lib.ptr(Zp), TI, B, Kp, Zd, _lib.ptr(u), None, n, _lib.ptr(idx), _lib.ptr(Zout),
            _lib.current_stream())
    _lib.check(L.cs_final_resample_static(*args))   # the address look-up of the first call on the device
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        _lib.check(L.cs_final_resample_static(*args))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0


def test_static_workspace_rejects_sizes_beyond_its_maximum():
    from vae_conditional_sampling_amd import _lib
    L = _lib.lib()
    one = torch.zeros(8, device="cuda")
    idx = torch.zeros(8, dtype=torch.int64, device="cuda")
    # B * TI * K' = 8 388 609 > 8 388 608: rejected before anything is launched or read
    rc = L.cs_final_resample_static(_lib.ptr(one), _lib.ptr(one), 8388609, 1, 1, 1, None, None, 1, _lib.ptr(idx),
                                    _lib.ptr(one), _lib.current_stream())
    assert rc == -1 and b"static workspace" in L.cs_last_error()
    rc = L.cs_final_resample_static(_lib.ptr(one), _lib.ptr(one), 1 << 40, 1 << 40, 1, 1, None, None, 1, _lib.ptr(idx),
                                    _lib.ptr(one), _lib.current_stream())
    assert rc == -1
