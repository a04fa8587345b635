"""fp16-split (2-term) arithmetic of the 3x3 halo convolution kernel: per-image power-of-two operand scales.

The truth here is the same layer program under the exact-f32 matrix instructions; the reference-pinned goldens of
the convolutional networks run under all three arithmetics in test_gpu_conv.py.  These cases look at what is new in
the fp16 split: operands are scaled per IMAGE (from the per-image maxima the producing layer leaves, or a reduction
pass over the network input), so images of very different magnitudes in one launch, images of zeros, and the
composition of the batch must not matter.
"""
import pytest
import torch

from vae_conditional_sampling_amd import _lib

pytestmark = pytest.mark.gpu

RTOL = 1e-5   # of an image's largest output magnitude (float32-equivalent arithmetic, north-star tolerance)


@pytest.fixture(autouse=True, params=["small_blocks_for_few_images", "large_blocks_always", "passes_in_two_halves"])
def block_shapes(request):
    """Launches of a few images run in blocks of 32 output channels (so that they reach more CUs); every case here
    runs with that rule and with the large blocks that full-size launches use.  Passes with enough work run as two
    halves on two streams: the third setting divides every pass of 8 rows and more that way."""
    prev = _lib.lib().cs_debug_set_small_launch_steps(18 if request.param.startswith("small") else 0)
    prev2 = _lib.lib().cs_debug_set_prog_two_streams(-8 if request.param == "passes_in_two_halves" else 0)
    yield
    _lib.lib().cs_debug_set_small_launch_steps(prev)
    _lib.lib().cs_debug_set_prog_two_streams(prev2)


def _program(c_in, c_out, hw, n_layers, seed, upsample_last=False, residual=False):
    g = torch.Generator().manual_seed(seed)
    out_hw = hw * (2 if upsample_last else 1)
    b = _lib.ProgramBuilder(c_in * hw * hw, c_out * out_hw * out_hw, 1)
    src, c = 0, c_in
    for i in range(n_layers):
        last = i == n_layers - 1
        dst = _lib.CS_NET_BUF_OUT if last else 1 + (i & 1)
        w = torch.randn(c_out, c, 3, 3, generator=g) / (3 * c ** 0.5)
        bias = torch.randn(c_out, generator=g) * 0.1
        res = src if (residual and i > 0 and not last and c == c_out) else -1
        b.conv(src, dst, w, bias, hin=hw, win=hw, relu=not last, upsample=2 if (last and upsample_last) else 1, res=res)
        src, c = dst, c_out
    return b.build()


def _forward(net, x, arith):
    net.set_arith(arith)
    try:
        return net.forward(x)
    finally:
        net.set_arith(None)


def _per_image_err(y, ref):
    """max |y - ref| per image over the image's largest |ref| (an image of zeros: absolute)."""
    d = (y - ref).abs().amax(dim=1)
    m = ref.abs().amax(dim=1).clamp_min(1e-30)
    return (d / m).max().item()


# (64 / 32 channels at 40 x 40: wider than the 4-wave block shapes stage, so the 8-wave ones run)
@pytest.mark.parametrize("c,hw,rows", [(64, 14, 37), (128, 7, 70), (256, 4, 130), (64, 28, 9), (32, 14, 33), (64, 40, 3),
                                       (32, 40, 3)])
def test_images_of_very_different_magnitudes_share_a_launch(c, hw, rows):
    net = _program(c, c, hw, 3, seed=c + hw)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(rows, c * hw * hw, generator=g)
    scale = 10.0 ** (torch.rand(rows, 1, generator=g) * 8 - 4)   # 1e-4 .. 1e4 per image
    # layers carry biases of order 0.1, so small images are dominated by them: scale only to probe the operand scales
    x = (x * scale).cuda()
    ref = _forward(net, x, "f32")
    y = _forward(net, x, "f16x2")
    y3 = _forward(net, x, "bf16x3")
    assert _per_image_err(y, ref) < RTOL
    assert _per_image_err(y3, ref) < RTOL
    assert not torch.equal(y, y3), "the fp16-split halo kernel did not run (results identical to the bf16 split)"


def test_upsampled_and_residual_layers():
    g = torch.Generator().manual_seed(5)
    net = _program(64, 64, 7, 3, seed=3, upsample_last=True, residual=True)
    x = (torch.randn(50, 64 * 49, generator=g) * 10.0 ** (torch.rand(50, 1, generator=g) * 4 - 2)).cuda()
    ref = _forward(net, x, "f32")
    y = _forward(net, x, "f16x2")
    assert _per_image_err(y, ref) < RTOL
    assert not torch.equal(y, _forward(net, x, "bf16x3"))


def test_images_of_zeros_and_single_spikes():
    c, hw = 64, 14
    net = _program(c, c, hw, 2, seed=11)
    x = torch.zeros(6, c * hw * hw)
    x[1, 5] = 3.0e4            # one huge entry in an otherwise empty image
    x[2, :] = 1.0e-30          # tiny everywhere
    x[3] = torch.randn(c * hw * hw, generator=torch.Generator().manual_seed(2))
    x[4, 100] = -7.0e-12
    x = x.cuda()
    ref = _forward(net, x, "f32")
    y = _forward(net, x, "f16x2")
    assert torch.isfinite(y).all()
    assert _per_image_err(y, ref) < RTOL
    # through a single layer an image of zeros comes out as the biases, exactly
    one = _program(c, c, hw, 1, seed=12)
    assert torch.equal(_forward(one, x, "f16x2")[0], _forward(one, x, "f32")[0])


def test_an_image_does_not_depend_on_the_rest_of_the_batch():
    c, hw = 128, 7
    net = _program(c, c, hw, 3, seed=21)
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(40, c * hw * hw, generator=g) * 10.0 ** (torch.rand(40, 1, generator=g) * 6 - 3)).cuda()
    y = _forward(net, x, "f16x2")
    perm = torch.randperm(40, generator=g).cuda()
    yp = _forward(net, x[perm], "f16x2")
    assert torch.equal(yp, y[perm])
    ys = _forward(net, x[7:19], "f16x2")
    assert torch.equal(ys, y[7:19])


def test_a_pass_in_two_halves_gives_the_bits_of_one_pass():
    """Rows are independent: a pass divided over two streams must return exactly what the undivided pass returns
    (odd row counts, a residual layer program, all three arithmetics)."""
    net = _program(64, 64, 14, 4, seed=31, upsample_last=True, residual=True)
    g = torch.Generator().manual_seed(8)
    for rows in (9, 37, 130):
        x = torch.randn(rows, 64 * 14 * 14, generator=g).cuda()
        for arith in ("f16x2", "bf16x3", "f32"):
            L = _lib.lib()
            prev = L.cs_debug_set_prog_two_streams(0)
            try:
                one = _forward(net, x, arith)
                L.cs_debug_set_prog_two_streams(-8)
                two = _forward(net, x, arith)
                again = _forward(net, x, arith)
            finally:
                L.cs_debug_set_prog_two_streams(prev)
            assert torch.equal(one, two) and torch.equal(two, again), (rows, arith)
