"""GPU: the one-wave-per-SIMD f16c8 convolution kernel (csrc/conv_f16c8_w4.hip: 4 waves, 4 x 4 blocks of 32 x 32 per
wave, 16-row tiles) against the 8-wave kernel it relieves (same operands, same products: only the fp32 summation
order differs) and against plain fp32 torch on the CPU.  EOVAE_C8_W4=1 routes every layer that fits the kernel's
tile shape to it, =0 none; since round 4 the switches are library options (eovae_set_option: read from the environment once,
then only changed through the ABI -- no os.environ mutation under a running launch)."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.fixture(autouse=True, params=['4', '2'])
def f16c8_mode(request):
    """Both tile shapes of the kernel: 4 rows per wave (16-row tiles, one block per CU) and 2 rows per wave (8-row
    tiles, two skewed blocks per CU)."""
    from eo_vae_amd import lib as L, ops
    lib = L.load()
    if request.param == '2' and not (lib.eovae_build_flags() & 1):
        pytest.skip('the two-rows-per-wave instantiations are compiled only into parked builds (EOVAE_PARKED = 0)')
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f16c8'
    keep = {k: lib.eovae_get_option(k) for k in (b'c8_w4', b'c8_w4_ni', b'c8_mx', b'c8_wgrad_corr')}
    L.check(lib.eovae_set_option(b'c8_w4_ni', int(request.param)), 'set_option')
    # the comparisons below are "same products, other summation order": both kernels on the fp8 form of the correction
    # product (the one-wave kernel's MX-FP6 form has its own tests, tests/test_gpu_mx.py)
    L.check(lib.eovae_set_option(b'c8_mx', 0), 'set_option')
    # ... and the weight gradient with its correction product: the plain-f16 form rounds the activations to f16, and an
    # activation that differs in its last fp32 bit between the two forward kernels can round the other way (2^-11 apart)
    L.check(lib.eovae_set_option(b'c8_wgrad_corr', 1), 'set_option')
    yield
    ops.CONV_PRECISION = old
    for k, v in keep.items():
        lib.eovae_set_option(k, v)


def run(fn, w4):
    from eo_vae_amd import lib as L
    L.check(L.load().eovae_set_option(b'c8_w4', 1 if w4 else 0), 'set_option')
    out = fn()
    torch.cuda.synchronize()
    return out


# (N, Cin, Cout, H, W): one chunk / several chunks; one co-block / several (XCD-aware map on and off); a grid that
# gives some CUs two tiles (320 tiles on 256 CUs: the tile-to-tile hand-over) and image borders on every side
SHAPES = [(1, 32, 128, 16, 32), (2, 64, 128, 32, 64), (3, 96, 256, 48, 32), (5, 32, 256, 128, 128),
          (2, 128, 384, 32, 96)]


@pytest.mark.parametrize('n,cin,cout,h,w', SHAPES)
def test_forward_and_data_gradient(n, cin, cout, h, w):
    from eo_vae_amd import ops
    g = torch.Generator().manual_seed(n * 1000 + cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, h, w, generator=g)
    gy = torch.randn(n, cout, h, w, generator=g) * 1e-4      # gradients sit far below f16's range: the amax path
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, wt, b, padding=1) + res
    yr.backward(gy)

    def step():
        xg = x.cuda().requires_grad_(True)
        y = ops.conv2d(xg, wt.cuda(), b.cuda(), res.cuda(), 'normal')
        y.backward(gy.cuda())
        return y.detach(), xg.grad.detach(), ops.gn_sums_of(y, 32)
    y1, gx1, s1 = run(step, True)
    y0, gx0, s0 = run(step, False)
    assert rel(y1, yr) < 5e-5 and rel(gx1, xr.grad) < 5e-5          # f16c8 vs fp32: ~1e-5 per product
    assert rel(y1, y0) < 2e-6 and rel(gx1, gx0) < 2e-6               # same products, other summation order
    if s1 is not None:
        yd = y1.double().view(n, 32, -1)
        want = torch.stack([yd.sum(-1), (yd * yd).sum(-1)], -1)
        assert (s1 - want).abs().max() <= 2e-6 * want.abs().max()
        assert (s1 - s0).abs().max() <= 2e-6 * want.abs().max()


@pytest.mark.parametrize('n,c,cout,h,w', [(2, 64, 128, 32, 64), (5, 128, 256, 64, 128)])
def test_record_input(n, c, cout, h, w):
    """GroupNorm + SiLU hands its result over as operand records: both kernels copy them to LDS unconverted."""
    from eo_vae_amd import ops
    torch.manual_seed(c + h)
    x = torch.randn(n, c, h, w, device='cuda') * 2 + 0.3
    gamma, beta = torch.randn(c, device='cuda'), torch.randn(c, device='cuda')
    wt = (torch.randn(cout, c, 3, 3, device='cuda') / (c * 9) ** 0.5)
    b = torch.randn(cout, device='cuda')

    def fwd():
        with torch.no_grad():
            a = ops.group_norm_silu(x, gamma, beta, packed=True)
            assert ops.is_packed(a)
            return ops.conv2d(a, wt, b, None, 'normal')
    y1, y0 = run(fwd, True), run(fwd, False)
    with torch.no_grad():
        z = F.silu(F.group_norm(x.cpu(), 32, gamma.cpu(), beta.cpu(), eps=1e-6))
        want = F.conv2d(z, wt.cpu(), b.cpu(), padding=1)
    assert rel(y1, want) < 5e-5
    assert rel(y1, y0) < 2e-6


def test_resnet_block_gradients_through_both_kernels():
    """A whole ResnetBlock forward + backward (records in, fused GroupNorm sums out, residual epilogue, scaled
    data gradient) gives the same parameter gradients through either kernel."""
    from eo_vae_amd.layers import ResnetBlock
    torch.manual_seed(3)
    blk = ResnetBlock(128, 128).cuda()
    x = torch.randn(4, 128, 64, 64, device='cuda')
    gy = torch.randn(4, 128, 64, 64, device='cuda') * 1e-3

    def step():
        for p in blk.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        y = blk(xg)
        y.backward(gy)
        return [y.detach(), xg.grad.detach()] + [p.grad.detach().clone() for p in blk.parameters()]
    a, b = run(step, True), run(step, False)
    for u, v in zip(a, b):
        assert rel(u, v) < 5e-6
