"""GPU: the one-wave-per-SIMD f16c8 weight-gradient kernel (csrc/conv_f16c8_wgrad_w4.hip) against the 8-wave kernel
(same operands and products, other summation order and atomic order) and plain fp32 torch on the CPU.
The kernel is PARKED (slower than the 8-wave kernels): it only serves the atomic-accumulator path
(ops.WGRAD_DIRECT = False) with the library option c8_wgrad_w4 = 1, which is how this file reaches it."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope='module')
def needs_parked_build():
    """The kernel is compiled only into parked builds (EOVAE_BUILD_PARKED=1 python -m eo_vae_amd.build --force; csrc/eovae_common.h)."""
    from eo_vae_amd import lib as L
    if not (L.load().eovae_build_flags() & 1):
        pytest.skip('conv_f16c8_wgrad_w4.hip is not compiled into this build (EOVAE_PARKED = 0)')


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.fixture(autouse=True)
def f16c8_mode():
    from eo_vae_amd import lib as L, ops
    lib = L.load()
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f16c8'
    direct, ops.WGRAD_DIRECT = ops.WGRAD_DIRECT, False       # the atomic-accumulator path: the only one the parked kernel serves
    keep = lib.eovae_get_option(b'c8_wgrad_w4')
    yield
    ops.CONV_PRECISION, ops.WGRAD_DIRECT = old, direct
    lib.eovae_set_option(b'c8_wgrad_w4', keep)


def run(fn, w4):
    from eo_vae_amd import lib as L
    L.check(L.load().eovae_set_option(b'c8_wgrad_w4', 1 if w4 else 0), 'set_option')
    out = fn()
    torch.cuda.synchronize()
    return out


# (N, Cin, Cout, H, W): one slab / several; tiles on every image border; several tiles per block
@pytest.mark.parametrize('n,cin,cout,h,w', [(1, 64, 128, 2, 32), (2, 64, 128, 16, 64), (3, 128, 256, 24, 32),
                                             (2, 192, 128, 64, 96), (4, 64, 384, 128, 128)])
def test_weight_and_bias_gradient(n, cin, cout, h, w):
    from eo_vae_amd import ops
    g = torch.Generator().manual_seed(n * 100 + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).requires_grad_(True)
    b = torch.randn(cout, generator=g).requires_grad_(True)
    gy = torch.randn(n, cout, h, w, generator=g) * 1e-4      # gradients sit far below f16's range: the amax path
    F.conv2d(x, wt, b, padding=1).backward(gy)

    def step():
        wg, bg = wt.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
        ops.conv2d(x.cuda(), wg, bg, None, 'normal').backward(gy.cuda())
        return wg.grad.detach(), bg.grad.detach()
    w1, b1 = run(step, True)
    w0, b0 = run(step, False)
    assert rel(w1, wt.grad) < 1e-4 and rel(b1, b.grad) < 2e-5
    assert rel(w1, w0) < 5e-6 and rel(b1, b0) < 5e-6


@pytest.mark.parametrize('n,c,cout,h,w', [(2, 64, 128, 32, 64), (3, 128, 256, 64, 64)])
def test_record_input(n, c, cout, h, w):
    """x arrives as GroupNorm operand records (the model's hand-over): both kernels copy them to LDS unconverted."""
    from eo_vae_amd import ops
    torch.manual_seed(c + h)
    x = torch.randn(n, c, h, w, device='cuda') * 2 + 0.3
    gamma, beta = torch.randn(c, device='cuda'), torch.randn(c, device='cuda')
    wt = (torch.randn(cout, c, 3, 3, device='cuda') / (c * 9) ** 0.5)
    b = torch.randn(cout, device='cuda')
    gy = torch.randn(n, cout, h, w, device='cuda') * 1e-3

    def step():
        wg, bg = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
        a = ops.group_norm_silu(x, gamma, beta, packed=True)
        ops.conv2d(a, wg, bg, None, 'normal').backward(gy)
        return wg.grad.detach(), bg.grad.detach()
    w1, b1 = run(step, True)
    w0, b0 = run(step, False)
    z = F.silu(F.group_norm(x.cpu(), 32, gamma.cpu(), beta.cpu(), eps=1e-6))
    wr = wt.cpu().clone().requires_grad_(True)
    F.conv2d(z, wr, b.cpu(), padding=1).backward(gy.cpu())
    assert rel(w1, wr.grad) < 1e-4
    assert rel(w1, w0) < 5e-6 and rel(b1, b0) < 5e-6
