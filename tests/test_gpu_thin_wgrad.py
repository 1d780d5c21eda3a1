"""GPU: the fp32 weight-gradient kernel of the wavelength-conditioned first / last 3x3 layers (csrc/conv_thin.hip: thin
side <= 4 channels) against float64 torch, directly through the C-ABI and through the autograd function."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def hashed(shape, salt, scale=1.0):
    from oracle.eo_vae_cpu import _hash_uniform
    n = 1
    for s in shape:
        n *= s
    return (_hash_uniform(n, salt).reshape(shape) * (scale * 12 ** 0.5)).float()


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


# n, cin, cout, h, w   (first layer: few input channels; last layer: few output channels)
CASES = [(2, 3, 128, 64, 64), (2, 128, 3, 64, 64), (1, 2, 128, 40, 48), (3, 64, 4, 33, 100), (1, 1, 32, 70, 256),
         (2, 128, 3, 256, 256), (1, 4, 256, 37, 5)]


@pytest.mark.parametrize('n,cin,cout,h,w', CASES)
def test_thin_weight_gradient_against_float64(n, cin, cout, h, w):
    from eo_vae_amd import lib as L
    lib = L.load()
    assert lib.eovae_conv3x3_thin_wgrad_eligible(n, cin, h, w, cout, 0) == 1
    x = hashed((n, cin, h, w), 61, 1.0)
    dy = hashed((n, cout, h, w), 62, 1.0)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, torch.zeros(cout, dtype=torch.float64), padding=1).backward(dy.double())
    xg, dyg = x.cuda(), dy.cuda()
    dw = torch.full((cout, cin, 3, 3), float('nan'), device='cuda')
    db = torch.full((cout,), float('nan'), device='cuda')
    ws = torch.empty(lib.eovae_conv3x3_thin_wgrad_workspace(n, cin, h, w, cout), dtype=torch.uint8, device='cuda')
    outs = []
    for _ in range(2):
        L.check(lib.eovae_conv3x3_thin_wgrad(L.ptr(xg), L.ptr(dyg), L.ptr(dw), L.ptr(db), n, cin, h, w, cout, 0, L.ptr(ws),
                                             L.stream()), 'thin')
        torch.cuda.synchronize()
        outs.append(dw.clone())
    assert rel(dw, wt.grad) < 1e-5
    assert torch.equal(outs[0], outs[1])                   # fixed summation order
    if cout > cin:
        assert rel(db, dy.double().sum((0, 2, 3))) < 1e-5
    else:
        assert bool(torch.isnan(db).all())                 # not produced for the few-output-channel layer


def test_thin_weight_gradient_reads_operand_records():
    """The last layer's input is GroupNorm + SiLU output handed over as operand records: the kernel rebuilds
    x = f16 + e5m2(lo') * 2^-12 from them (what the matrix-core kernels multiply with)."""
    from eo_vae_amd import lib as L, ops
    lib = L.load()
    n, cin, cout, h, w = 2, 128, 3, 64, 96
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f16c8'
    try:
        x0 = hashed((n, cin, h, w), 71, 2.0).cuda()
        g = (1.0 + hashed((cin,), 72, 0.2)).cuda()
        b = hashed((cin,), 73, 0.2).cuda()
        rec = ops.group_norm_silu(x0, g, b, 32, 1e-6, True, True).detach()
        y = ops.group_norm_silu(x0, g, b, 32, 1e-6, True, False).detach()
        assert ops.is_packed(rec) and not ops.is_packed(y)
    finally:
        ops.CONV_PRECISION = old
    dy = hashed((n, cout, h, w), 74, 1.0).cuda()
    assert lib.eovae_conv3x3_thin_wgrad_eligible(n, cin, h, w, cout, 1) == 1
    assert lib.eovae_conv3x3_thin_wgrad_eligible(n, cout, h, w, cin, 1) == 0     # records are never the thin tensor
    ws = torch.empty(lib.eovae_conv3x3_thin_wgrad_workspace(n, cin, h, w, cout), dtype=torch.uint8, device='cuda')
    dw_r, dw_f = torch.empty(cout, cin, 3, 3, device='cuda'), torch.empty(cout, cin, 3, 3, device='cuda')
    L.check(lib.eovae_conv3x3_thin_wgrad(L.ptr(rec), L.ptr(dy), L.ptr(dw_r), None, n, cin, h, w, cout, 1, L.ptr(ws),
                                         L.stream()), 'thin records')
    L.check(lib.eovae_conv3x3_thin_wgrad(L.ptr(y), L.ptr(dy), L.ptr(dw_f), None, n, cin, h, w, cout, 0, L.ptr(ws),
                                         L.stream()), 'thin fp32')
    torch.cuda.synchronize()
    assert rel(dw_r, dw_f) < 2e-5          # records carry f16 + a 2-3 bit correction: ~2^-14 per element, averaged out


@pytest.mark.parametrize('mode', ['bf16x3', 'f16c8'])
def test_dynamic_conv_shapes_take_the_thin_kernel_through_autograd(mode):
    from eo_vae_amd import ops
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, mode
    try:
        for cin, cout in ((3, 128), (128, 3)):
            x = hashed((2, cin, 32, 64), 81, 1.0)
            wt = hashed((cout, cin, 3, 3), 82, 1.0 / (cin * 9) ** 0.5)
            b = hashed((cout,), 83, 0.1)
            gy = hashed((2, cout, 32, 64), 84, 1.0)
            wr, br = wt.clone().double().requires_grad_(True), b.clone().double().requires_grad_(True)
            F.conv2d(x.double(), wr, br, padding=1).backward(gy.double())
            xg, wg, bg = x.cuda().requires_grad_(True), wt.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
            ops.conv2d(xg, wg, bg).backward(gy.cuda())
            torch.cuda.synchronize()
            assert rel(wg.grad, wr.grad) < 1e-5       # fp32 FMAs, whatever the arithmetic mode of the other layers
            assert rel(bg.grad, br.grad) < 1e-5
    finally:
        ops.CONV_PRECISION = old


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 128, 3, 64, 64), (1, 64, 4, 33, 100), (3, 32, 1, 17, 70), (2, 128, 2, 256, 256)])
def test_thin_forward_against_float64(n, cin, cout, h, w):
    from eo_vae_amd import lib as L
    lib = L.load()
    assert lib.eovae_conv3x3_thin_fwd_eligible(n, cin, h, w, cout) == 1
    x = hashed((n, cin, h, w), 91, 1.0)
    wt = hashed((cout, cin, 3, 3), 92, 1.0 / (cin * 9) ** 0.5)
    b = hashed((cout,), 93, 0.1)
    want = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    xg, wg, bg = x.cuda(), wt.cuda(), b.cuda()
    y = torch.full((n, cout, h, w), float('nan'), device='cuda')
    L.check(lib.eovae_conv3x3_thin_fwd(L.ptr(xg), L.ptr(wg), L.ptr(bg), L.ptr(y), n, cin, h, w, cout, 0, L.stream()), 'thin fwd')
    torch.cuda.synchronize()
    assert rel(y, want) < 2e-6
    L.check(lib.eovae_conv3x3_thin_fwd(L.ptr(xg), L.ptr(wg), None, L.ptr(y), n, cin, h, w, cout, 0, L.stream()), 'thin fwd')
    torch.cuda.synchronize()
    assert rel(y, want - b.double().view(1, -1, 1, 1)) < 2e-6


def test_thin_forward_reads_operand_records():
    from eo_vae_amd import lib as L, ops
    lib = L.load()
    n, cin, cout, h, w = 2, 128, 3, 48, 80
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f16c8'
    try:
        x0 = hashed((n, cin, h, w), 95, 2.0).cuda()
        g = (1.0 + hashed((cin,), 96, 0.2)).cuda()
        b0 = hashed((cin,), 97, 0.2).cuda()
        rec = ops.group_norm_silu(x0, g, b0, 32, 1e-6, True, True).detach()
        yv = ops.group_norm_silu(x0, g, b0, 32, 1e-6, True, False).detach()
        wt = hashed((cout, cin, 3, 3), 98, 1.0 / (cin * 9) ** 0.5).cuda()
        b = hashed((cout,), 99, 0.1).cuda()
        want = F.conv2d(yv.double().cpu(), wt.double().cpu(), b.double().cpu(), padding=1)
        got = ops.conv2d(rec, wt, b)                      # the autograd function routes the record input to the thin kernel
        torch.cuda.synchronize()
        assert not ops.is_packed(got)
        assert rel(got, want) < 2e-5                      # records carry ~2^-14 per element
    finally:
        ops.CONV_PRECISION = old
