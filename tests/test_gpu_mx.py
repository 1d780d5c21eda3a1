"""GPU: the MX-FP6 form of the f16c8 correction product (round 5; csrc/conv_f16c8_common.h).

The operands are checked at three levels: (1) the bytes -- the planes of the GroupNorm's MX operand records and the weight
packs' FP6 part decode to exactly what oracle/mx_emul.py says they hold; (2) the product -- a convolution on those operands equals the
float64 sum of the SAME three products (f16 x f16 + the two corrections on the block-scaled e2m3 images) to summation-order
accuracy; (3) the scheme -- it stays within the f16c8 error of plain fp32, and the fp8 form (library option c8_mx = 0) is
still there behind the same call."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.fixture(autouse=True)
def f16c8_mx():
    from eo_vae_amd import lib as L, ops
    lib = L.load()
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f16c8'
    keep = {k: lib.eovae_get_option(k) for k in (b'c8_w4', b'c8_mx', b'c8_wgrad_corr', b'c8_q16')}
    L.check(lib.eovae_set_option(b'c8_mx', 1), 'set_option')
    L.check(lib.eovae_set_option(b'c8_wgrad_corr', 0), 'set_option')
    yield
    ops.CONV_PRECISION = old
    for k, v in keep.items():
        lib.eovae_set_option(k, v)


@pytest.fixture(params=[2, 0], ids=['q16', 'q32'])
def mfma_shape(request):
    """The one-wave kernel's two forms (library option c8_q16): 2 = conv_f16c8_q16.hip (16 x 16 MFMAs, correction product over
    tap pairs; MX records by DMA, fp32 inputs quantised in the kernel; the default), 0 = conv_f16c8_w4.hip (32 x 32 MFMAs)."""
    from eo_vae_amd import lib as L
    L.check(L.load().eovae_set_option(b'c8_q16', request.param), 'set_option')
    return request.param


def set_opt(name, v):
    from eo_vae_amd import lib as L
    L.check(L.load().eovae_set_option(name, v), 'set_option')


def plane_values(rec):
    """MX operand records -> (f16 values, lo, hi) as float64 [n, c, hw]."""
    from eo_vae_amd import ops
    from oracle import mx_emul as M
    n, c, h, w = rec.shape
    fpl, spl = ops.mx_planes(rec)
    f = fpl.cpu().numpy().view(np.float16).reshape(n, c // 8, h * w, 8).astype(np.float64)
    out = [torch.from_numpy(f).permute(0, 1, 3, 2).reshape(n, c, h * w)]
    a = spl.cpu().numpy()                                                                 # [n, c/32, 4, hw, 4]
    for part in range(2):
        pieces = np.concatenate([a[:, :, 2 * part], a[:, :, 2 * part + 1]], axis=-1)      # [n, c/32, hw, 8]
        v = M.decode_blocks(pieces)                                                       # [n, c/32, hw, 32]
        out.append(torch.from_numpy(v).permute(0, 1, 3, 2).reshape(n, c, h * w))
    return out


@pytest.mark.parametrize('n,c,h,w,spread', [(2, 64, 16, 32, 0), (1, 128, 8, 64, 12)])
def test_records_hold_the_emulated_operands(n, c, h, w, spread):
    """spread: per-channel magnitudes over 2^-spread .. 1 -- the block scales follow each pixel's 32 channels."""
    from eo_vae_amd import ops
    from oracle import mx_emul as M
    torch.manual_seed(c + h)
    x = torch.randn(n, c, h, w, device='cuda') * 2 + 0.3
    gamma = torch.randn(c, device='cuda')
    if spread:
        gamma = gamma * torch.exp2(-torch.randint(0, spread, (c,), device='cuda').float())
    beta = torch.randn(c, device='cuda') * 0.1
    with torch.no_grad():
        a = ops.group_norm_silu(x, gamma, beta, packed=True, cout=128)
        a1 = ops.group_norm_silu(x, gamma, beta, packed=True)            # consumer unknown: format 1
        z = ops.group_norm_silu(x, gamma, beta, packed=False).cpu()
    assert ops.packed_format(a) == 2 and ops.packed_format(a1) == 1 and ops.packed_format(a.detach().clone()) == 2
    with pytest.raises(ops.L.EovaeError):
        a + 1                                                            # the bytes are not fp32 values
    f, lo, hi = plane_values(a)
    zf = M.f16(z)
    # (the two GroupNorm kernels may round exp() apart by an ulp: a handful of elements may sit one code apart)
    assert (f != zf.double().reshape(n, c, h * w)).float().mean().item() < 2e-3
    zl = z - zf
    for got, src in ((hi, z), (lo, zl)):
        want = M.mx6_quant(src, 1).reshape(n, c, h * w)
        assert (got != want).float().mean().item() < 2e-3
        blockmax = src.abs().reshape(n, c // 32, 32, h * w).amax(2, keepdim=True).expand(-1, -1, 32, -1).reshape(n, c, h * w)
        assert ((got - want).abs() <= 0.1251 * blockmax.double() + 1e-300).all()    # never more than one code step of the block


@pytest.mark.parametrize('cout,cin', [(128, 64), (160, 96)])
def test_weight_pack_holds_the_emulated_operands(cout, cin):
    from eo_vae_amd import lib as L
    from oracle import mx_emul as M
    lib = L.load()
    torch.manual_seed(cout)
    w = (torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5) * torch.exp2(torch.randint(-6, 2, (cout, 1, 1, 1)).float())
    wd = w.cuda()
    for flip in (0, 1):
        rows, ks = (cin, cout) if flip else (cout, cin)
        nchunk, ncob = (ks + 31) // 32, (rows + 127) // 128
        wp = torch.zeros(lib.eovae_conv3x3_f16c8_packed_bytes(cout, cin, flip), dtype=torch.uint8, device='cuda')
        L.check(lib.eovae_conv3x3_f16c8_pack(L.ptr(wd), L.ptr(wp), cout, cin, flip, 1.0, L.stream()), 'pack')
        tiles = wp[:9 * nchunk * ncob * 24576].cpu().numpy().reshape(9, nchunk, ncob, 24576)
        fp6 = tiles[..., 16384:].copy().view(np.int32).reshape(9, nchunk, ncob, 2, 2, 128, 4)   # [part][piece][m][4]
        # K axis: ci (forward) / co (data gradient, taps reversed); rows m: co / ci
        wk = w if not flip else w.flip(2, 3).transpose(0, 1)       # [m][k][3][3]
        wl = wk - M.f16(wk)
        for part, src in ((0, wk), (1, wl)):
            want = M.mx6_quant(src, 1)                              # blocks of 32 k per (m, tap)
            pieces = np.concatenate([fp6[:, :, :, part, 0], fp6[:, :, :, part, 1]], axis=-1)    # [9, nchunk, ncob, 128, 8]
            got = M.decode_blocks(pieces)                                                       # [9, nchunk, ncob, 128, 32]
            got = torch.from_numpy(got).permute(2, 3, 1, 4, 0).reshape(ncob * 128, nchunk * 32, 9)[:rows, :ks]
            assert torch.equal(got, want.reshape(rows, ks, 9)), (flip, part)
            # padding rows / channels are zero blocks
            pad = torch.from_numpy(M.decode_blocks(pieces)).permute(2, 3, 1, 4, 0).reshape(ncob * 128, nchunk * 32, 9)[rows:]
            assert pad.numel() == 0 or pad.abs().max() == 0


@pytest.mark.parametrize('n,c,cout,h,w', [(2, 64, 128, 32, 64), (5, 128, 256, 64, 128), (1, 192, 128, 16, 32), (3, 64, 128, 48, 96)])
def test_forward_on_records_equals_the_emulated_product(n, c, cout, h, w, mfma_shape):
    from eo_vae_amd import ops
    from oracle import mx_emul as M
    torch.manual_seed(c + h)
    x = torch.randn(n, c, h, w, device='cuda') * 2 + 0.3
    gamma, beta = torch.randn(c, device='cuda'), torch.randn(c, device='cuda')
    wt = torch.randn(cout, c, 3, 3, device='cuda') / (c * 9) ** 0.5
    b = torch.randn(cout, device='cuda')
    res = torch.randn(n, cout, h, w, device='cuda')
    set_opt(b'c8_w4', 1)

    def fwd(mx):
        set_opt(b'c8_mx', mx)
        with torch.no_grad():
            a = ops.group_norm_silu(x, gamma, beta, packed=True, cout=cout)
            assert ops.packed_format(a) == (2 if mx else 1)
            y = ops.conv2d(a, wt, b, res, 'normal')
        torch.cuda.synchronize()
        return y
    y_mx, y_f8 = fwd(1), fwd(0)
    with torch.no_grad():
        z = ops.group_norm_silu(x, gamma, beta, packed=False).cpu()
    conv = lambda a_, b_: F.conv2d(a_, b_, padding=1)
    want = M.split_product(z, wt.cpu(), conv) + b.cpu().double().view(1, -1, 1, 1) + res.cpu().double()
    exact = conv(z.double(), wt.cpu().double()) + b.cpu().double().view(1, -1, 1, 1) + res.cpu().double()
    assert rel(y_mx, want) < 3e-6          # the same three products: fp32 summation order (+ rare one-code differences of z)
    assert rel(y_mx, exact) < 5e-5         # the scheme against fp64
    assert rel(y_f8, exact) < 5e-5 and rel(y_mx, y_f8) < 5e-5
    assert rel(y_mx, y_f8) > 1e-8          # and the two forms really are different arithmetic


@pytest.mark.parametrize('n,c,cout,h,w', [(2, 128, 64, 32, 64), (3, 128, 256, 48, 96), (16, 256, 128, 32, 32)])
def test_data_gradient_equals_the_emulated_product(n, c, cout, h, w, mfma_shape):
    """dX of a plain 3x3 layer.  Both one-wave kernels quantise dY themselves -- f16(dY 2^k) and the block-scaled e2m3 images of
    dY 2^k and of its f16 residual, blocks of 32 output channels per pixel -- and multiply with the data-gradient pack's parts:
    the float64 sum of the same three products.  (conv_f16c8_w4.hip with c8_dgrad_mx = 0: fp8 correction product, checked
    against fp64 only.)"""
    import math
    from eo_vae_amd import ops
    from oracle import mx_emul as M
    torch.manual_seed(c + cout)
    x = torch.randn(n, c, h, w, device='cuda')
    wt = torch.randn(cout, c, 3, 3, device='cuda') / (c * 9) ** 0.5
    dy = torch.randn(n, cout, h, w, device='cuda') * 3e-4 * torch.exp2(torch.randint(-3, 3, (1, cout, 1, 1), device='cuda').float())
    set_opt(b'c8_w4', 1)
    xg = x.clone().requires_grad_(True)
    ops.conv2d(xg, wt, None, None, 'normal').backward(dy)
    torch.cuda.synchronize()
    dx = xg.grad
    dyc, wc = dy.cpu(), wt.cpu()
    exact = F.conv_transpose2d(dyc.double(), wc.double(), padding=1)
    assert rel(dx, exact) < 5e-5
    s = 2.0 ** (13 - math.floor(math.log2(float(dyc.abs().max()))))       # the kernel's range pre-scale
    want = M.split_product(dyc * s, wc, lambda a_, b_: F.conv_transpose2d(a_, b_, padding=1), 1, 0) / s
    assert rel(dx, want) < 3e-6
    if not mfma_shape:      # ... and the fp8 form is still there behind the same call
        set_opt(b'c8_dgrad_mx', 0)
        try:
            xg2 = x.clone().requires_grad_(True)
            ops.conv2d(xg2, wt, None, None, 'normal').backward(dy)
            torch.cuda.synchronize()
            assert rel(xg2.grad, exact) < 5e-5 and 1e-8 < rel(xg2.grad, dx) < 5e-5
        finally:
            set_opt(b'c8_dgrad_mx', 1)


def test_resnet_block_with_and_without_mx():
    """Forward + backward of a ResnetBlock with MX operand records (MX-FP6 forward, f16 weight gradient) and with format-1
    records (fp8 correction product everywhere): the two arithmetic forms stay within the scheme's error of each other; MX
    records survive clone() as what they are."""
    from eo_vae_amd import ops
    from eo_vae_amd.layers import ResnetBlock
    torch.manual_seed(3)
    blk = ResnetBlock(128, 128).cuda()
    x = torch.randn(4, 128, 64, 64, device='cuda')
    gy = torch.randn(4, 128, 64, 64, device='cuda') * 1e-3

    def step(mx):
        set_opt(b'c8_mx', mx)
        set_opt(b'c8_wgrad_corr', 0 if mx else 1)
        for p in blk.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        y = blk(xg)
        y.backward(gy)
        torch.cuda.synchronize()
        return [y.detach(), xg.grad.detach()] + [p.grad.detach().clone() for p in blk.parameters()]
    a, b = step(1), step(0)
    for u, v in zip(a, b):
        assert rel(u, v) < 5e-4
    set_opt(b'c8_mx', 1)
    set_opt(b'c8_wgrad_corr', 0)
    with torch.no_grad():
        rec = ops.group_norm_silu(x, blk.norm1.weight, blk.norm1.bias, packed=True, cout=128)
        y1 = ops.conv2d(rec, blk.conv1.weight, blk.conv1.bias, None, 'normal')
        y2 = ops.conv2d(rec.clone(), blk.conv1.weight, blk.conv1.bias, None, 'normal')
    assert ops.packed_format(rec) == 2 and ops.packed_format(rec.clone()) == 2
    assert torch.equal(y1, y2)


@pytest.mark.parametrize('kernel', [0, 1])
def test_both_forward_kernels_read_mx_records(kernel, mfma_shape):
    """The 8-wave kernel (small grids: the 512-channel 32 x 32 layers) and the one-wave kernel on the same MX records."""
    from eo_vae_amd import ops
    from oracle import mx_emul as M
    torch.manual_seed(11)
    n, c, cout, h, w = 2, 128, 256, 32, 32
    x = torch.randn(n, c, h, w, device='cuda') + 0.2
    gamma, beta = torch.randn(c, device='cuda'), torch.randn(c, device='cuda')
    wt = torch.randn(cout, c, 3, 3, device='cuda') / (c * 9) ** 0.5
    b = torch.randn(cout, device='cuda')
    set_opt(b'c8_w4', kernel)
    with torch.no_grad():
        a = ops.group_norm_silu(x, gamma, beta, packed=True, cout=cout)
        assert ops.packed_format(a) == 2
        y = ops.conv2d(a, wt, b, None, 'normal')
        z = ops.group_norm_silu(x, gamma, beta, packed=False).cpu()
    conv = lambda a_, b_: F.conv2d(a_, b_, padding=1)
    want = M.split_product(z, wt.cpu(), conv) + b.cpu().double().view(1, -1, 1, 1)
    assert rel(y, want) < 3e-6
    sums = ops.gn_sums_of(y, 32)
    if sums is not None:
        yd = y.double().view(n, 32, -1)
        ref = torch.stack([yd.sum(-1), (yd * yd).sum(-1)], -1)
        assert (sums - ref).abs().max() <= 2e-6 * ref.abs().max()


@pytest.mark.parametrize('n,c,cout,hw,res', [(16, 128, 128, 256, True), (16, 256, 512, 64, False), (4, 512, 256, 128, True)])
def test_16x16_kernel_at_the_model_shapes(n, c, cout, hw, res):
    """conv_f16c8_q16.hip at full layer shapes (several tiles per block, several co blocks per halo tile, residual and GroupNorm
    sums): run to run bit-identical -- its waits are counted by hand, a race would show here -- and within summation order of
    the 32 x 32 kernel on the same records (in practice the two are bit-identical: same products, same order inside the MFMAs)."""
    from eo_vae_amd import ops
    torch.manual_seed(hw + c)
    x = torch.randn(n, c, hw, hw, device='cuda') + 0.1
    gamma, beta = torch.randn(c, device='cuda'), torch.randn(c, device='cuda')
    wt = torch.randn(cout, c, 3, 3, device='cuda') / (c * 9) ** 0.5
    b = torch.randn(cout, device='cuda')
    r = torch.randn(n, cout, hw, hw, device='cuda') if res else None
    set_opt(b'c8_w4', 1)
    with torch.no_grad():
        a = ops.group_norm_silu(x, gamma, beta, packed=True, cout=cout)
        assert ops.packed_format(a) == 2

        def run(q16):
            set_opt(b'c8_q16', q16)
            y = ops.conv2d(a, wt, b, r, 'normal')
            sums = ops.gn_sums_of(y, 32)
            torch.cuda.synchronize()
            return y, (sums.clone() if sums is not None else None)
        y1, s1 = run(1)
        for _ in range(5):
            y2, s2 = run(1)
            assert torch.equal(y1, y2)
        y0, s0 = run(0)
    assert rel(y1, y0) < 2e-6
    if s1 is not None and s0 is not None:
        assert (s1 - s0).abs().max() <= 2e-6 * s0.abs().max()
