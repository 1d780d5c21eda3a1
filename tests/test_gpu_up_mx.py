"""GPU: the x2-upsample forward on MX operand records (round 6; csrc/conv_f16c8_q16up.hip, layers.py:47-50).

Checked at the levels tests/test_gpu_mx.py uses for the plain layers: (1) the bytes -- the records eovae_records_mx_from_f32
writes from an fp32 tensor decode to exactly what oracle/mx_emul.py says they hold (f16 of the pre-scaled x; block-scaled
e2m3 of x and of its f16 residual per pixel and 32 channels); (2) the product -- the kernel's output equals the float64 sum of
the SAME three products on the folded parity weights to summation-order accuracy; (3) the scheme -- it stays within the f16c8
error of the fp64 layer as the reference executes it (nearest x2, then 3x3), and agrees with the 8-wave kernel's sub-pixel
mode (library option c8_up_q16 = 0) that it replaces; (4) the GroupNorm sums its epilogue hands to the next layer."""

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
    keep = {k: lib.eovae_get_option(k) for k in (b'c8_mx', b'c8_up_q16')}
    L.check(lib.eovae_set_option(b'c8_mx', 1), 'set_option')
    L.check(lib.eovae_set_option(b'c8_up_q16', 1), 'set_option')
    yield
    ops.CONV_PRECISION = old
    for k, v in keep.items():
        lib.eovae_set_option(k, v)


def set_opt(name, v):
    from eo_vae_amd import lib as L
    L.check(L.load().eovae_set_option(name, v), 'set_option')


def records_of(x, amax_bits):
    from eo_vae_amd import lib as L, ops
    n, c, h, w = x.shape
    rec = torch.empty_like(x)
    L.check(L.load().eovae_records_mx_from_f32(L.ptr(x), L.ptr(amax_bits), L.ptr(rec), n, c, h * w, L.stream()),
            'eovae_records_mx_from_f32')
    return ops.mark_packed_mx(rec)


def plane_values(rec):
    """MX operand records -> (f16 values, lo, hi) as float64 [n, c, hw] (tests/test_gpu_mx.py:plane_values)."""
    from eo_vae_amd import ops
    from oracle import mx_emul as M
    n, c, h, w = rec.shape
    fpl, spl = ops.mx_planes(rec)
    f = fpl.cpu().numpy().view(np.float16).reshape(n, c // 8, h * w, 8).astype(np.float64)
    out = [torch.from_numpy(f).permute(0, 1, 3, 2).reshape(n, c, h * w)]
    a = spl.cpu().numpy()
    for part in range(2):
        pieces = np.concatenate([a[:, :, 2 * part], a[:, :, 2 * part + 1]], axis=-1)
        out.append(torch.from_numpy(M.decode_blocks(pieces)).permute(0, 1, 3, 2).reshape(n, c, h * w))
    return out


def prescale_of(x):
    """2^(13 - e), e = exponent of max|x| (the range guard's power of two), as the kernels derive it from the amax bits."""
    e = int(np.floor(np.log2(float(x.abs().max()))))
    return 2.0 ** (13 - e)


@pytest.mark.parametrize('n,c,h,w,scale', [(2, 64, 8, 32, 1.0), (1, 128, 16, 16, 3e5), (1, 32, 5, 7, 1e-6)])
def test_records_from_f32_hold_the_emulated_operands(n, c, h, w, scale):
    from eo_vae_amd import ops
    from oracle import mx_emul as M
    torch.manual_seed(c + h)
    x = (torch.randn(n, c, h, w) * scale * torch.exp2(torch.randint(-6, 3, (n, c, 1, 1)).float())).cuda()
    bits = ops.absmax_bits(x)
    rec = records_of(x, bits)
    torch.cuda.synchronize()
    xs = x.cpu() * prescale_of(x.cpu())
    f, lo, hi = plane_values(rec)
    xh = M.f16(xs)
    assert torch.equal(f, xh.double().reshape(n, c, -1))
    assert torch.equal(hi, M.mx6_quant(xs, 1).reshape(n, c, -1))
    assert torch.equal(lo, M.mx6_quant(xs - xh, 1).reshape(n, c, -1))
    # without an amax pointer the values go in unscaled
    f0, _, _ = plane_values(records_of(x, None))
    assert torch.equal(f0, M.f16(x.cpu()).double().reshape(n, c, -1))


def folded_weights(wt):
    """eovae_subpix_expand on the CPU, in its fp32 order of additions: [4][Cout][Cin][3][3], parity par = 2 py + px."""
    def contrib(par, t, k):
        return (k == 0 if t == 0 else (k >= 1 if t == 1 else False)) if par == 0 else (False if t == 0 else (k <= 1 if t == 1 else k == 2))
    cout, cin = wt.shape[:2]
    we = torch.zeros(4, cout, cin, 3, 3)
    for par in range(4):
        py, px = par >> 1, par & 1
        for ty in range(3):
            for tx in range(3):
                v = torch.zeros(cout, cin)
                for ky in range(3):
                    for kx in range(3):
                        if contrib(py, ty, ky) and contrib(px, tx, kx):
                            v = v + wt[:, :, ky, kx]
                we[par, :, :, ty, tx] = v
    return we


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 64, 128, 16, 32), (1, 128, 256, 8, 64), (3, 96, 128, 24, 32), (2, 256, 256, 32, 32)])
def test_up_forward_equals_the_emulated_product(n, cin, cout, h, w):
    from eo_vae_amd import lib as L, ops
    from oracle import mx_emul as M
    assert L.load().eovae_conv3x3_f16c8_up_mx_eligible(n, cin, h, w, cout) == 1
    torch.manual_seed(cin + h)
    x = (torch.randn(n, cin, h, w) * 3 + 0.5).cuda()
    wt = (torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5).cuda()
    b = torch.randn(cout).cuda()

    def fwd(q16):
        set_opt(b'c8_up_q16', q16)
        with torch.no_grad():
            y = ops.conv2d(x, wt, b, None, 'up')
        torch.cuda.synchronize()
        return y
    y_new, y_old = fwd(1), fwd(0)
    xs = prescale_of(x.cpu())
    xc = x.cpu() * xs
    we = folded_weights(wt.cpu())
    conv = lambda a_, b_: F.conv2d(a_, b_, padding=1)                                     # noqa: E731
    want = torch.zeros(n, cout, 2 * h, 2 * w, dtype=torch.float64)
    for par in range(4):
        want[:, :, (par >> 1)::2, (par & 1)::2] = M.split_product(xc, we[par], conv)
    want = want / xs + b.cpu().double().view(1, -1, 1, 1)
    exact = F.conv2d(F.interpolate(x.cpu().double(), scale_factor=2, mode='nearest'), wt.cpu().double(), b.cpu().double(), padding=1)
    assert rel(y_new, want) < 3e-6          # the same three products: fp32 summation order
    assert rel(y_new, exact) < 5e-5         # the scheme against the layer as the reference executes it, in fp64
    assert rel(y_old, exact) < 5e-5 and rel(y_new, y_old) < 5e-5
    assert rel(y_new, y_old) > 1e-8         # MX-FP6 against fp8 correction products: different arithmetic
    # the following GroupNorm(32)'s sums from the epilogue
    sums = ops.gn_sums_of(fwd(1), 32)
    assert sums is not None
    yg = y_new.double().reshape(n, 32, -1)
    assert rel(sums[:, :, 0], yg.sum(-1)) < 1e-5 and rel(sums[:, :, 1], (yg * yg).sum(-1)) < 1e-5


def test_up_forward_at_the_model_shapes_agrees_with_the_8_wave_kernel():
    """The decoder's three Upsample layers at the benchmark's batch (model.py:260 -> layers.py:41-50): 512 -> 512 at 32^2 and
    64^2, 256 -> 256 at 128^2, B = 16 -- every tile of the persistent walk, the XCD map and both row parities; against the
    8-wave kernel's sub-pixel mode (its own tests pin it to fp64) within the two schemes' error, plus linearity in x (a
    size-independent property: the records' power-of-two pre-scale must cancel exactly)."""
    from eo_vae_amd import ops
    for n, c, hw in [(16, 512, 32), (16, 512, 64), (16, 256, 128)]:
        torch.manual_seed(hw)
        x = torch.randn(n, c, hw, hw, device='cuda') * 2
        wt = torch.randn(c, c, 3, 3, device='cuda') / (c * 9) ** 0.5
        b = torch.randn(c, device='cuda')
        with torch.no_grad():
            set_opt(b'c8_up_q16', 1)
            y1 = ops.conv2d(x, wt, b, None, 'up')
            y4 = ops.conv2d(x * 4.0, wt, None, None, 'up')
            y1n = ops.conv2d(x, wt, None, None, 'up')
            set_opt(b'c8_up_q16', 0)
            y0 = ops.conv2d(x, wt, b, None, 'up')
        torch.cuda.synchronize()
        assert rel(y1, y0) < 5e-5
        assert torch.equal(y4, y1n * 4.0)      # x * 4 -> the same records at another pre-scale: bit-identical up to the factor
        del x, wt, y1, y4, y1n, y0
        torch.cuda.empty_cache()


def test_up_layer_trains_through_the_new_forward():
    """Forward (new kernel) + the unchanged backward of the layer against fp64 autograd: y, dx, dw, db."""
    from eo_vae_amd import ops
    torch.manual_seed(9)
    n, c, h, w = 2, 128, 16, 32
    x = torch.randn(n, c, h, w, device='cuda', requires_grad=True)
    wt = (torch.randn(c, c, 3, 3, device='cuda') / (c * 9) ** 0.5).requires_grad_(True)
    b = torch.randn(c, device='cuda', requires_grad=True)
    gy = torch.randn(n, c, 2 * h, 2 * w, device='cuda') * 1e-3
    y = ops.conv2d(x, wt, b, None, 'up')
    y.backward(gy)
    xd, wd, bd = (t.detach().double().cpu().requires_grad_(True) for t in (x, wt, b))
    yd = F.conv2d(F.interpolate(xd, scale_factor=2, mode='nearest'), wd, bd, padding=1)
    yd.backward(gy.double().cpu())
    assert rel(y, yd) < 5e-5 and rel(x.grad, xd.grad) < 1e-4 and rel(b.grad, bd.grad) < 1e-5
    assert rel(wt.grad, wd.grad) < 5e-4       # one-f16-product weight gradient (tests/test_gpu_wgrad_rows.py's bar)


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 128, 64, 16, 32), (1, 256, 128, 32, 32), (3, 128, 96, 16, 64)])
def test_up_data_gradient_equals_the_emulated_product(n, cin, cout, h, w):
    """dX of the x2-upsample layer on MX records of the high-res dY (csrc/conv_f16c8_q16upg.hip): the float64 sum of the SAME
    three products -- f16(dY 2^k) x f16(W'), and the two corrections on the block-scaled e2m3 images (blocks of 32 OUTPUT
    channels per high-res pixel and per weight row / tap) -- over the four parity planes, to summation-order accuracy; the
    scheme against the fp64 adjoint of the layer as the reference executes it; and the 8-wave kernel's mode it replaces."""
    from eo_vae_amd import lib as L, ops
    from oracle import mx_emul as M
    set_opt(b'c8_up_q16', 2)
    assert L.load().eovae_conv3x3_f16c8_up_dgrad_mx_eligible(n, cin, h, w, cout) == 1
    torch.manual_seed(cin + w)
    wt = (torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5).cuda()
    dy = (torch.randn(n, cout, 2 * h, 2 * w) * 3e-4).cuda()

    def bwd(q16):     # the op itself (ops.conv2d takes the sub-pixel forms only for Cout % 128 == 0; the kernel needs Cout % 32)
        set_opt(b'c8_up_q16', 2 * q16)     # 2: the kernel whenever the shape allows (1: only with a tile per CU)
        dx = ops.up_conv_subpix_dgrad(dy, wt, h, w)
        torch.cuda.synchronize()
        return dx
    dx_new, dx_old = bwd(1), bwd(0)
    ds = prescale_of(dy.cpu())
    dyc = dy.cpu() * ds
    we = folded_weights(wt.cpu())                                 # [4][cout][cin][3][3]
    want = torch.zeros(n, cin, h, w, dtype=torch.float64)
    for par in range(4):
        plane = dyc[:, :, (par >> 1)::2, (par & 1)::2]            # P_p[co][r][c] = dY[co][2 r + py][2 c + px]
        wf = we[par].flip(2, 3).transpose(0, 1).contiguous()      # data-gradient operand: [cin][cout][kyf][kxf]
        conv = lambda a_, b_: F.conv2d(a_, b_, padding=1)          # noqa: E731
        want += M.split_product(plane, wf, conv)                   # K blocks: 32 output channels of the layer on both sides
    want = want / ds
    xd = torch.zeros(n, cin, h, w, dtype=torch.float64, requires_grad=True)
    F.conv2d(F.interpolate(xd, scale_factor=2, mode='nearest'), wt.cpu().double(), padding=1).backward(dy.cpu().double())
    assert rel(dx_new, want) < 3e-6
    assert rel(dx_new, xd.grad) < 5e-5 and rel(dx_old, xd.grad) < 5e-5
    assert 1e-8 < rel(dx_new, dx_old) < 5e-5


def test_up_data_gradient_at_the_model_shapes_agrees_with_the_8_wave_kernel():
    """The three layers at the benchmark's batch: every tile of the persistent walk, the XCD map, all four parity planes; against
    the 8-wave kernel's gather mode, plus linearity in dY (the records' power-of-two pre-scale cancels exactly)."""
    from eo_vae_amd import ops
    for n, c, hw in [(16, 512, 32), (16, 512, 64), (16, 256, 128)]:
        torch.manual_seed(hw + 1)
        wt = torch.randn(c, c, 3, 3, device='cuda') / (c * 9) ** 0.5
        dy = torch.randn(n, c, 2 * hw, 2 * hw, device='cuda') * 1e-4
        out = {}
        for q, scale in ((1, 1.0), (1, 8.0), (0, 1.0)):
            set_opt(b'c8_up_q16', 2 * q)
            x = torch.zeros(n, c, hw, hw, device='cuda', requires_grad=True)
            ops.conv2d(x, wt, None, None, 'up').backward(dy * scale)
            out[(q, scale)] = x.grad
            del x
        torch.cuda.synchronize()
        assert rel(out[(1, 1.0)], out[(0, 1.0)]) < 5e-5
        assert torch.equal(out[(1, 8.0)], out[(1, 1.0)] * 8.0)
        del out, wt, dy
        torch.cuda.empty_cache()


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 128, 256, 32, 64), (1, 256, 256, 16, 64), (16, 256, 256, 128, 128)])
def test_stride2_data_gradient_through_the_upsample_kernel(n, cin, cout, h, w):
    """The stride-2 layer's data gradient (adjoint of layers.py:33-37) is an x2-upsample forward on the space-to-depth weights
    Wd'' (ops.down_conv_s2d_dgrad, round 6): against the fp64 adjoint of the layer as the reference executes it (pad (0,1,0,1) +
    3x3 stride 2), against the 8-wave kernel's pixel-shuffle mode it replaces, and the max|dx| tag from the epilogue against a
    reduction pass, bit for bit."""
    from eo_vae_amd import ops
    torch.manual_seed(cin + h)
    wt = (torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5).cuda()
    dy = (torch.randn(n, cout, h // 2, w // 2) * 2e-4).cuda()
    keep = ops.DOWN_DGRAD_UP_MX
    try:
        ops.DOWN_DGRAD_UP_MX = True
        dx_new = ops.down_conv_s2d_dgrad(dy, wt, h, w)
        tag = dx_new._eovae_amax
        assert tag[1] == dx_new._version and torch.equal(tag[0], ops.absmax_bits(dx_new))
        ops.DOWN_DGRAD_UP_MX = False
        dx_old = ops.down_conv_s2d_dgrad(dy, wt, h, w)
    finally:
        ops.DOWN_DGRAD_UP_MX = keep
    torch.cuda.synchronize()
    assert 1e-8 < rel(dx_new, dx_old) < 5e-5
    if n <= 2:
        xd = torch.zeros(n, cin, h, w, dtype=torch.float64, requires_grad=True)
        F.conv2d(F.pad(xd, (0, 1, 0, 1)), wt.cpu().double(), stride=2).backward(dy.cpu().double())
        assert rel(dx_new, xd.grad) < 5e-5 and rel(dx_old, xd.grad) < 5e-5
