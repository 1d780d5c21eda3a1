"""Row-streaming f16c8 weight gradient (csrc/conv_f16c8_wgrad_rows.hip; reference call site: the autograd weight
gradient of nn.Conv2d 3x3, eo_vae/models/modules/layers.py:64-66, 81-83) against
  * the fp32 weight gradient computed by torch on the CPU from the same GroupNorm+SiLU output,
  * the tile kernel it replaces (same arithmetic per product, other summation order),
on shapes that exercise every branch of the walk: several strips per image, row segments (few strips for the blocks of
a slab), units that do not divide evenly among the blocks, one-row segments' halo rows, several ci / co slabs."""
import pytest
import torch

from test_gpu_kernels import hashed, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[1, 0], ids=['corr', 'f16'])
def corr(request):
    """Both arithmetic forms of the row-streaming kernels (library option c8_wgrad_corr): 1 = f16 product + fp8 correction
    product (the r4 arithmetic: the same products as the tile kernel, ~2e-5 of fp64), 0 = f16(dY) * f16(X) alone (round 5,
    the default: checked against the SAME products in float64 -- agreement to summation order -- and against fp64 within the
    rounding of its operands, ~2^-12 per term over incoherent sums)."""
    from eo_vae_amd import lib as L
    lib = L.load()
    keep = lib.eovae_get_option(b'c8_wgrad_corr')
    L.check(lib.eovae_set_option(b'c8_wgrad_corr', request.param), 'set_option')
    yield request.param
    lib.eovae_set_option(b'c8_wgrad_corr', keep)


def f16_prescaled(t):
    """f16(t * 2^k) / 2^k in float64, 2^k = the kernels' range pre-scale: the power of two that puts max|t| into [2^13, 2^14)."""
    import math
    t = t.detach().cpu().float()
    amax = float(t.abs().max())
    s = 2.0 ** (13 - math.floor(math.log2(amax))) if amax > 0 else 1.0
    return (t * s).to(torch.float16).double() / s


def wgrad_f64(xin, gy, wshape, stride=1, padding=1):
    return torch.nn.grad.conv2d_weight(xin.double(), wshape, gy.double(), stride=stride, padding=padding)

SHAPES = [  # n, cin, cout, h, w
    (2, 64, 128, 32, 32),     # one strip per image, segments of 8 rows
    (3, 128, 128, 24, 64),    # 6 strips, 24 rows: segments of 12 rows, units % blocks != 0
    (1, 192, 256, 40, 96),    # 3 ci slabs x 2 co slabs, H not a power of two
    (2, 512, 128, 16, 32),    # 8 ci slabs: few blocks per slab
    (16, 64, 128, 8, 32),     # short strips: 8 rows + 2 halo stages each
    (1, 64, 128, 9, 32),      # odd H: a single segment
]


def _case(n, cin, cout, h, w):
    from eo_vae_amd import ops
    x = hashed((n, cin, h, w), 311, 1.0).cuda()
    wt = hashed((cout, cin, 3, 3), 312, 1.0 / (cin * 9) ** 0.5).cuda()
    b = hashed((cout,), 313, 0.1).cuda()
    gy = hashed((n, cout, h, w), 314, 1e-3).cuda()
    g = (1.0 + hashed((cin,), 315, 0.2)).cuda()
    be = hashed((cin,), 316, 0.2).cuda()
    return ops, x, wt, b, gy, g, be


def _run(ops, x, wt, b, gy, g, be):
    wg, bg = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    inp = ops.group_norm_silu(x, g, be, 32, 1e-6, True, True)
    ops.conv2d(inp, wg, bg).backward(gy)
    torch.cuda.synchronize()
    return wg.grad.clone(), bg.grad.clone()


@pytest.mark.parametrize('n,cin,cout,h,w', SHAPES)
def test_rows_weight_gradient(n, cin, cout, h, w, corr):
    from eo_vae_amd import lib as L
    lib = L.load()
    ops, x, wt, b, gy, g, be = _case(n, cin, cout, h, w)
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f16c8'
    keep = lib.eovae_get_option(b'c8_wgrad_rows')
    try:
        assert ops.WGRAD_DIRECT, 'the row-streaming kernel serves the partial-tile path'
        L.check(lib.eovae_set_option(b'c8_wgrad_rows', 1), 'set_option')
        d1, b1 = _run(ops, x, wt, b, gy, g, be)
        d2, _ = _run(ops, x, wt, b, gy, g, be)
        L.check(lib.eovae_set_option(b'c8_wgrad_rows', 0), 'set_option')
        t1, bt = _run(ops, x, wt, b, gy, g, be)
        assert torch.equal(d1, d2)                      # fixed summation order: bit-reproducible
        # fp32 reference on the CPU: the weight gradient of a plain convolution of the GroupNorm + SiLU output
        xc = x.cpu().double()
        act = torch.nn.functional.silu(torch.nn.functional.group_norm(xc, 32, g.cpu().double(), be.cpu().double(), 1e-6))
        wref = wt.cpu().double().requires_grad_(True)
        torch.nn.functional.conv2d(act, wref, None, padding=1).backward(gy.cpu().double())
        e_ref, e_tile = rel_err(d1.cpu().double(), wref.grad), rel_err(d1, t1)
        print(f'rows vs fp64 {e_ref:.2e}, tile kernel vs fp64 {rel_err(t1.cpu().double(), wref.grad):.2e}, rows vs tile {e_tile:.2e}')
        if corr:
            assert e_ref < 5e-5 and e_tile < 2e-6      # f16c8 arithmetic: ~2e-5 per tensor in either kernel
        else:
            # the f16 parts the records hold (decoded from their bytes: a GroupNorm output recomputed by another kernel may
            # land on the other side of an f16 rounding boundary, 2^-11 of a term); dY is pre-scaled from its amax
            with torch.no_grad():
                zh = ops.records_f16(ops.group_norm_silu(x, g, be, 32, 1e-6, True, True)).cpu()
            want = wgrad_f64(zh, f16_prescaled(gy), wt.shape)
            e_emu = rel_err(d1, want)
            print(f'rows (f16 x f16) vs the same products in fp64 {e_emu:.2e}')
            assert e_emu < 3e-6 and e_ref < 5e-4 and rel_err(t1.cpu().double(), wref.grad) < 5e-5
        assert rel_err(b1, bt) < 2e-6 and rel_err(b1.cpu().double(), gy.cpu().double().sum((0, 2, 3))) < 1e-5
    finally:
        lib.eovae_set_option(b'c8_wgrad_rows', keep)
        ops.CONV_PRECISION = old


@pytest.mark.parametrize('n,cin,cout,h,w', SHAPES)
def test_rows_mfma_shapes_agree(n, cin, cout, h, w, corr):
    """Library option c8_q16: the f16-only kernel on v_mfma_f32_16x16x32_f16 (default) and on v_mfma_f32_32x32x16_f16 multiply
    the same operands; only the order inside the fp32 sums differs."""
    if corr:
        pytest.skip('the correction-product form has one MFMA shape')
    from eo_vae_amd import lib as L
    lib = L.load()
    ops, x, wt, b, gy, g, be = _case(n, cin, cout, h, w)
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f16c8'
    keep = lib.eovae_get_option(b'c8_q16')
    try:
        assert keep >= 1
        got = {}
        for v in (1, 0):
            L.check(lib.eovae_set_option(b'c8_q16', v), 'set_option')
            got[v] = _run(ops, x, wt, b, gy, g, be)
        e = rel_err(got[1][0], got[0][0])
        print(f'16x16x32 vs 32x32x16: {e:.2e}')
        assert e < 2e-6 and float(got[1][0].abs().max()) > 0
        assert rel_err(got[1][1], got[0][1]) < 2e-6    # (the bias gradient does not pass through the MFMAs; float atomics)
    finally:
        lib.eovae_set_option(b'c8_q16', keep)
        ops.CONV_PRECISION = old


@pytest.mark.parametrize('form', ['up', 'down'])
def test_sub_pixel_forms_on_both_mfma_shapes(form, corr):
    """The x2-upsample (16 x 16 x 32 by default) and stride-2 (32 x 32 x 16 by default; c8_q16 = 3 selects the other) weight
    gradients: the two MFMA shapes multiply the same operands."""
    if corr:
        pytest.skip('the correction-product form has one MFMA shape')
    import torch.nn.functional as F
    from eo_vae_amd import ops, lib as L
    lib = L.load()
    n, cin, cout, h, w = 2, 128, 128, 16, 64
    x = (hashed((n, cin, h, w), 341, 1.0) * 2.0).cuda()
    wt = hashed((cout, cin, 3, 3), 342, 1.0 / (cin * 9) ** 0.5).cuda()
    gy = hashed((n, cout, 2 * h, 2 * w) if form == 'up' else (n, cout, h // 2, w // 2), 343, 1e-3).cuda()
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f16c8'
    keep = lib.eovae_get_option(b'c8_q16')
    try:
        got = {}
        for v in (0, 3):
            L.check(lib.eovae_set_option(b'c8_q16', v), 'set_option')
            wg = wt.clone().requires_grad_(True)
            ops.conv2d(x, wg, None, None, form).backward(gy)
            torch.cuda.synchronize()
            got[v] = wg.grad.clone()
        assert float(got[0].abs().max()) > 0 and rel_err(got[3], got[0]) < 2e-6
    finally:
        lib.eovae_set_option(b'c8_q16', keep)
        ops.CONV_PRECISION = old


def test_rows_kernel_is_the_default_and_switchable():
    from eo_vae_amd import lib as L
    lib = L.load()
    assert lib.eovae_get_option(b'c8_wgrad_rows') == 1
    assert lib.eovae_set_option(b'no_such_option', 1) != 0 and b'unknown option' in lib.eovae_last_error()


UP_SHAPES = [  # n, cin, cout, h, w (low-res input)
    (2, 64, 128, 16, 32),     # one strip, one segment pair
    (3, 128, 128, 12, 64),    # two strips, units % blocks != 0
    (1, 192, 256, 20, 96),    # 3 ci slabs x 2 co slabs
    (2, 512, 512, 8, 32),     # the model's 512-channel upsample at a small grid
    (1, 64, 128, 9, 32),      # odd H
]


@pytest.mark.parametrize('n,cin,cout,h,w', UP_SHAPES)
def test_upsample_weight_gradient_rows(n, cin, cout, h, w, corr):
    """x2-upsample layer (layers.py Upsample: nearest x2 + 3x3): the sub-pixel row-streaming weight gradient against fp64
    torch on the CPU and against the tile kernel's sub-pixel mode; bit-reproducible; the bias gradient rides along."""
    import torch.nn.functional as F
    from eo_vae_amd import ops
    x = (hashed((n, cin, h, w), 321, 1.0) * 3.0).cuda()          # not a GroupNorm output: exercises the range pre-scale
    wt = hashed((cout, cin, 3, 3), 322, 1.0 / (cin * 9) ** 0.5).cuda()
    b = hashed((cout,), 323, 0.1).cuda()
    gy = hashed((n, cout, 2 * h, 2 * w), 324, 1e-3).cuda()
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f16c8'
    keep = ops.UP_WGRAD_ROWS

    def run(rows):
        ops.UP_WGRAD_ROWS = rows
        wg, bg = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
        xin = x.clone().requires_grad_(True)
        ops.conv2d(xin, wg, bg, None, 'up').backward(gy)
        torch.cuda.synchronize()
        return wg.grad.clone(), bg.grad.clone()
    try:
        d1, b1 = run(True)
        d2, _ = run(True)
        t1, bt = run(False)
        assert torch.equal(d1, d2)
        xc = F.interpolate(x.cpu().double(), scale_factor=2.0, mode='nearest')
        wref = wt.cpu().double().requires_grad_(True)
        bref = b.cpu().double().requires_grad_(True)
        F.conv2d(xc, wref, bref, padding=1).backward(gy.cpu().double())
        assert rel_err(t1, wref.grad) < 5e-5
        if corr:
            assert rel_err(d1, wref.grad) < 5e-5, rel_err(d1, wref.grad)
            assert rel_err(d1, t1) < 1e-5
        else:
            want = wgrad_f64(F.interpolate(f16_prescaled(x), scale_factor=2.0, mode='nearest'), f16_prescaled(gy), wt.shape)
            assert rel_err(d1, want) < 3e-6, rel_err(d1, want)
            assert rel_err(d1, wref.grad) < 5e-4, rel_err(d1, wref.grad)
        assert rel_err(b1, bref.grad) < 2e-6 and rel_err(bt, bref.grad) < 2e-6
    finally:
        ops.CONV_PRECISION, ops.UP_WGRAD_ROWS = old, keep


DOWN_SHAPES = [  # n, cin, cout, h, w (hi-res input)
    (2, 128, 128, 32, 64),    # one strip
    (3, 128, 64, 24, 128),    # two strips, one 64-column block of co
    (1, 256, 192, 40, 192),   # 2 ci blocks x 3 co blocks
    (2, 512, 512, 16, 64),    # the model's 512-channel stride-2 layer at a small grid
    (1, 128, 128, 18, 64),    # odd output height
]


@pytest.mark.parametrize('n,cin,cout,h,w', DOWN_SHAPES)
def test_stride2_weight_gradient_rows(n, cin, cout, h, w, corr):
    """Stride-2 layer (layers.py Downsample: pad (0,1,0,1) + 3x3 stride 2): the weight gradient through the sub-pixel
    row-streaming kernel with swapped operands, against fp64 torch on the CPU and the tile kernel's space-to-depth mode;
    bit-reproducible; the bias gradient comes from eovae_bias_grad."""
    import torch.nn.functional as F
    from eo_vae_amd import ops
    x = (hashed((n, cin, h, w), 331, 1.0) * 3.0).cuda()
    wt = hashed((cout, cin, 3, 3), 332, 1.0 / (cin * 9) ** 0.5).cuda()
    b = hashed((cout,), 333, 0.1).cuda()
    gy = hashed((n, cout, h // 2, w // 2), 334, 1e-3).cuda()
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f16c8'
    keep = ops.UP_WGRAD_ROWS

    def run(rows):
        ops.UP_WGRAD_ROWS = rows
        wg, bg = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
        xin = x.clone().requires_grad_(True)
        ops.conv2d(xin, wg, bg, None, 'down').backward(gy)
        torch.cuda.synchronize()
        return wg.grad.clone(), bg.grad.clone()
    try:
        d1, b1 = run(True)
        d2, _ = run(True)
        t1, bt = run(False)
        assert torch.equal(d1, d2)
        xc = F.pad(x.cpu().double(), (0, 1, 0, 1))
        wref = wt.cpu().double().requires_grad_(True)
        bref = b.cpu().double().requires_grad_(True)
        F.conv2d(xc, wref, bref, stride=2).backward(gy.cpu().double())
        assert rel_err(t1, wref.grad) < 5e-5
        if corr:
            assert rel_err(d1, wref.grad) < 5e-5, rel_err(d1, wref.grad)
            assert rel_err(d1, t1) < 1e-5
        else:
            want = wgrad_f64(F.pad(f16_prescaled(x), (0, 1, 0, 1)), f16_prescaled(gy), wt.shape, stride=2, padding=0)
            assert rel_err(d1, want) < 3e-6, rel_err(d1, want)
            assert rel_err(d1, wref.grad) < 5e-4, rel_err(d1, wref.grad)
        assert rel_err(b1, bref.grad) < 2e-6 and rel_err(bt, bref.grad) < 2e-6
    finally:
        ops.CONV_PRECISION, ops.UP_WGRAD_ROWS = old, keep


def test_up_and_down_entries_edge_cases():
    """The C entries themselves: an all-zero dy gives exact zeros (and an untouched db accumulator), db = NULL is accepted,
    xamax = NULL (no range pre-scale) agrees with the scaled call inside f16's range, ineligible shapes are refused."""
    import ctypes as C
    from eo_vae_amd import lib as L, ops
    lib = L.load()
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f16c8'
    try:
        n, cin, cout, h, w = 2, 64, 128, 8, 32
        x = hashed((n, cin, h, w), 341, 1.0).cuda()
        dy = hashed((n, cout, 2 * h, 2 * w), 342, 1e-3).cuda()
        ws = torch.empty(lib.eovae_conv3x3_f16c8_wgrad_up_workspace(n, cin, h, w, cout) // 4 + 64, device='cuda')
        amax_dy, amax_x = ops.absmax_bits(dy), ops.absmax_bits(x)

        def up(xx, dd, db, ax):
            dw = torch.full((cout, cin, 3, 3), float('nan'), device='cuda')
            L.check(lib.eovae_conv3x3_f16c8_wgrad_up(L.ptr(xx), L.ptr(dd), L.ptr(dw), L.ptr(db), L.ptr(ops.absmax_bits(dd)), L.ptr(ax),
                                                     n, cin, h, w, cout, L.ptr(ws), L.stream()), 'up')
            torch.cuda.synchronize()
            return dw
        db = torch.zeros(cout, device='cuda')
        z = up(x, torch.zeros_like(dy), db, amax_x)
        assert torch.count_nonzero(z) == 0 and torch.count_nonzero(db) == 0
        a = up(x, dy, None, amax_x)                      # no bias gradient asked for
        b = up(x, dy, None, None)                        # no pre-scale: |x| ~ 1 is inside f16's range
        assert torch.isfinite(a).all() and rel_err(a, b) < 2e-5
        assert lib.eovae_conv3x3_f16c8_wgrad_up_eligible(n, cin, h, 48, cout) == 0       # W % 32
        assert lib.eovae_conv3x3_f16c8_wgrad_up_eligible(n, 32, h, w, cout) == 0         # Cin % 64
        assert lib.eovae_conv3x3_f16c8_wgrad_up(L.ptr(x), L.ptr(dy), L.ptr(a), None, L.ptr(amax_dy), None, n, 32, h, w, cout,
                                                L.ptr(ws), L.stream()) != 0
        # stride-2 entry: x hi-res [n, 128, 16, 64], dy [n, 64, 8, 32]
        ci2, co2 = 128, 64
        x2 = hashed((n, ci2, 2 * h, 2 * w), 343, 1.0).cuda()
        dy2 = hashed((n, co2, h, w), 344, 1e-3).cuda()
        ws2 = torch.empty(lib.eovae_conv3x3_f16c8_wgrad_down_workspace(n, ci2, h, w, co2) // 4 + 64, device='cuda')
        dw2 = torch.full((co2, ci2, 3, 3), float('nan'), device='cuda')
        L.check(lib.eovae_conv3x3_f16c8_wgrad_down(L.ptr(x2), L.ptr(torch.zeros_like(dy2)), L.ptr(dw2), L.ptr(ops.absmax_bits(dy2)),
                                                   L.ptr(ops.absmax_bits(x2)), n, ci2, h, w, co2, L.ptr(ws2), L.stream()), 'down')
        torch.cuda.synchronize()
        assert torch.count_nonzero(dw2) == 0
        assert lib.eovae_conv3x3_f16c8_wgrad_down_eligible(n, 64, h, w, co2) == 0        # Cin % 128
    finally:
        ops.CONV_PRECISION = old
