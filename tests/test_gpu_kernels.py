"""GPU: every HIP kernel against a plain fp32 PyTorch-CPU statement of the same op.
Tolerances are written per test; fp32 MFMA is an exact fmaf chain, so differences are
summation-order round-off only."""
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


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def ref_conv(x, w, b, mode):
    if mode == 'normal':
        return F.conv2d(x, w, b, padding=(w.shape[-1] - 1) // 2)
    if mode == 'down':
        return F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    return F.conv2d(F.interpolate(x, scale_factor=2.0, mode='nearest'), w, b, padding=1)


CONV_CASES = [
    # n, cin, cout, h, w, ks, mode
    (2, 32, 32, 16, 16, 3, 'normal'),
    (1, 128, 128, 32, 32, 3, 'normal'),
    (2, 3, 128, 24, 40, 3, 'normal'),      # dynamic conv_in shape (ragged tile edges)
    (2, 128, 3, 24, 40, 3, 'normal'),      # dynamic conv_out shape
    (1, 13, 128, 9, 33, 3, 'normal'),      # odd sizes, S2L1C channel count
    (1, 256, 64, 16, 16, 3, 'normal'),     # Cout < tile
    (2, 64, 192, 16, 16, 1, 'normal'),     # 1x1 shortcut, Cout not a multiple of 128
    (1, 512, 512, 8, 8, 1, 'normal'),      # attention q/k/v/proj
    (2, 32, 32, 16, 16, 3, 'down'),
    (1, 128, 128, 64, 64, 3, 'down'),
    (2, 32, 32, 8, 8, 3, 'up'),
    (1, 256, 256, 16, 16, 3, 'up'),
]


@pytest.mark.parametrize('n,cin,cout,h,w,ks,mode', CONV_CASES)
def test_conv2d_fwd_bwd(n, cin, cout, h, w, ks, mode):
    from eo_vae_amd import ops
    x = hashed((n, cin, h, w), 1, 1.0)
    wt = hashed((cout, cin, ks, ks), 2, 1.0 / (cin * ks * ks) ** 0.5)
    b = hashed((cout,), 3, 0.1)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = ref_conv(xr, wr, br, mode)
    res = hashed(tuple(yr.shape), 5, 1.0)
    rr = res.clone().requires_grad_(True)
    gy = hashed(tuple(yr.shape), 4, 1.0)
    (yr + rr).backward(gy)
    xg, wg, bg, rg = [t.cuda().requires_grad_(True) for t in (x, wt, b, res)]
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f32'     # exact-fp32 MFMA kernels (conv_f32.hip)
    try:
        yg = ops.conv2d(xg, wg, bg, rg, mode)
        yg.backward(gy.cuda())
        torch.cuda.synchronize()
    finally:
        ops.CONV_PRECISION = old
    assert rel_err(yg, yr + rr) < 2e-6
    assert rel_err(xg.grad, xr.grad) < 1e-5
    assert rel_err(wg.grad, wr.grad) < 2e-5      # atomically accumulated split-K partial sums
    assert rel_err(bg.grad, br.grad) < 1e-5
    assert rel_err(rg.grad, rr.grad) == 0.0


GN_CASES = [(2, 32, 8, 8, True), (2, 128, 32, 32, True), (1, 512, 16, 16, False),
            (2, 64, 9, 7, True), (1, 128, 128, 128, True)]


@pytest.mark.parametrize('n,c,h,w,silu', GN_CASES)
def test_groupnorm_silu(n, c, h, w, silu):
    from eo_vae_amd import ops
    x = hashed((n, c, h, w), 11, 2.0) + 0.7
    g = 1.0 + hashed((c,), 12, 0.2)
    b = hashed((c,), 13, 0.2)
    gy = hashed((n, c, h, w), 14, 1.0)
    xr, gr, br = [t.clone().requires_grad_(True) for t in (x, g, b)]
    yr = F.group_norm(xr, 32, gr, br, eps=1e-6)
    if silu:
        yr = yr * torch.sigmoid(yr)
    yr.backward(gy)
    xg, gg, bg = [t.cuda().requires_grad_(True) for t in (x, g, b)]
    yg = ops.group_norm_silu(xg, gg, bg, 32, 1e-6, silu)
    yg.backward(gy.cuda())
    torch.cuda.synchronize()
    assert rel_err(yg, yr) < 1e-5
    assert rel_err(xg.grad, xr.grad) < 2e-5
    assert rel_err(gg.grad, gr.grad) < 2e-5
    assert rel_err(bg.grad, br.grad) < 2e-5


def test_fails_loudly_on_cpu_tensor():
    from eo_vae_amd import ops
    from eo_vae_amd.lib import EovaeError
    with pytest.raises(EovaeError):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3))


LOWP_CASES = [(2, 32, 32, 16, 16, 'normal'), (1, 128, 128, 64, 64, 'normal'), (2, 3, 128, 24, 40, 'normal'),
              (2, 128, 3, 24, 40, 'normal'), (1, 13, 128, 9, 33, 'normal'), (1, 256, 64, 16, 16, 'normal'),
              (4, 512, 512, 32, 32, 'normal'), (2, 32, 32, 16, 16, 'down'), (2, 32, 32, 8, 8, 'up'),
              (1, 256, 256, 32, 32, 'up'), (2, 64, 128, 12, 20, 'up'), (1, 512, 256, 16, 16, 'up'),
              (2, 128, 128, 32, 32, 'down'), (1, 256, 256, 24, 40, 'down'), (1, 128, 64, 64, 64, 'down')]


@pytest.mark.parametrize('precision,tol', [('bf16x3', 3e-5), ('bf16', 2e-2), ('f16c8', 1e-4)])
@pytest.mark.parametrize('n,cin,cout,h,w,mode', LOWP_CASES)
def test_conv3x3_split_bf16(n, cin, cout, h, w, mode, precision, tol):
    _lowp_case(n, cin, cout, h, w, 3, mode, precision, tol)


@pytest.mark.parametrize('gscale', [2.0 ** -24, 2.0 ** 20])
@pytest.mark.parametrize('n,cin,cout,h,w,mode', [(1, 128, 128, 64, 64, 'normal'), (1, 256, 256, 32, 32, 'up'),
                                                 (2, 128, 128, 32, 32, 'down'), (2, 64, 96, 17, 35, 'normal')])
def test_conv3x3_f16c8_gradient_range(n, cin, cout, h, w, mode, gscale):
    """f16c8 data gradients far outside f16's range (loss gradients are ~1/numel): the kernel pre-scales
    dy by the power of two derived from its amax and un-scales in the epilogue, so the result must be
    exactly as accurate as at unit scale."""
    _lowp_case(n, cin, cout, h, w, 3, mode, 'f16c8', 1e-4, gscale=gscale)


@pytest.mark.parametrize('precision,tol', [('bf16x3', 3e-5), ('bf16', 2e-2)])
@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 64, 192, 16, 16), (4, 512, 512, 32, 32), (2, 32, 32, 8, 8),
                                            (1, 256, 128, 40, 24), (3, 64, 64, 9, 7),
                                            (1, 48, 40, 10, 6)])     # K % 32 != 0: predicated stage loads
def test_conv1x1_split_bf16(n, cin, cout, h, w, precision, tol):
    _lowp_case(n, cin, cout, h, w, 1, 'normal', precision, tol)


@pytest.mark.parametrize('xscale', [1e4, 3e5, 1e-6, 2.0 ** -30])
@pytest.mark.parametrize('n,cin,cout,h,w,mode', [(2, 32, 128, 32, 32, 'normal'), (2, 128, 128, 32, 32, 'down'),
                                                 (1, 128, 128, 16, 32, 'up'), (1, 64, 128, 32, 64, 'normal')])
def test_conv3x3_f16c8_input_range(n, cin, cout, h, w, mode, xscale):
    """Un-normalised inputs of 3x3 layers (the residual stream in front of Downsample / Upsample, the latent in front
    of the decoder's conv_in) far outside f16's range -- beyond 65 504 they would become inf, below 6e-5 subnormal:
    forward, data gradient and weight gradient pre-scale x by the power of two derived from its amax (ops.range_amax),
    so the result is as accurate as at unit scale.  """
    _lowp_case(n, cin, cout, h, w, 3, mode, 'f16c8', 1e-4, xscale=xscale)


def test_f16c8_range_guard_can_be_switched_off():
    """EOVAE_F16C8_RANGE_GUARD=0 (ops.RANGE_GUARD) restores the round-2 behaviour: no absmax pass, |x| < 65 504 assumed
    -- and an input beyond it overflows, which is what the guard is for."""
    from eo_vae_amd import ops
    old, oldg = ops.CONV_PRECISION, ops.RANGE_GUARD
    ops.CONV_PRECISION, ops.RANGE_GUARD = 'f16c8', False
    try:
        x = torch.full((1, 32, 16, 32), 1e5, device='cuda')
        w = torch.ones(128, 32, 3, 3, device='cuda') / 288
        with torch.no_grad():
            y = ops.conv2d(x, w, None, None, 'normal')
        assert not torch.isfinite(y).all()
        ops.RANGE_GUARD = True
        with torch.no_grad():
            y = ops.conv2d(x, w, None, None, 'normal')
        assert torch.isfinite(y).all() and abs(float(y[0, 0, 8, 16]) / 1e5 - 1.0) < 1e-4
    finally:
        ops.CONV_PRECISION, ops.RANGE_GUARD = old, oldg


def _lowp_case(n, cin, cout, h, w, ks, mode, precision, tol, gscale=1.0, xscale=1.0):
    """bf16x3 (hi*hi + hi*lo + lo*hi on the bf16 MFMA) must be fp32-grade: ~2^-17 per product;
    plain bf16 is the flagged fast mode (2^-9) and only has to be sane."""
    from eo_vae_amd import ops
    x = hashed((n, cin, h, w), 1, 1.0) * xscale
    wt = hashed((cout, cin, ks, ks), 2, 1.0 / (cin * ks * ks) ** 0.5)
    b = hashed((cout,), 3, 0.1)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = ref_conv(xr, wr, br, mode)
    gy = hashed(tuple(yr.shape), 4, 1.0) * gscale
    yr.backward(gy)
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = precision
    try:
        xg, wg, bg = [t.cuda().requires_grad_(True) for t in (x, wt, b)]
        yg = ops.conv2d(xg, wg, bg, None, mode)
        yg.backward(gy.cuda())
        torch.cuda.synchronize()
    finally:
        ops.CONV_PRECISION = old
    assert rel_err(yg, yr) < tol
    assert rel_err(xg.grad, xr.grad) < tol
    wtol = max(tol, 2e-5)
    if precision == 'f16c8' and mode in ('up', 'down') and ops.L.load().eovae_get_option(b'c8_wgrad_corr') == 0:
        # round 5: the row-streaming weight gradients (here: the sub-pixel forms) multiply f16(dY) * f16(X) alone -- two f16
        # roundings per term, which on this test's incoherent data is ~3e-4 of the largest entry.  That the kernel computes
        # EXACTLY those products is pinned in tests/test_gpu_wgrad_rows.py (float64 sum of the same products, 3e-6).
        wtol = 5e-4
    assert rel_err(wg.grad, wr.grad) < wtol
    assert rel_err(bg.grad, br.grad) < 2e-5      # fp32 sums of dy (fused into the wgrad kernel)


@pytest.mark.parametrize('n,c,cout,h,w', [(2, 64, 96, 24, 40), (1, 128, 128, 64, 64), (2, 256, 128, 17, 35)])
def test_groupnorm_operand_records_equal_the_fp32_hand_over(n, c, cout, h, w):
    """f16c8: GroupNorm+SiLU handing its result to the 3x3 convolution as operand records (converted once, in the
    GroupNorm-apply pass) must give what the fp32 hand-over gives, up to the last-bit differences of the two
    GroupNorm-apply kernels' own arithmetic and the weight gradient's atomic ordering."""
    from eo_vae_amd import ops
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 'f16c8'
    lib = ops.L.load()
    # "the same result" holds for the same arithmetic: the fp32 hand-over runs the fp8 correction product in the forward
    # and in the weight gradient, so the records get the same here (round 5's defaults for records -- MX-FP6 forward, plain
    # f16 weight gradient -- are pinned in tests/test_gpu_mx.py and tests/test_gpu_wgrad_rows.py)
    keep = {k: lib.eovae_get_option(k) for k in (b'c8_mx', b'c8_wgrad_corr')}
    lib.eovae_set_option(b'c8_mx', 0)
    lib.eovae_set_option(b'c8_wgrad_corr', 1)
    try:
        x = hashed((n, c, h, w), 11, 1.0)
        gam, bet = 1.0 + hashed((c,), 12, 0.2), hashed((c,), 13, 0.2)
        wt, b = hashed((cout, c, 3, 3), 14, 1.0 / (c * 9) ** 0.5), hashed((cout,), 15, 0.1)
        gy = hashed((n, cout, h, w), 16, 1.0)
        outs = []
        for packed in (False, True):
            xg, gg, bg, wg, cb = [t.cuda().requires_grad_(True) for t in (x, gam, bet, wt, b)]
            a = ops.group_norm_silu(xg, gg, bg, 32, 1e-6, True, packed)
            assert ops.is_packed(a) == packed
            y = ops.conv2d(a, wg, cb, None, 'normal')
            y.backward(gy.cuda())
            torch.cuda.synchronize()
            outs.append((y.detach().cpu(), xg.grad.cpu(), wg.grad.cpu(), gg.grad.cpu(), cb.grad.cpu()))
        for k in (0, 1, 2, 3, 4):
            assert rel_err(outs[1][k], outs[0][k]) < 1e-5, k      # (one-ulp differences of y flip a few e5m2 roundings)
        # and against plain fp32 torch
        xr, gr, br, wr, cr = [t.clone().requires_grad_(True) for t in (x, gam, bet, wt, b)]
        ref = torch.nn.functional.conv2d(torch.nn.functional.silu(torch.nn.functional.group_norm(xr, 32, gr, br, 1e-6)), wr, cr, padding=1)
        ref.backward(gy)
        assert rel_err(outs[1][0], ref) < 1e-4 and rel_err(outs[1][2], wr.grad) < 1e-4 and rel_err(outs[1][1], xr.grad) < 2e-4
        with pytest.raises(ops.L.EovaeError):     # a consumer that does not understand the records must refuse them
            ops.conv2d(ops.group_norm_silu(x.cuda(), gam.cuda(), bet.cuda(), 32, 1e-6, True, True),
                       hashed((cout, c, 1, 1), 17, 0.1).cuda(), None, None, 'normal')
    finally:
        ops.CONV_PRECISION = old
        for k, v in keep.items():
            lib.eovae_set_option(k, v)


@pytest.mark.parametrize('n,cin,cout,h,w,res,mode', [
    (2, 64, 128, 20, 44, False, 'normal'),      # 4 channels per group, ragged tiles
    (2, 128, 256, 32, 32, True, 'normal'),      # 8 per group: the two lane halves share a group
    (1, 64, 512, 17, 33, True, 'normal'),       # 16 per group, 4 co-blocks
    (3, 128, 128, 64, 64, True, 'normal'),      # several tiles per persistent block
    (2, 128, 128, 32, 48, False, 'down'),       # space-to-depth stride-2 form
    (2, 64, 256, 12, 20, False, 'up'),          # sub-pixel upsample form (pixel-shuffled store)
    (1, 128, 128, 32, 32, False, 'up'),
])
def test_conv_epilogue_groupnorm_sums(n, cin, cout, h, w, res, mode):
    """f16c8 forward: the (sum, sum of squares) per (image, group of Cout/32 channels) that the epilogue adds up
    for the following GroupNorm must equal the sums over the tensor it stored, and a GroupNorm fed with the tag
    must equal a GroupNorm that measures the tensor itself (reference layers.py:91-92: norm reads conv's output)."""
    from eo_vae_amd import ops
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = 'f16c8'
    try:
        x = hashed((n, cin, h, w), 21, 1.0).cuda()
        wt, b = hashed((cout, cin, 3, 3), 22, 1.0 / (cin * 9) ** 0.5).cuda(), hashed((cout,), 23, 0.5).cuda()
        ho, wo = (h // 2, w // 2) if mode == 'down' else ((2 * h, 2 * w) if mode == 'up' else (h, w))
        r = hashed((n, cout, ho, wo), 24, 1.0).cuda() if res else None
        y = ops.conv2d(x, wt, b, r, mode)
        sums = ops.gn_sums_of(y, 32)
        assert sums is not None and sums.dtype == torch.float64 and tuple(sums.shape) == (n, 32, 2)
        yd = y.double().view(n, 32, -1)
        want = torch.stack([yd.sum(-1), (yd * yd).sum(-1)], -1)
        torch.cuda.synchronize()
        assert (sums - want).abs().max() <= 2e-6 * want.abs().max()
        gam, bet = (1.0 + hashed((cout,), 25, 0.2)).cuda(), hashed((cout,), 26, 0.2).cuda()
        a_tag = ops.group_norm_silu(y, gam, bet, 32, 1e-6, True)
        a_own = ops.group_norm_silu(y.clone(), gam, bet, 32, 1e-6, True)        # clone: no tag
        assert ops.gn_sums_of(y.clone(), 32) is None and ops.gn_sums_of(y, 16) is None
        assert rel_err(a_tag.cpu(), a_own.cpu()) < 2e-6
        y.add_(1.0)                                                             # edited in place: tag is void
        assert ops.gn_sums_of(y, 32) is None
    finally:
        ops.CONV_PRECISION = old


def test_packed_records_survive_detach_and_clone():
    """detach() / clone() of GroupNorm's operand-record output must not silently turn the bytes into fp32
    values.  The record tensor is a Tensor subclass: both copies are still records and convolve to the same result."""
    from eo_vae_amd import ops
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f16c8'
    try:
        torch.manual_seed(5)
        x = torch.randn(2, 64, 16, 32, device='cuda')
        g, b = torch.randn(64, device='cuda'), torch.randn(64, device='cuda')
        w = torch.randn(128, 64, 3, 3, device='cuda') / 24
        with torch.no_grad():
            a = ops.group_norm_silu(x, g, b, packed=True)
            assert ops.is_packed(a) and ops.is_packed(a.detach()) and ops.is_packed(a.clone())
            y = ops.conv2d(a, w, None, None, 'normal')
            assert torch.equal(ops.conv2d(a.detach(), w, None, None, 'normal'), y)
            assert torch.equal(ops.conv2d(a.clone(), w, None, None, 'normal'), y)
            ref = ops.conv2d(ops.group_norm_silu(x, g, b, packed=False), w, None, None, 'normal')
            assert rel_err(y, ref) < 2e-5
            with pytest.raises(ops.L.EovaeError, match='operand records'):
                a + 1.0
            with pytest.raises(ops.L.EovaeError):
                ops.conv2d(a, torch.randn(128, 64, 1, 1, device='cuda'), None, None, 'normal')   # not a 3x3 consumer
        torch.cuda.synchronize()
    finally:
        ops.CONV_PRECISION = old


@pytest.mark.parametrize('n,cin,cout,h,w,packed', [(2, 64, 128, 32, 32, False), (3, 128, 192, 24, 40, True), (1, 512, 64, 16, 32, True)])
def test_plain_weight_gradient_is_bit_reproducible(n, cin, cout, h, w, packed):
    """Round 3: the plain f16c8 weight gradient goes through per-block partial tiles and a fold in a fixed order (no
    float atomics): two runs give identical bits, and the result equals the atomic path's up to summation order."""
    from eo_vae_amd import ops
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, 'f16c8'
    try:
        x = hashed((n, cin, h, w), 301, 1.0).cuda()
        wt = hashed((cout, cin, 3, 3), 302, 1.0 / (cin * 9) ** 0.5).cuda()
        b = hashed((cout,), 303, 0.1).cuda()
        gy = hashed((n, cout, h, w), 304, 1e-3).cuda()
        g = (1.0 + hashed((cin,), 305, 0.2)).cuda()
        be = hashed((cin,), 306, 0.2).cuda()

        def run(direct):
            keep, ops.WGRAD_DIRECT = ops.WGRAD_DIRECT, direct
            try:
                wg, bg = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
                inp = ops.group_norm_silu(x, g, be, 32, 1e-6, True, True) if packed else x
                ops.conv2d(inp, wg, bg).backward(gy)
                torch.cuda.synchronize()
                return wg.grad.clone(), bg.grad.clone()
            finally:
                ops.WGRAD_DIRECT = keep
        d1, b1 = run(True)
        d2, _ = run(True)
        a1, ba = run(False)
        assert torch.equal(d1, d2)
        assert rel_err(d1, a1) < 2e-6 and rel_err(b1, ba) < 2e-6
    finally:
        ops.CONV_PRECISION = old


def test_weight_prepack_equals_the_per_layer_packs():
    """ops.WeightPrepack (two launches for all plain 3x3 layers) writes the bytes the per-layer pack_pair calls write:
    tiles and the amax header (the scratch behind the header holds the amax partials and may differ)."""
    from eo_vae_amd import lib as L, ops
    lib = L.load()
    shapes = [(128, 128), (256, 128), (64, 512), (512, 512), (96, 32)]
    ws = [hashed((co, ci, 3, 3), 400 + i, (i + 1) * 0.1 / (ci * 9) ** 0.5).cuda() for i, (co, ci) in enumerate(shapes)]
    ws[2].mul_(1e-3)                           # different amax per layer
    pp = ops.WeightPrepack(ws)
    assert len(pp.weights) == len(ws) and not pp.stale()
    pp.run()
    torch.cuda.synchronize()
    try:
        for w in ws:
            got0, got1 = ops.pack_pair_f16c8(w)            # served from the prepack
            assert got0.data_ptr() == pp.packs[w.data_ptr()][0].data_ptr()
        ops.WeightPrepack.release()
        L.check(lib.eovae_set_option(b'c8_pack_gather', 1), 'set_option')   # reference: the per-element gather kernel
        for w in ws:
            co, ci = w.shape[0], w.shape[1]
            ref0, ref1 = ops.pack_pair_f16c8(w)            # the per-layer launches
            torch.cuda.synchronize()
            got0, got1 = pp.packs[w.data_ptr()]
            assert ref0.data_ptr() != got0.data_ptr()
            for flip, (ref, got) in enumerate(((ref0, got0), (ref1, got1))):
                nb = lib.eovae_conv3x3_f16c8_packed_bytes(co, ci, flip) - 256 * 4 - 12   # tiles + the header word
                assert torch.equal(ref[:nb], got[:nb]), (co, ci, flip)
        # the per-layer entries on the LDS-staged kernel (round 4) write the same bytes as the gather kernel: pair, and each
        # orientation alone (the expanded sub-pixel / space-to-depth weights are packed this way)
        for w in ws:
            co, ci = w.shape[0], w.shape[1]
            L.check(lib.eovae_set_option(b'c8_pack_gather', 1), 'set_option')
            ref0, ref1 = ops.pack_pair_f16c8(w)
            L.check(lib.eovae_set_option(b'c8_pack_gather', 0), 'set_option')
            got0, got1 = ops.pack_pair_f16c8(w)
            one0, one1 = ops._pack3x3(w, co, ci, False, True), ops._pack3x3(w, co, ci, True, True)
            torch.cuda.synchronize()
            for flip, (ref, got, one) in enumerate(((ref0, got0, one0), (ref1, got1, one1))):
                nb = lib.eovae_conv3x3_f16c8_packed_bytes(co, ci, flip) - 256 * 4 - 12
                assert torch.equal(ref[:nb], got[:nb]) and torch.equal(ref[:nb], one[:nb]), (co, ci, flip)
    finally:
        L.check(lib.eovae_set_option(b'c8_pack_gather', 0), 'set_option')
        ops.WeightPrepack.release()
    # a pack is handed out only for the tensor it was made from -- a view that shares offset 0 but not the
    # shape, or a weight edited in place after run(), falls through to the per-layer pack
    pp.run()
    try:
        view = ws[0][:64]
        assert view.data_ptr() == ws[0].data_ptr()
        assert ops.pack_pair_f16c8(view)[0].data_ptr() != pp.packs[ws[0].data_ptr()][0].data_ptr()
        assert ops.pack_pair_f16c8(ws[3])[0].data_ptr() == pp.packs[ws[3].data_ptr()][0].data_ptr()
        ws[3].mul_(2.0)
        assert ops.pack_pair_f16c8(ws[3])[0].data_ptr() != pp.packs[ws[3].data_ptr()][0].data_ptr()
    finally:
        ops.WeightPrepack.release()
    torch.cuda.synchronize()
    ws[1] = ws[1].clone()                      # a weight whose storage moved
    pp.weights[1] = ws[1]
    assert pp.stale()
