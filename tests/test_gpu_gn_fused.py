"""GPU: the one-pass (persistent, cooperative) GroupNorm backward against the two-pass kernels and against fp64 torch.
Both go through the same C-ABI entry (eovae_gn_silu_bwd); eovae_gn_bwd_fused(0 / 1) picks the kernel."""
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


@pytest.fixture
def fused_switch():
    from eo_vae_amd import lib as L
    lib = L.load()
    before = lib.eovae_gn_bwd_fused(-1)
    yield lib
    lib.eovae_gn_bwd_fused(before if before in (0, 1, 2) else 1)


# n, c, groups, h, w, silu, fork          slab / 1024 -> (V, team)
CASES = [
    (2, 64, 32, 64, 32, True, False),     # 4  -> V 4, team 1
    (3, 32, 8, 128, 32, True, True),      # 16 -> V 8, team 2
    (2, 96, 32, 32, 32, True, False),     # 3  -> V 1, team 3
    (2, 48, 8, 32, 32, False, False),     # 6  -> V 2, team 3   (AttnBlock norm: no swish)
    (1, 64, 32, 128, 64, False, True),    # 16 -> V 8, team 2
    (2, 512, 32, 32, 32, False, False),   # 16 -> V 8, team 2: the bottleneck's attention norm
    (2, 128, 32, 256, 256, True, True),   # 256 -> V 8, team 32: the encoder's first level at full resolution
    (5, 512, 32, 64, 64, True, False),    # 64 -> V 8, team 8, 160 slabs over 64 teams (ragged last round)
    (1, 32, 2, 512, 512, True, True),     # 4096 -> V 8, team 512: every resident workgroup in ONE team
]


def run(lib, on, x, g, b, gy, gskip, groups, silu, fork):
    from eo_vae_amd import ops
    lib.eovae_gn_bwd_fused(on)
    xg, gg, bg = [t.clone().requires_grad_(True) for t in (x, g, b)]
    if fork:
        y, skip = ops.group_norm_silu_fork(xg, gg, bg, groups, 1e-6, silu)
        torch.autograd.backward([y, skip], [gy, gskip])
    else:
        y = ops.group_norm_silu(xg, gg, bg, groups, 1e-6, silu)
        y.backward(gy)
    torch.cuda.synchronize()
    amax = getattr(xg.grad, '_eovae_amax', None)
    return xg.grad, gg.grad, bg.grad, amax


@pytest.mark.parametrize('n,c,groups,h,w,silu,fork', CASES)
def test_one_pass_backward_equals_two_pass(fused_switch, n, c, groups, h, w, silu, fork):
    lib = fused_switch
    assert lib.eovae_gn_bwd_fused_eligible(n, c, h * w, groups) == 1
    x = (hashed((n, c, h, w), 21, 2.0) + 0.7).cuda()
    g = (1.0 + hashed((c,), 22, 0.2)).cuda()
    b = hashed((c,), 23, 0.2).cuda()
    gy = hashed((n, c, h, w), 24, 1.0).cuda()
    gskip = hashed((n, c, h, w), 25, 0.5).cuda()
    dx1, dg1, db1, _ = run(lib, 2, x, g, b, gy, gskip, groups, silu, fork)   # 2: the one-pass kernel for every shape
    dx0, dg0, db0, _ = run(lib, 0, x, g, b, gy, gskip, groups, silu, fork)
    # same arithmetic per element; only the order of the partial sums differs
    assert rel_err(dx1, dx0) < 2e-6
    assert rel_err(dg1, dg0) < 2e-6
    assert rel_err(db1, db0) < 2e-6
    # and against fp64 torch (the bar of test_groupnorm_silu)
    xr, gr, br = [t.double().cpu().requires_grad_(True) for t in (x, g, b)]
    yr = F.group_norm(xr, groups, gr, br, eps=1e-6)
    if silu:
        yr = yr * torch.sigmoid(yr)
    yr.backward(gy.double().cpu())
    want = xr.grad + (gskip.double().cpu() if fork else 0)
    assert rel_err(dx1, want) < 2e-5
    assert rel_err(dg1, gr.grad) < 2e-5
    assert rel_err(db1, br.grad) < 2e-5
    # twice the same launch: bit-identical (fixed summation order, no atomics)
    dx2, dg2, db2, _ = run(lib, 2, x, g, b, gy, gskip, groups, silu, fork)
    assert torch.equal(dx1, dx2) and torch.equal(dg1, dg2) and torch.equal(db1, db2)


def test_one_pass_backward_reports_the_gradient_maximum(fused_switch):
    """max |dx| (the f16c8 data-gradient pre-scale) comes out of the fused pass exactly"""
    from eo_vae_amd import lib as L
    lib = fused_switch
    n, c, groups, hw = 2, 64, 32, 4096  
    x = (hashed((n, c, 64, 64), 31, 1.5)).cuda()
    g = (1.0 + hashed((c,), 32, 0.2)).cuda()
    b = hashed((c,), 33, 0.2).cuda()
    dy = hashed((n, c, 64, 64), 34, 1e-3).cuda()
    stats = torch.empty(n * groups * 2, device='cuda')
    y = torch.empty_like(x)
    ws = torch.empty(lib.eovae_gn_workspace_bytes(n, c, hw, groups), dtype=torch.uint8, device='cuda')
    L.check(lib.eovae_gn_silu_fwd(L.ptr(x), L.ptr(g), L.ptr(b), L.ptr(y), L.ptr(stats), None, n, c, hw, groups, 1e-6, 1,
                                  L.ptr(ws), L.stream()), 'fwd')
    out = {}
    for on in (2, 0):
        lib.eovae_gn_bwd_fused(on)
        dx = torch.empty_like(x)
        dg, db = torch.empty_like(g), torch.empty_like(b)
        bits = torch.full((1,), 123, dtype=torch.int32, device='cuda')
        L.check(lib.eovae_gn_silu_bwd(L.ptr(x), L.ptr(dy), L.ptr(g), L.ptr(b), L.ptr(stats), None, L.ptr(dx), L.ptr(dg),
                                      L.ptr(db), L.ptr(bits), n, c, hw, groups, 1, L.ptr(ws), L.stream()), 'bwd')
        torch.cuda.synchronize()
        out[on] = (dx, bits.view(torch.float32).item())
        assert out[on][1] == dx.abs().max().item()
    assert abs(out[2][1] - out[0][1]) <= 2e-6 * out[0][1]


def test_shapes_outside_the_plan_take_the_two_pass_kernels(fused_switch):
    lib = fused_switch
    lib.eovae_gn_bwd_fused(1)
    assert lib.eovae_gn_bwd_fused_eligible(2, 512, 16 * 16, 32) == 0      # HW not a multiple of 1024
    assert lib.eovae_gn_bwd_fused_eligible(1, 64, 9 * 7, 32) == 0
    assert lib.eovae_gn_bwd_fused_eligible(1, 32, 4096 * 2048, 1) == 0    # one slab would need more members than CUs
    lib.eovae_gn_bwd_fused(0)
    assert lib.eovae_gn_bwd_fused_eligible(2, 128, 256 * 256, 32) == 0


def test_one_pass_backward_replays_inside_a_hip_graph(fused_switch):
    """A captured launch is replayed with frozen kernel arguments: the hand-off token must still differ from replay to
    replay (device-side epoch), or the granules of the previous replay would be taken for this one's."""
    from eo_vae_amd import lib as L
    lib = fused_switch
    n, c, groups, h, w = 2, 128, 32, 64, 64                      # slab 16 units -> V 4, team 4
    hw = h * w
    x = (hashed((n, c, h, w), 41, 2.0) + 0.3).cuda()
    g = (1.0 + hashed((c,), 42, 0.2)).cuda()
    b = hashed((c,), 43, 0.2).cuda()
    dy = torch.zeros(n, c, h, w, device='cuda')
    stats = torch.empty(n * groups * 2, device='cuda')
    y = torch.empty_like(x)
    ws = torch.empty(lib.eovae_gn_workspace_bytes(n, c, hw, groups), dtype=torch.uint8, device='cuda')
    L.check(lib.eovae_gn_silu_fwd(L.ptr(x), L.ptr(g), L.ptr(b), L.ptr(y), L.ptr(stats), None, n, c, hw, groups, 1e-6, 1,
                                  L.ptr(ws), L.stream()), 'fwd')
    dx, dg, db = torch.empty_like(x), torch.empty_like(g), torch.empty_like(b)

    def call():
        L.check(lib.eovae_gn_silu_bwd(L.ptr(x), L.ptr(dy), L.ptr(g), L.ptr(b), L.ptr(stats), None, L.ptr(dx), L.ptr(dg),
                                      L.ptr(db), None, n, c, hw, groups, 1, L.ptr(ws), L.stream()), 'bwd')
    lib.eovae_gn_bwd_fused(2)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call()                                                   # first use allocates the ticket word: not inside a capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call()
    for r in range(3):
        dy.copy_(hashed((n, c, h, w), 50 + r, 1.0 + r).cuda())   # new gradients, same buffers
        graph.replay()
        torch.cuda.synchronize()
        got = (dx.clone(), dg.clone(), db.clone())
        lib.eovae_gn_bwd_fused(0)
        call()
        torch.cuda.synchronize()
        lib.eovae_gn_bwd_fused(2)
        assert rel_err(got[0], dx) < 2e-6 and rel_err(got[1], dg) < 2e-6 and rel_err(got[2], db) < 2e-6, r


def test_two_streams_use_separate_control_blocks_and_no_timeout_is_recorded(fused_switch):
    """the ticket counter / epoch of the hand-off protocol is per (device, STREAM), so backward passes enqueued
    on two streams at once do not draw tickets from one counter (their teams would interleave and time out); a time-out
    would land in the host-visible word eovae_gn_bwd_timeouts reads, which training_step checks once per step."""
    from eo_vae_amd import ops
    lib = fused_switch
    lib.eovae_gn_bwd_fused(2)
    lib.eovae_gn_bwd_timeouts(1)
    n, c, groups, h, w = 4, 128, 32, 128, 128
    x = hashed((n, c, h, w), 901, 1.0).cuda()
    g = (1.0 + hashed((c,), 902, 0.2)).cuda()
    b = hashed((c,), 903, 0.2).cuda()
    gy = hashed((n, c, h, w), 904, 1e-3).cuda()
    ref = run(lib, 2, x, g, b, gy, None, groups, True, False)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None, None]
    torch.cuda.synchronize()
    for rep in range(4):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                xg, gg, bg = [t.clone().requires_grad_(True) for t in (x, g, b)]
                ops.group_norm_silu(xg, gg, bg, groups, 1e-6, True).backward(gy)
                outs[i] = (xg.grad, gg.grad, bg.grad)
    torch.cuda.synchronize()
    for o in outs:
        for a_, r_ in zip(o, ref[:3]):
            assert torch.equal(a_, r_)            # fixed summation order: bit-identical whichever stream ran it
    assert lib.eovae_gn_bwd_timeouts(0) == 0
    ops.raise_on_gn_backward_timeouts()           # no-op without a time-out
