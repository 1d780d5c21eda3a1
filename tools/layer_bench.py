"""Per-layer timing of the 3x3 conv kernels at the real layer shapes (B=16): algorithmic TFLOP/s.
Calls the C-ABI wrappers directly (no autograd) so only kernel time is measured.
    python tools/layer_bench.py bf16x3 f16c8        (interleaved rounds in one process: rule 24)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eo_vae_amd import ops, lib as L

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

shapes = [(128, 128, 256), (256, 256, 128), (512, 512, 64), (512, 512, 32), (256, 128, 256), (512, 256, 128)]
B = 16
lib = L.load()
precs = sys.argv[1:] or ['bf16x3']
for cin, cout, hw in shapes:
    x = torch.randn(B, cin, hw, hw, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (cin * 9) ** 0.5
    dy = torch.randn(B, cout, hw, hw, device='cuda')
    y = torch.empty_like(dy)
    res = torch.randn_like(dy) if os.environ.get('LB_RES') == '1' else None   # conv2 of a ResnetBlock adds its skip
    gns = torch.zeros(B, 32, 2, dtype=torch.float64, device='cuda') if os.environ.get('LB_GN') == '1' else None   # GroupNorm sums from the epilogue
    flops = 2.0 * B * hw * hw * cin * cout * 9
    fns = {}
    for prec in precs:
        if prec == 'f16c8':
            wp = torch.empty(lib.eovae_conv3x3_f16c8_packed_bytes(cout, cin, 0), dtype=torch.uint8, device='cuda')
            L.check(lib.eovae_conv3x3_f16c8_pack(L.ptr(w), L.ptr(wp), cout, cin, 0, 1.0, L.stream()), 'pack')
            dwp8 = torch.zeros(9 * cout * cin, device='cuda')
            amax = ops.absmax_bits(dy)
            xin, xp = x, 0
            if os.environ.get('LB_XPACK') == '1':   # the layer's input as GroupNorm+SiLU operand records (the model's hand-over)
                xin, xp = ops.group_norm_silu(x, torch.ones(cin, device='cuda'), torch.zeros(cin, device='cuda'), 32, 1e-6, True, True).detach(), 1
            fns[prec] = (lambda wp=wp, xin=xin, xp=xp: lib.eovae_conv3x3_f16c8_fwd(L.ptr(xin), L.ptr(wp), None, L.ptr(res), L.ptr(y), None, B, cin, hw, hw, cout, hw, hw, 0, 0, xp, L.ptr(gns), 32 if gns is not None else 0, L.stream()),
                         lambda dwp8=dwp8, amax=amax, xin=xin, xp=xp: lib.eovae_conv3x3_f16c8_wgrad(L.ptr(xin), L.ptr(dy), L.ptr(dwp8), None, L.ptr(amax), None, B, cin, hw, hw, cout, hw, hw, 0, 0, xp, L.stream()))
        else:
            p = ops._PREC[prec]
            wp = torch.empty(lib.eovae_conv3x3_bf16_packed_elems(cout, cin, 0), dtype=torch.bfloat16, device='cuda')
            lib.eovae_conv3x3_bf16_pack(L.ptr(w), L.ptr(wp), cout, cin, 0, 1.0, L.stream())
            dwp = torch.zeros(9 * cout * cin, device='cuda')
            fns[prec] = (lambda wp=wp, p=p: lib.eovae_conv3x3_bf16_fwd(L.ptr(x), L.ptr(wp), None, None, L.ptr(y), B, cin, hw, hw, cout, hw, hw, 0, p, 0, L.stream()),
                         lambda p=p, dwp=dwp: lib.eovae_conv3x3_bf16_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dwp), None, B, cin, hw, hw, cout, hw, hw, 0, p, 0, L.stream()))
    best = {p: [1e9, 1e9] for p in precs}
    for rnd in range(3):
        for prec in precs:
            f, g = fns[prec]
            best[prec][0] = min(best[prec][0], timeit(f))
            if g is not None:
                best[prec][1] = min(best[prec][1], timeit(g))
    for prec in precs:
        t_f, t_w = best[prec]
        msg = f'{prec:7s} {cin:4d}->{cout:4d} @{hw:3d}^2  fwd {t_f:7.3f} ms {flops/t_f/1e9:7.1f} TF'
        if t_w < 1e9:
            msg += f' | wgrad {t_w:7.3f} ms {flops/t_w/1e9:7.1f} TF'
        print(msg, flush=True)
