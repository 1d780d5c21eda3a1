"""A/B of the two f16c8 forward / data-gradient kernels at the real layer shapes (B=16), interleaved rounds in ONE
process (rule 24): library option c8_w4 = 0 -> conv_f16c8.hip (8 waves, 2x2 blocks), =1 -> conv_f16c8_w4.hip (4 waves, 4x4 blocks).
    python tools/w4_bench.py            fp32 input and record input, per layer: ms and algorithmic TFLOP/s"""
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

shapes = [(128, 128, 256), (256, 256, 128), (512, 512, 64), (256, 128, 256), (512, 256, 128), (512, 512, 32)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
B = 16
lib = L.load()
ops.CONV_PRECISION = 'f16c8'
for cin, cout, hw in shapes:
    x = torch.randn(B, cin, hw, hw, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (cin * 9) ** 0.5
    y = torch.empty(B, cout, hw, hw, device='cuda')
    res = torch.randn_like(y)
    gns = torch.zeros(B, 32, 2, dtype=torch.float64, device='cuda')
    bias = torch.randn(cout, device='cuda')
    amax = ops.absmax_bits(x)
    flops = 2.0 * B * hw * hw * cin * cout * 9
    wp = torch.empty(lib.eovae_conv3x3_f16c8_packed_bytes(cout, cin, 0), dtype=torch.uint8, device='cuda')
    L.check(lib.eovae_conv3x3_f16c8_pack(L.ptr(w), L.ptr(wp), cout, cin, 0, 1.0, L.stream()), 'pack')
    g1, g0 = torch.ones(cin, device='cuda'), torch.zeros(cin, device='cuda')
    with torch.no_grad():
        xrec = ops.group_norm_silu(x, g1, g0, 32, 1e-6, True, True)                  # format 1: f16 | e5m2 halves
        xmx = ops.group_norm_silu(x, g1, g0, 32, 1e-6, True, True, cout=cout)        # format 2 where the layer takes it
    have_mx = ops.packed_format(xmx) == 2
    # variant '2' = the w4 kernel on MX operand records: correction product on block-scaled FP6 operands (round 5);
    # variant '3' = the same records through conv_f16c8_q16.hip (16 x 16 MFMAs, all-DMA feed; library option c8_q16)
    xin = lambda v: (xmx, 2) if v in '23' else (xrec, 1)
    cases = {
        'fwd rec+gn+res': lambda v: lib.eovae_conv3x3_f16c8_fwd_mx(L.ptr(xin(v)[0]), None, L.ptr(wp), L.ptr(bias), L.ptr(res), L.ptr(y), None, B, cin, hw, hw, cout, hw, hw, 0, 0, xin(v)[1], L.ptr(gns), 32, None, L.stream()),
        'fwd rec+gn    ': lambda v: lib.eovae_conv3x3_f16c8_fwd_mx(L.ptr(xin(v)[0]), None, L.ptr(wp), L.ptr(bias), None, L.ptr(y), None, B, cin, hw, hw, cout, hw, hw, 0, 0, xin(v)[1], L.ptr(gns), 32, None, L.stream()),
        'dgrad fp32+amax': lambda v: lib.eovae_conv3x3_f16c8_fwd(L.ptr(x), L.ptr(wp), None, None, L.ptr(y), L.ptr(amax), B, cin, hw, hw, cout, hw, hw, 0, 0, 0, None, 0, L.stream()),
    }
    variants = '0123' if have_mx else '01'
    best = {(k, v): 1e9 for k in cases for v in variants}
    for rnd in range(3):
        for k, f in cases.items():
            for v in variants:
                if v == '2' and k.startswith('dgrad'):
                    continue
                lib.eovae_set_option(b'c8_w4', 0 if v == '0' else 1)
                lib.eovae_set_option(b'c8_q16', 2 if v == '3' else 0)
                best[(k, v)] = min(best[(k, v)], timeit(lambda: f(v)))
    for k in cases:
        t0, t1 = best[(k, '0')], best[(k, '1')]
        t2 = best.get((k, '2'), 1e9)
        t3 = best.get((k, '3'), 1e9)
        mx = f' | w4 mx-fp6 {t2:7.3f} ms {flops/t2/1e9:6.1f} TF x{t1/t2:.3f} of w4' if t2 < 1e8 else ''
        if t3 < 1e8:
            mx += f' | q16 {t3:7.3f} ms {flops/t3/1e9:6.1f} TF x{(t2 if t2 < 1e8 else t1)/t3:.3f} of {"mx" if t2 < 1e8 else "w4"}'
        print(f'{cin:4d}->{cout:4d} @{hw:3d}^2 {k}: 8-wave {t0:7.3f} ms {flops/t0/1e9:6.1f} TF | w4 {t1:7.3f} ms {flops/t1/1e9:6.1f} TF | x{t0/t1:.3f}{mx}', flush=True)
