"""WGRAD_BENCH_Q16=1: instead, the row-streaming kernel's two MFMA shapes (library option c8_q16 = 0: 32x32x16, 1: 16x16x32),
alternating, plain layers only.
A/B of the plain f16c8 weight-gradient kernels at the real layer shapes (B=16), interleaved rounds in one process,
through the product entry (eovae_conv3x3_f16c8_wgrad_dw: partial tiles + fold, record input):
option c8_wgrad_rows = 0 -> wgrad3x3_f16c8_kernel (tile kernel), 1 -> conv_f16c8_wgrad_rows.hip (row streaming)."""
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
if os.environ.get('WGRAD_BENCH_Q16') == '1':
    for cin, cout, hw in shapes:
        x = torch.randn(B, cin, hw, hw, device='cuda')
        dy = torch.randn(B, cout, hw, hw, device='cuda') * 1e-4
        dw = torch.zeros(cout, cin, 3, 3, device='cuda')
        db = torch.zeros(cout, device='cuda')
        amax = ops.absmax_bits(dy)
        flops = 2.0 * B * hw * hw * cin * cout * 9
        xrec = ops.group_norm_silu(x, torch.ones(cin, device='cuda'), torch.zeros(cin, device='cuda'), 32, 1e-6, True, True, cout=cout)
        fmt = ops.packed_format(xrec)
        ws = torch.empty(lib.eovae_conv3x3_f16c8_wgrad_dw_workspace(cin, cout) // 4, device='cuda')
        run = lambda: lib.eovae_conv3x3_f16c8_wgrad_dw(L.ptr(xrec), L.ptr(dy), L.ptr(dw), L.ptr(db), L.ptr(amax), None, B, cin, hw, hw,
                                                       cout, hw, hw, fmt, L.ptr(ws), L.stream())
        best, res = {0: 1e9, 1: 1e9}, {}
        for rnd in range(4):
            for v in (0, 1):
                lib.eovae_set_option(b'c8_q16', v)
                best[v] = min(best[v], timeit(run, 20))
                res[v] = dw.clone()
        err = ((res[1] - res[0]).abs().max() / res[0].abs().max()).item()
        t0, t1 = best[0], best[1]
        print(f'{cin:4d}->{cout:4d} @{hw:3d}^2 rows wgrad (records format {fmt}, incl. fold): 32x32x16 {t0:7.3f} ms {flops/t0/1e9:6.1f} TF | '
              f'16x16x32 {t1:7.3f} ms {flops/t1/1e9:6.1f} TF | x{t0/t1:.3f} | difference {err:.1e}', flush=True)
    sys.exit(0)
for cin, cout, hw in shapes:
    x = torch.randn(B, cin, hw, hw, device='cuda')
    dy = torch.randn(B, cout, hw, hw, device='cuda') * 1e-4
    dw = torch.zeros(cout, cin, 3, 3, device='cuda')
    db = torch.zeros(cout, device='cuda')
    amax = ops.absmax_bits(dy)
    flops = 2.0 * B * hw * hw * cin * cout * 9
    xrec = ops.group_norm_silu(x, torch.ones(cin, device='cuda'), torch.zeros(cin, device='cuda'), 32, 1e-6, True, True).detach()
    ws = torch.empty(lib.eovae_conv3x3_f16c8_wgrad_dw_workspace(cin, cout) // 4, device='cuda')
    run = lambda: lib.eovae_conv3x3_f16c8_wgrad_dw(L.ptr(xrec), L.ptr(dy), L.ptr(dw), L.ptr(db), L.ptr(amax), None, B, cin, hw, hw,
                                                   cout, hw, hw, 1, L.ptr(ws), L.stream())
    best = {0: 1e9, 1: 1e9}
    res = {}
    for rnd in range(3):
        for v in (0, 1):
            lib.eovae_set_option(b'c8_wgrad_rows', v)
            best[v] = min(best[v], timeit(run))
            res[v] = dw.clone()
    err = ((res[1] - res[0]).abs().max() / res[0].abs().max()).item()
    t0, t1 = best[0], best[1]
    print(f'{cin:4d}->{cout:4d} @{hw:3d}^2 wgrad (records, incl. fold): tile {t0:7.3f} ms {flops/t0/1e9:6.1f} TF | rows {t1:7.3f} ms '
          f'{flops/t1/1e9:6.1f} TF | x{t0/t1:.3f} | rows vs tile {err:.1e}', flush=True)

# x2-upsample layers (low-res shapes): tile kernel's sub-pixel mode vs the sub-pixel row-streaming form
if os.environ.get('WGRAD_BENCH_UP', '1') != '0':
    for cin, cout, hw in [(512, 512, 32), (512, 512, 64), (256, 256, 128)]:
        x = torch.randn(B, cin, hw, hw, device='cuda') * 2
        dy = torch.randn(B, cout, 2 * hw, 2 * hw, device='cuda') * 1e-4
        w = torch.randn(cout, cin, 3, 3, device='cuda')
        flops = 2.0 * B * 4 * hw * hw * cin * cout * 9
        best, res = {False: 1e9, True: 1e9}, {}
        for rnd in range(3):
            for v in (False, True):
                ops.UP_WGRAD_ROWS = v
                run = lambda: ops.up_conv_subpix_wgrad(x, dy, w, torch.zeros(cout, device='cuda'))
                best[v] = min(best[v], timeit(run, 5))
                res[v] = run().clone()
        err = ((res[True] - res[False]).abs().max() / res[False].abs().max()).item()
        t0, t1 = best[False], best[True]
        print(f'up {cin:4d}->{cout:4d} @{hw:3d}^2 -> {2*hw}^2 wgrad (incl. amax, records, fold): tile {t0:7.3f} ms {flops/t0/1e9:6.1f} TF alg | '
              f'rows {t1:7.3f} ms {flops/t1/1e9:6.1f} TF alg ({flops*4/9/t1/1e9:5.1f} actual) | x{t0/t1:.3f} | rows vs tile {err:.1e}', flush=True)

# stride-2 layers (hi-res input shapes): tile kernel's space-to-depth mode vs the swapped-operand row-streaming form
if os.environ.get('WGRAD_BENCH_DOWN', '1') != '0':
    for cin, cout, hw in [(128, 128, 256), (256, 256, 128), (512, 512, 64)]:
        x = torch.randn(B, cin, hw, hw, device='cuda') * 2
        dy = torch.randn(B, cout, hw // 2, hw // 2, device='cuda') * 1e-4
        w = torch.randn(cout, cin, 3, 3, device='cuda')
        flops = 2.0 * B * (hw // 2) * (hw // 2) * cin * cout * 9
        best, res = {False: 1e9, True: 1e9}, {}
        for rnd in range(3):
            for v in (False, True):
                ops.UP_WGRAD_ROWS = v
                run = lambda: ops.down_conv_s2d_wgrad(x, dy, w, torch.zeros(cout, device='cuda'))
                best[v] = min(best[v], timeit(run, 5))
                res[v] = run().clone()
        err = ((res[True] - res[False]).abs().max() / res[False].abs().max()).item()
        t0, t1 = best[False], best[True]
        print(f'down {cin:4d}->{cout:4d} @{hw:3d}^2 -> {hw//2}^2 wgrad (incl. amax, records, fold): tile {t0:7.3f} ms {flops/t0/1e9:6.1f} TF | '
              f'rows {t1:7.3f} ms {flops/t1/1e9:6.1f} TF | x{t0/t1:.3f} | rows vs tile {err:.1e}', flush=True)
