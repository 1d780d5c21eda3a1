"""Per-tile timeline of conv_f16c8_q16.hip from a diagnostic build
(SRC=conv_f16c8_q16 tools/build_variant.sh q16stamp -DEOVAE_Q16_STAMP=1): cycles of the phase sequence and of the epilogue per
tile of block 0, and the shader clock held.
    EOVAE_LIB=eo_vae_amd/build/exp/lib_q16stamp.so python tools/q16_stamps.py [cin cout hw [res]]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eo_vae_amd import ops, lib as L
cin, cout, hw = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (128, 128, 256)
use_res = 'res' in sys.argv[4:]
use_fp32 = 'fp32' in sys.argv[4:]   # the data-gradient form: fp32 input quantised in the kernel
ops.CONV_PRECISION = 'f16c8'
lib = L.load()
lib.eovae_set_option(b'c8_w4', 1)
B = int(os.environ.get('STAMP_B', '16'))
x = torch.randn(B, cin, hw, hw, device='cuda')
w = torch.randn(cout, cin, 3, 3, device='cuda') / (cin * 9) ** 0.5
y = torch.empty(B, cout, hw, hw, device='cuda')
res = torch.randn_like(y) if use_res else None
gns = torch.zeros(B, 32, 2, dtype=torch.float64, device='cuda')
xrec = ops.group_norm_silu(x, torch.ones(cin, device='cuda'), torch.zeros(cin, device='cuda'), 32, 1e-6, True, True, cout=cout)
assert ops.packed_format(xrec) == 2
wp = torch.empty(lib.eovae_conv3x3_f16c8_packed_bytes(cout, cin, 0), dtype=torch.uint8, device='cuda')
L.check(lib.eovae_conv3x3_f16c8_pack(L.ptr(w), L.ptr(wp), cout, cin, 0, 1.0, L.stream()), 'pack')
amax = ops.absmax_bits(x)
for _ in range(30):
    if use_fp32:
        L.check(lib.eovae_conv3x3_f16c8_fwd(L.ptr(x), L.ptr(wp), None, None, L.ptr(y), L.ptr(amax), B, cin, hw, hw, cout, hw, hw, 0, 0, 0, None, 0, L.stream()), 'fwd')
        continue
    L.check(lib.eovae_conv3x3_f16c8_fwd_mx(L.ptr(xrec), None, L.ptr(wp), None, L.ptr(res), L.ptr(y), None, B, cin, hw, hw, cout, hw, hw, 0, 0, 2,
                                           L.ptr(gns), 32, None, L.stream()), 'fwd')
torch.cuda.synchronize()
buf = np.zeros(4 * 32 * 4, dtype=np.uint64)
fn = C.CDLL(L.LIB_PATH).eovae_debug_q16_stamps
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(4, 32, 4).astype(np.int64)
nchunk = cin // 32
print(f'{cin}->{cout} @{hw}^2 res={use_res}: {nchunk} chunks x 14 phases x 1024 = {nchunk * 14 * 1024} MFMA cycles per tile')
for wave in (0, 3):
    print(f'wave {wave}: per tile (phases, epilogue | tile to tile)')
    n = 0
    for rec in range(32):
        r = t[wave, rec]
        if r[0] == 0:
            break
        n += 1
        nxt = t[wave, rec + 1, 0] if rec + 1 < 32 and t[wave, rec + 1, 0] else 0
        print('   ', f'{int(r[1] - r[0]):7d} {int(r[2] - r[1]):7d} | {int(nxt - r[0]) if nxt else -1:7d}')
    if n >= 3:
        c = 100.0 * (t[wave, n - 1, 0] - t[wave, 1, 0]) / (t[wave, n - 1, 3] - t[wave, 1, 3])
        print(f'wave {wave}: shader clock {c:.0f} MHz')

try:
    fp = C.CDLL(L.LIB_PATH).eovae_debug_q16_phase_stamps
    fp.argtypes = [C.c_void_p, C.c_size_t]
    pb = np.zeros(4 * 14 * 4, dtype=np.uint64)
    if fp(pb.ctypes.data, pb.nbytes) == 0 and pb.any():
        q = pb.reshape(4, 14, 4).astype(np.int64)
        for wave in (0, 3):
            print(f'wave {wave}, second tile, chunk 1 -- per phase: wait at the barrier | MFMA block | wait for the requests | phase total')
            for p in range(14):
                r = q[wave, p]
                nxt = q[wave, p + 1, 0] if p + 1 < 14 else r[3]
                print(f'   phase {p:2d}: {int(r[1] - r[0]):5d} | {int(r[2] - r[1]):5d} | {int(r[3] - r[2]):5d} | {int(nxt - r[0]):5d}')
except AttributeError:
    pass
