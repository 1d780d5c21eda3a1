"""Per-tile timeline of the one-wave-per-SIMD convolution kernel from a diagnostic build
(SRC=conv_f16c8_w4 tools/build_variant.sh w4stamp -DEOVAE_W4_STAMP=1): cycles of the phase sequence, the epilogue,
the accumulator reset and the GroupNorm flush of each tile, and the shader clock held.
    EOVAE_LIB=eo_vae_amd/build/exp/lib_w4stamp.so python tools/w4_stamps.py [cin cout hw [res] [mx]]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eo_vae_amd import ops, lib as L
cin, cout, hw = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (128, 128, 256)
use_res = 'res' in sys.argv[4:]
use_mx = 'mx' in sys.argv[4:]     # MX operand records (format 2): the FP6 correction product
os.environ['EOVAE_C8_W4'] = '1'
ops.CONV_PRECISION = 'f16c8'
lib = L.load()
B = 16
x = torch.randn(B, cin, hw, hw, device='cuda')
w = torch.randn(cout, cin, 3, 3, device='cuda') / (cin * 9) ** 0.5
y = torch.empty(B, cout, hw, hw, device='cuda')
res = torch.randn_like(y) if use_res else None
gns = torch.zeros(B, 32, 2, dtype=torch.float64, device='cuda')
xrec = ops.group_norm_silu(x, torch.ones(cin, device='cuda'), torch.zeros(cin, device='cuda'), 32, 1e-6, True, True,
                           cout=cout if use_mx else None)
fmt = ops.packed_format(xrec)
assert fmt == (2 if use_mx else 1), fmt
wp = torch.empty(lib.eovae_conv3x3_f16c8_packed_bytes(cout, cin, 0), dtype=torch.uint8, device='cuda')
L.check(lib.eovae_conv3x3_f16c8_pack(L.ptr(w), L.ptr(wp), cout, cin, 0, 1.0, L.stream()), 'pack')
for _ in range(30):   # warm: clocks settle under load
    L.check(lib.eovae_conv3x3_f16c8_fwd_mx(L.ptr(xrec), None, L.ptr(wp), None, L.ptr(res), L.ptr(y), None, B, cin, hw, hw, cout, hw, hw, 0, 0, fmt,
                                           L.ptr(gns), 32, None, L.stream()), 'fwd')
torch.cuda.synchronize()
NPT, MAXREC, NBLK = 7, 32, 8
buf = np.zeros(NBLK * 4 * MAXREC * NPT, dtype=np.uint64)
fn = C.CDLL(L.LIB_PATH).eovae_debug_w4_stamps
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(NBLK, 4, MAXREC, NPT).astype(np.int64)
nchunk = cin // 32
print(f'{cin}->{cout} @{hw}^2 res={use_res}: {nchunk} chunks x 18 phases = {nchunk * 18 * 1024} MFMA cycles per tile')
t00 = int(t[0, 0, 0, 0])
for blk in range(NBLK):
    hw = int(t[blk, 0, 0, 6])
    n = sum(1 for rec in range(MAXREC) if t[blk, 0, rec, 0])
    print(f'slot {blk} (block {(blk >> 2) * 256 + (blk & 3)}): first tile starts at {int(t[blk, 0, 0, 0]) - t00:9d}, last tile ends at {int(t[blk, 0, n - 1, 4]) - t00 if n else -1:9d} cycles after block 0 started; {n} tiles')
    print(f'block {blk}: HW_ID {hw:#010x}  wave slot {hw & 15} simd {(hw >> 4) & 3} cu {(hw >> 8) & 15} sh {(hw >> 12) & 1} se {(hw >> 13) & 7} tg {(hw >> 16) & 15}')
for blk in range(0, NBLK, 2):
    for wave in (0, 3):
        rows = []
        for rec in range(MAXREC):
            r = t[blk, wave, rec]
            if r[0] == 0:
                break
            nxt = t[blk, wave, rec + 1, 0] if rec + 1 < MAXREC and t[blk, wave, rec + 1, 0] else 0
            rows.append((int(r[1] - r[0]), int(r[2] - r[1]), int(r[3] - r[2]), int(r[4] - r[3]), int(nxt - r[0]) if nxt else -1))
        print(f'block {blk} wave {wave}: per tile (phases, epilogue, zero, gn flush | tile to tile):')
        for row in rows:
            print('   ', ' '.join(f'{v:7d}' for v in row[:4]), '|', f'{row[4]:7d}')
    n = sum(1 for rec in range(MAXREC) if t[blk, 0, rec, 0])
    if n >= 3:
        c = 100.0 * (t[blk, 0, n - 1, 0] - t[blk, 0, 1, 0]) / (t[blk, 0, n - 1, 5] - t[blk, 0, 1, 5])
        print(f'block {blk}: shader clock {c:.0f} MHz')
