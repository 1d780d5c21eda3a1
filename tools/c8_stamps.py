"""Stage timeline of the f16c8 convolution kernel from a diagnostic build
(SRC=conv_f16c8 tools/build_variant.sh stamp -DEOVAE_C8_STAMP=1):  per pipeline stage and wave, cycles from
stage start to: 1 issue done, 2 MFMAs done, 3 conversion done, 4-5 halo store, 6 own DMAs landed, 7 barrier released.
    EOVAE_LIB=eo_vae_amd/build/exp/lib_stamp.so python tools/c8_stamps.py [cin cout hw]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eo_vae_amd import lib as L
cin, cout, hw = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (128, 128, 256)
lib = L.load()
B = 16
x = torch.randn(B, cin, hw, hw, device='cuda')
w = torch.randn(cout, cin, 3, 3, device='cuda') / (cin * 9) ** 0.5
y = torch.empty(B, cout, hw, hw, device='cuda')
wp = torch.empty(lib.eovae_conv3x3_f16c8_packed_bytes(cout, cin, 0), dtype=torch.uint8, device='cuda')
L.check(lib.eovae_conv3x3_f16c8_pack(L.ptr(w), L.ptr(wp), cout, cin, 0, 1.0, L.stream()), 'pack')
for _ in range(20):   # warm: clocks settle under load
    L.check(lib.eovae_conv3x3_f16c8_fwd(L.ptr(x), L.ptr(wp), None, None, L.ptr(y), None, B, cin, hw, hw, cout, hw, hw, 0, 0, 0, None, 0, L.stream()), 'fwd')
torch.cuda.synchronize()
NPT, MAXREC, NBLK = 8, 96, 4
buf = np.zeros(NBLK * 8 * MAXREC * NPT, dtype=np.uint64)
fn = C.CDLL(L.LIB_PATH).eovae_debug_c8_stamps
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(NBLK, 8, MAXREC, NPT).astype(np.int64)
nchunk = cin // 32
per_tile = nchunk * 5 + 1
print(f'{cin}->{cout} @{hw}^2: {nchunk} chunks x 5 stages + epilogue record per tile; cycles (shader clock)')
for blk in (0, 1):
    for wave in (0, 4):
        print(f'block {blk} wave {wave} ({"DMA" if wave < 4 else "halo"} wave): stage, then deltas from stage start to points 1..7, and stage length')
        for rec in range(per_tile, min(3 * per_tile, MAXREC)):   # second and third tile of the block (steady state)
            r = t[blk, wave, rec]
            if r[0] == 0:
                continue
            kind = 'epi' if (rec % per_tile) == per_tile - 1 else f'c{(rec % per_tile) // 5}s{(rec % per_tile) % 5}'
            d = [(int(v - r[0]) if v else -1) for v in r[1:]]
            nxt = t[blk, wave, rec + 1, 0] if rec + 1 < MAXREC else 0
            print(f'  {kind:5s}', ' '.join(f'{v:6d}' for v in d), f'| to next stage start {int(nxt - r[0]) if nxt else -1:6d}')
print('all eight waves of block 1, records of the second tile: start offset vs wave 0, then points 1 2 3 6 7')
for rec in range(per_tile, 2 * per_tile):
    kind = 'epi' if (rec % per_tile) == per_tile - 1 else f'c{(rec % per_tile) // 5}s{(rec % per_tile) % 5}'
    if kind == 'epi':
        continue
    base = t[1, 0, rec, 0]
    row = []
    for wv in range(8):
        r = t[1, wv, rec]
        row.append(f'w{wv}:{int(r[0]-base):5d}|' + ' '.join(f'{int(r[k]-r[0]) if r[k] else -1:5d}' for k in (1, 2, 3, 6, 7)))
    print(f' {kind}', '  '.join(row))
tile0 = t[:, :, per_tile, 0]
tile1 = t[:, :, 2 * per_tile, 0]
for b in range(NBLK):   # shader clock held under this load: d(s_memtime) / d(s_memrealtime) x 100 MHz over three tiles
    last = min(4, MAXREC // per_tile)
    e0, e1 = t[b, 0, per_tile - 1], t[b, 0, last * per_tile - 1]
    if e0[1] and e1[1]:
        print(f'block {b}: shader clock {100.0 * (e1[7] - e0[7]) / (e1[1] - e0[1]):.0f} MHz')
print('cycles per tile (wave 0 of each block):', [int(tile1[b, 0] - tile0[b, 0]) for b in range(NBLK)])
