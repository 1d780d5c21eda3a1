import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from eo_vae_amd import ops, lib as L
from oracle import mx_emul as M
lib = L.load()
ops.CONV_PRECISION = 'f16c8'
lib.eovae_set_option(b'c8_w4', 1)
rel = lambda a_, b_: ((a_ - b_).abs().max() / b_.abs().max()).item()
for (n, c, cout, h, w) in [(1, 32, 128, 16, 32), (2, 32, 128, 32, 64), (2, 64, 128, 32, 64), (16, 128, 128, 64, 64)]:
    torch.manual_seed(5)
    x = torch.randn(n, c, h, w, device='cuda') * 2 + 0.3
    gamma, beta = torch.randn(c, device='cuda'), torch.randn(c, device='cuda')
    wt = torch.randn(cout, c, 3, 3, device='cuda') / (c * 9) ** 0.5
    def run(q16):
        lib.eovae_set_option(b'c8_q16', q16)
        with torch.no_grad():
            a = ops.group_norm_silu(x, gamma, beta, packed=True, cout=cout)
            y = ops.conv2d(a, wt, None, None, 'normal')
        torch.cuda.synchronize()
        return y.cpu().double()
    y16a, y16b, y32 = run(1), run(1), run(0)
    print((n, c, cout, h, w), 'q16 vs q32', f'{rel(y16a, y32):.3e}', 'q16 run-to-run equal', torch.equal(y16a, y16b), 'tiles', n * (h // 16) * (w // 32) * (cout // 128))
