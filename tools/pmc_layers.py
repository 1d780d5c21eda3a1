"""Run the two dominant kernels once each at 3 layer shapes (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eo_vae_amd import ops, lib as L
lib = L.load()
B = 16
for cin, cout, hw in [(128, 128, 256), (512, 512, 64), (512, 512, 32)]:
    x = torch.randn(B, cin, hw, hw, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (cin * 9) ** 0.5
    dy = torch.randn(B, cout, hw, hw, device='cuda')
    wp = torch.empty(lib.eovae_conv3x3_bf16_packed_elems(cout, cin, 0), dtype=torch.bfloat16, device='cuda')
    lib.eovae_conv3x3_bf16_pack(L.ptr(w), L.ptr(wp), cout, cin, 0, 1.0, L.stream())
    y = torch.empty_like(dy)
    dwp = torch.zeros(9 * cout * cin, device='cuda')
    for _ in range(2):
        lib.eovae_conv3x3_bf16_fwd(L.ptr(x), L.ptr(wp), None, None, L.ptr(y), B, cin, hw, hw, cout, hw, hw, 0, 3, 0, L.stream())
        lib.eovae_conv3x3_bf16_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dwp), None, B, cin, hw, hw, cout, hw, hw, 0, 3, 0, L.stream())
    torch.cuda.synchronize()
