"""Weight-generator kernels (csrc/generator.hip through eo_vae_amd.generator_ops) against fp64 PyTorch on the CPU:
the strided GEMM with its epilogues and split-K, the Linear function, and one whole post-norm encoder layer
(reference configuration: modules/dynamic_conv.py:86-101) -- output, input gradient and all twelve parameter gradients,
for the full layer and for the last-layer form that only computes the rows the generator reads."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# exact-fp32 FMA arithmetic: the bar is fp32 rounding of <= 2048-term sums (the library path this replaces sat at the same level)
TOL = 2e-5


def rel(a, b):
    b = b.double()
    return ((a.double().cpu() - b.cpu()).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize('M,N,K,ksplit', [(132, 256, 256, 1), (141, 768, 256, 1), (5, 2048, 256, 1), (132, 256, 2048, 6),
                                          (33, 70, 19, 1), (4, 256, 256, 4)])
def test_gemm_plain_and_split(M, N, K, ksplit):
    from eo_vae_amd import generator_ops as G
    g = torch.Generator().manual_seed(M * 7 + N)
    a, b, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    A, B, bi = a.cuda(), b.cuda(), bias.cuda()
    want = a.double() @ b.double().t()
    if ksplit == 1:
        c = torch.empty(M, N, device='cuda')
        G.gemm(A, 0, K, 1, B, 0, K, 1, c, 0, N, M, N, K, bias=bi, alpha=0.5)
        assert rel(c, 0.5 * want + bias.double()) < TOL
        # transposed operands through the strides: A^T stored [K, M], B^T stored [K, N]
        At, Bt = A.t().contiguous(), B.t().contiguous()
        c2 = torch.empty(M, N, device='cuda')
        G.gemm(At, 0, 1, M, Bt, 0, 1, N, c2, 0, N, M, N, K)
        assert rel(c2, want) < TOL
        # accumulate epilogue
        G.gemm(A, 0, K, 1, B, 0, K, 1, c2, 0, N, M, N, K, epi=3)
        assert rel(c2, 2 * want) < TOL
    else:
        parts = torch.full((ksplit, M, N), float('nan'), device='cuda')
        G.gemm(A, 0, K, 1, B, 0, K, 1, parts, 0, N, M, N, K, ksplit=ksplit, part=M * N)
        assert rel(parts.sum(0), want) < TOL


def test_gemm_gelu_epilogues_and_batches():
    from eo_vae_amd import generator_ops as G
    g = torch.Generator().manual_seed(5)
    M, N, K = 37, 200, 64
    a, b, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / 8, torch.randn(N, generator=g)
    A, B, bi = a.cuda(), b.cuda(), bias.cuda()
    pre, act = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
    G.gemm(A, 0, K, 1, B, 0, K, 1, act, 0, N, M, N, K, bias=bi, aux_out=pre, epi=1)
    want_pre = a.double() @ b.double().t() + bias.double()
    assert rel(pre, want_pre) < TOL and rel(act, F.gelu(want_pre)) < TOL
    xg = want_pre.clone().requires_grad_(True)
    F.gelu(xg).sum().backward()
    out = torch.empty(M, N, device='cuda')
    G.gemm(A, 0, K, 1, B, 0, K, 1, out, 0, N, M, N, K, aux=pre, epi=2)
    assert rel(out, (a.double() @ b.double().t()) * xg.grad) < TOL
    # batched: 4 heads cut out of one [L, 3d] matrix, as the attention scores are
    L_, d, h = 21, 32, 4
    qkv = torch.randn(L_, 3 * d, generator=g)
    Q = qkv.cuda()
    s = torch.empty(h, L_, L_, device='cuda')
    G.gemm(Q, 0, 3 * d, 1, Q, d, 3 * d, 1, s, 0, L_, L_, L_, d // h, batch=h, batA=d // h, batB=d // h, batC=L_ * L_, alpha=0.25)
    q, k = qkv[:, :d].view(L_, h, -1).transpose(0, 1).double(), qkv[:, d:2 * d].view(L_, h, -1).transpose(0, 1).double()
    assert rel(s, 0.25 * q @ k.transpose(1, 2)) < TOL


@pytest.mark.parametrize('shape', [(3, 128), (128,), (7, 256)])
def test_linear_function(shape):
    from eo_vae_amd import generator_ops as G
    torch.manual_seed(1)
    lin = nn.Linear(shape[-1], 200).double()
    x = torch.randn(*shape, dtype=torch.float64, requires_grad=True)
    y = lin(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    xc = x.detach().float().cuda().requires_grad_(True)
    w, b = lin.weight.detach().float().cuda().requires_grad_(True), lin.bias.detach().float().cuda().requires_grad_(True)
    yc = G.linear(xc, w, b)
    yc.backward(gy.float().cuda())
    assert yc.shape == y.shape
    assert rel(yc, y) < TOL and rel(xc.grad, x.grad) < TOL and rel(w.grad, lin.weight.grad) < TOL and rel(b.grad, lin.bias.grad) < TOL


@pytest.mark.parametrize('seq,keep_from', [(132, 0), (132, 128), (141, 0), (141, 128), (131, 128)])
def test_encoder_layer_matches_torch_fp64(seq, keep_from):
    from eo_vae_amd import generator_ops as G
    torch.manual_seed(seq + keep_from)
    d = 256
    ref = nn.TransformerEncoderLayer(d_model=d, nhead=4, activation='gelu', norm_first=False, batch_first=False, dropout=0.0)
    for p in ref.parameters():          # non-trivial LayerNorm weights / biases
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    ref = ref.double()
    x = torch.randn(seq, d, dtype=torch.float64, requires_grad=True)
    y = ref(x)[keep_from:]
    gy = torch.randn_like(y)
    y.backward(gy)
    import copy
    prod = copy.deepcopy(ref).float().cuda()
    for p in prod.parameters():
        p.grad = None
    xc = x.detach().float().cuda().requires_grad_(True)
    yc = G.encoder_layer(prod, xc, keep_from)
    yc.backward(gy.float().cuda())
    assert rel(yc, y) < TOL
    assert rel(xc.grad, x.grad) < 5 * TOL
    for (n, pr), (_, pp) in zip(ref.named_parameters(), prod.named_parameters()):
        assert pp.grad is not None, n
        assert rel(pp.grad, pr.grad) < 5 * TOL, n


def test_generators_on_kernels_match_the_torch_formulation(monkeypatch):
    """Both dynamic convolutions' generated (weight, bias) and the generator gradients: kernels vs the PyTorch-op path."""
    from eo_vae_amd import dynamic_conv as DC
    for cls, layers in ((DC.DynamicConv, 4), (DC.DynamicConv_decoder, 2)):
        torch.manual_seed(11)
        m = cls(wv_planes=256, embed_dim=128, num_layers=layers).cuda().train()
        wvs = torch.tensor([0.443, 0.49, 0.56, 0.665, 0.705, 0.74, 0.783, 0.842, 0.865, 0.945, 1.61, 2.19], device='cuda')
        outs = {}
        for flag in (False, True):
            monkeypatch.setattr(DC, 'GENERATOR_KERNELS', flag)
            m.zero_grad(set_to_none=True)
            w, b = m.get_distillation_weight(wvs)
            (w.square().sum() + b.square().sum()).backward()
            outs[flag] = (w.detach().clone(), b.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()})
        assert rel(outs[True][0], outs[False][0]) < 1e-4 and rel(outs[True][1], outs[False][1]) < 1e-4
        for n, gk in outs[True][2].items():
            assert rel(gk, outs[False][2][n]) < 2e-3, n     # fp32 vs fp32 through 2-4 layers: summation-order level
