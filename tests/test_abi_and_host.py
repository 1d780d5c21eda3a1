"""CPU: C-ABI surface, host-side logic, drop-in import paths (no GPU compute)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'eovae_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(eovae_\w+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from eo_vae_amd import lib as L
    lib_path = L.LIB_PATH
    if not os.path.exists(lib_path):
        from eo_vae_amd import build
        build.build(verbose=False)
    cdll = ctypes.CDLL(lib_path)
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(cdll, s), f'{s} declared in include/eovae_hip.h but not exported'
    # the ctypes table mirrors the header one to one
    assert sorted(L.SIGNATURES) == syms
    assert cdll.eovae_abi_version() == L.ABI_VERSION == 12
    cdll.eovae_target_arch.restype = ctypes.c_char_p
    assert cdll.eovae_target_arch() == b'gfx950'


def test_argument_errors_are_reported_without_a_gpu():
    from eo_vae_amd import lib as L
    lib = L.load()
    kp, mp = ctypes.c_int(), ctypes.c_int()
    assert lib.eovae_conv_packed_dims(128, 3, 3, 0, 0, ctypes.byref(kp), ctypes.byref(mp)) == 0
    assert (kp.value, mp.value) == (8, 128)
    assert lib.eovae_conv_packed_dims(128, 3, 5, 0, 0, ctypes.byref(kp), ctypes.byref(mp)) != 0
    assert b'kernel size 5' in lib.eovae_last_error()
    with pytest.raises(L.EovaeError):
        L.check(-1, 'probe')
    # round-2 arguments of the f16c8 entries are validated before anything touches the device (fake pointers)
    fp = ctypes.c_void_p(4096)
    conv = lambda cout, subpix, xpacked, xamax, sums, groups: lib.eovae_conv3x3_f16c8_fwd(  # noqa: E731
        fp, fp, None, None, fp, xamax, 1, 64, 8, 8, cout, 8, 8, 0, subpix, xpacked, sums, groups, None)
    assert conv(96, 0, 0, None, fp, 32) != 0 and b'gn_sums needs' in lib.eovae_last_error()      # 3 channels per group
    assert conv(128, 0, 0, None, fp, 0) != 0 and b'gn_sums needs' in lib.eovae_last_error()
    assert conv(128, 0, 1, fp, None, 0) != 0 and b'packed input' in lib.eovae_last_error()        # records + pre-scale
    assert conv(128, 1, 1, None, None, 0) != 0 and b'packed input' in lib.eovae_last_error()      # records in a sub-pixel form
    assert lib.eovae_conv3x3_f16c8_fwd(fp, fp, None, None, fp, None, 1, 48, 8, 8, 128, 8, 8, 0, 0, 0, None, 0,
                                       None) != 0 and b'multiple of 32' in lib.eovae_last_error()
    assert lib.eovae_conv3x3_f16c8_wgrad(fp, fp, fp, None, None, None, 1, 96, 8, 8, 128, 8, 8, 0, 0, 1, None) != 0   # records need Cin % 64
    assert lib.eovae_conv3x3_f16c8_wgrad(fp, fp, fp, None, None, fp, 1, 128, 8, 8, 128, 8, 8, 0, 0, 1, None) != 0   # records carry no x scale
    assert lib.eovae_gn_silu_fwd_packed(fp, fp, fp, fp, fp, None, 1, 100, 64, 25, 1e-6, 1, fp, None) != 0 \
        and b'gn_silu_fwd_packed' in lib.eovae_last_error()                                          # C % 8


def test_product_fails_loudly_on_cpu():
    from eo_vae_amd import ops
    from eo_vae_amd.lib import EovaeError
    with pytest.raises(EovaeError, match='no CPU fallback'):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3))
    with pytest.raises(EovaeError):
        ops.group_norm_silu(torch.zeros(1, 32, 4, 4), torch.ones(32), torch.zeros(32))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'eo_vae_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in re.sub(r'#.*', '', src), f'{f} references oracle/'


def test_dropin_import_paths_and_state_dict_contract(golden):
    import eo_vae.models as M
    from eo_vae.models.modules.consistency_loss import EOConsistencyLoss
    from eo_vae.models.modules.distributions import DiagonalGaussianDistribution  # noqa: F401
    from eo_vae.models.new_autoencoder import EOFluxVAE
    assert M.EOFluxVAE is EOFluxVAE
    dyn = {'num_layers': 4, 'wv_planes': 256}
    enc = M.Encoder(resolution=256, in_channels=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
                    z_channels=32, use_dynamic_ops=True, dynamic_conv_kwargs=dyn)
    dec = M.Decoder(ch=128, out_ch=3, ch_mult=[1, 2, 4, 4], num_res_blocks=2, resolution=256,
                    z_channels=32, use_dynamic_ops=True, dynamic_conv_kwargs=dyn)
    vae = EOFluxVAE(enc, dec, EOConsistencyLoss(), freeze_body=False)
    g = golden('full_cfg1')
    keys = [k for k in vae.state_dict().keys() if not k.startswith('loss_fn')]
    assert keys == [str(k) for k in g['keys']]          # the reference's 363 keys, same order
    n = sum(p.numel() for p in enc.parameters()) + sum(p.numel() for p in dec.parameters())
    assert n == 95514145
    with pytest.raises(ImportError):
        M.AutoencoderKL
    # freeze_body default True in the constructor, False through from_config (quirk q9)
    frozen = EOFluxVAE(enc, dec, torch.nn.Identity())
    assert not any(p.requires_grad for p in frozen.encoder.mid.parameters())
    assert all(p.requires_grad for p in frozen.encoder.conv_in.parameters())


def test_config_loader_and_from_config(tmp_path):
    from eo_vae_amd.autoencoder import EOFluxVAE
    from eo_vae_amd.config import instantiate, load_config
    cfg = load_config(os.path.join(ROOT, 'configs', 'eo-vae-mi355x.yaml'))
    assert cfg['model']['decay_end_epoch'] == cfg['trainer']['max_epochs'] == 100   # ${trainer.max_epochs}
    assert isinstance(cfg['model']['base_lr'], float) and cfg['model']['base_lr'] == 1e-4
    loss = instantiate(cfg['model']['loss_fn'])
    assert loss.rec_loss_type == 'char' and loss.starts['msssim'] == 2000
    dm = instantiate(cfg['datamodule'])
    assert dm.batch_size == 16
    tiny = tmp_path / 'tiny.yaml'
    tiny.write_text('''
model:
  base_lr: 3e-4
  clip_grad: 1.0
  encoder: {_target_: eo_vae.models.Encoder, z_channels: 32, resolution: 64, in_channels: 3, ch: 32,
            ch_mult: [1, 2], num_res_blocks: 1, use_dynamic_ops: true,
            dynamic_conv_kwargs: {num_layers: 1, wv_planes: 64}}
  decoder: {_target_: eo_vae.models.Decoder, z_channels: 32, resolution: 64, out_ch: 3, ch: 32,
            ch_mult: [1, 2], num_res_blocks: 1, use_dynamic_ops: true,
            dynamic_conv_kwargs: {num_layers: 1, wv_planes: 64}}
''')
    m = EOFluxVAE.from_config(str(tiny))
    assert m.base_lr == 3e-4 and m.clip_grad == 1.0 and not m.training and m.freeze_body is False
    with pytest.raises(FileNotFoundError):
        EOFluxVAE.from_config(str(tmp_path / 'missing.yaml'))
    bad = tmp_path / 'bad.yaml'
    bad.write_text('model: {encoder: {}}\n')
    with pytest.raises(ValueError):
        EOFluxVAE.from_config(str(bad))
    # checkpoint round trip + critical-missing-key error (reference :382-407)
    ck = tmp_path / 'w.ckpt'
    torch.save({'state_dict': m.state_dict()}, ck)
    m2 = EOFluxVAE.from_config(str(tiny), str(ck))
    assert torch.equal(m2.encoder.mid.attn_1.q.weight, m.encoder.mid.attn_1.q.weight)
    sd = {k: v for k, v in m.state_dict().items() if 'mid.attn_1' not in k}
    torch.save(sd, ck)
    with pytest.raises(RuntimeError, match='Critical weights missing'):
        EOFluxVAE.from_config(str(tiny), str(ck))
    m._load_checkpoint(str(tmp_path / 'nope.ckpt'), [])   # missing file: prints and returns


def test_lr_schedule_matches_oracle():
    from oracle import eo_vae_cpu as O
    from eo_vae_amd.autoencoder import get_cosine_schedule_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-4)
    sch = get_cosine_schedule_with_warmup(opt, 2000, 200000, 1e-4, 2e-5)
    for step in [0, 1, 1999, 2000, 2001, 100000, 200000]:
        sch.last_epoch = step - 1
        opt.step()
        sch.step()
        want = 1e-4 * O.lr_factor(step, 2000, 200000, 1e-4, 2e-5)
        assert abs(opt.param_groups[0]['lr'] - want) < 1e-15
    assert O.lr_factor(0, 2000, 200000, 1e-4, 2e-5) == 0.0     # quirk q6


def test_pixel_shuffle_helpers_match_einops():
    from einops import rearrange
    from eo_vae_amd.autoencoder import _pack2x2, _unpack2x2
    z = torch.randn(2, 5, 6, 8)
    p = rearrange(z, '... c (i pi) (j pj) -> ... (c pi pj) i j', pi=2, pj=2)
    assert torch.equal(_pack2x2(z), p)
    assert torch.equal(_unpack2x2(p), z)
    z5 = torch.randn(3, 2, 5, 6, 8)
    assert torch.equal(_pack2x2(z5), rearrange(z5, '... c (i pi) (j pj) -> ... (c pi pj) i j', pi=2, pj=2))


def test_generator_cache_is_keyed_on_wavelength_values(monkeypatch):
    """alternating two same-length wavelength sets through FRESH tensors in eval /
    no_grad must never return the other set's kernels (a recycled address used to hit a stale entry),
    and the same tensor object passed again must hit without being recomputed."""
    import gc
    from eo_vae_amd import dynamic_conv
    from eo_vae_amd.dynamic_conv import DynamicConv
    monkeypatch.setattr(dynamic_conv, 'GENERATOR_KERNELS', False)   # host logic of the cache on CPU tensors (no HIP here)
    torch.manual_seed(3)
    m = DynamicConv(wv_planes=64, embed_dim=32, num_layers=1).eval()
    sets = [[0.665, 0.56, 0.49], [0.842, 1.61, 2.19]]
    with torch.no_grad():
        m._gen_cache.clear()
        want = []
        for vals in sets:
            m.train()
            want.append([t.clone() for t in m._raw(torch.tensor(vals))])   # uncached (training mode)
            m.eval()
        wrong = 0
        for i in range(200):
            w = torch.tensor(sets[i % 2])     # fresh tensor every time: addresses get recycled
            got = m._raw(w)
            wrong += int(not all(torch.equal(a, b) for a, b in zip(got, want[i % 2])))
            del w, got
            if i % 16 == 0:
                gc.collect()
        assert wrong == 0
        # identity fast path + in-place edit of a cached tensor object
        w = torch.tensor(sets[0])
        a = m._raw(w)
        assert m._raw(w)[0] is a[0]
        w.copy_(torch.tensor(sets[1]))
        assert torch.equal(m._raw(w)[0], want[1][0])


def test_from_pretrained_with_stubbed_hub(tmp_path, monkeypatch):
    """from_pretrained (reference new_autoencoder.py:223-268): two hf_hub_download calls (config, checkpoint)
    with the caller's repo / revision / cache arguments, then from_config on the two paths; ImportError when
    huggingface_hub is absent."""
    import builtins
    import sys
    import types
    from eo_vae_amd.autoencoder import EOFluxVAE
    cfg = tmp_path / 'model_config.yaml'
    cfg.write_text("""
model:
  encoder: {_target_: eo_vae.models.Encoder, z_channels: 32, resolution: 64, in_channels: 3, ch: 32,
            ch_mult: [1, 2], num_res_blocks: 1, use_dynamic_ops: true,
            dynamic_conv_kwargs: {num_layers: 1, wv_planes: 64}}
  decoder: {_target_: eo_vae.models.Decoder, z_channels: 32, resolution: 64, out_ch: 3, ch: 32,
            ch_mult: [1, 2], num_res_blocks: 1, use_dynamic_ops: true,
            dynamic_conv_kwargs: {num_layers: 1, wv_planes: 64}}
""")
    src = EOFluxVAE.from_config(str(cfg))
    with torch.no_grad():
        src.encoder.mid.attn_1.q.weight.fill_(0.125)
    torch.save({'state_dict': src.state_dict()}, tmp_path / 'eo-vae.ckpt')
    calls = []

    def fake_download(repo_id, filename, revision=None, cache_dir=None, local_files_only=False, **kw):
        calls.append((repo_id, filename, revision, cache_dir, local_files_only))
        return str(tmp_path / filename)
    stub = types.ModuleType('huggingface_hub')
    stub.hf_hub_download = fake_download
    monkeypatch.setitem(sys.modules, 'huggingface_hub', stub)
    m = EOFluxVAE.from_pretrained('nilsleh/eo-vae', revision='abc', cache_dir=str(tmp_path), local_files_only=True)
    assert [c[1] for c in calls] == ['model_config.yaml', 'eo-vae.ckpt']
    assert all(c[0] == 'nilsleh/eo-vae' and c[2] == 'abc' and c[3] == str(tmp_path) and c[4] is True for c in calls)
    assert not m.training and m.freeze_body is False
    assert torch.all(m.encoder.mid.attn_1.q.weight == 0.125)
    # without the package: ImportError, as the reference (:240-243)
    monkeypatch.delitem(sys.modules, 'huggingface_hub')
    real_import = builtins.__import__

    def no_hub(name, *a, **k):
        if name == 'huggingface_hub':
            raise ImportError('no hub')
        return real_import(name, *a, **k)
    monkeypatch.setattr(builtins, '__import__', no_hub)
    with pytest.raises(ImportError, match='huggingface_hub'):
        EOFluxVAE.from_pretrained('nilsleh/eo-vae')


def test_bench_quotes_hbm_traffic_only_for_the_measured_build_and_mode(monkeypatch):
    """roofline.traffic comes from a committed PMC fold; it must turn into null -- with the reason -- when the kernel
    sources or the arithmetic mode differ from what the fold was measured on."""
    import bench
    from eo_vae_amd import ops
    val, src = bench.pmc_traffic('conv3x3_f16c8_kernel')
    assert src['file'] and src['file'].startswith('profiles/r')
    if val is not None:                                   # fold of this very source tree
        assert src['csrc_digest'] == bench.csrc_digest() and 'stale' not in src and val > 0
    else:
        assert 'stale' in src
    monkeypatch.setattr(ops, 'CONV_PRECISION', 'bf16')
    val2, src2 = bench.pmc_traffic('conv3x3_bf16_kernel')
    assert val2 is None and 'conv precision' in src2['stale']
    monkeypatch.setattr(bench, 'csrc_digest', lambda: 'deadbeefdeadbeef')
    monkeypatch.setattr(ops, 'CONV_PRECISION', ops.DEFAULT_PRECISION)
    val3, src3 = bench.pmc_traffic('conv3x3_f16c8_kernel')
    assert val3 is None and 'kernel sources changed' in src3['stale']


def test_packed_records_fail_loudly_outside_the_convolution():
    """GroupNorm's f16c8 operand records ride in a float32 shell (ops.PackedRecords).  The marking is the tensor's
    TYPE, so it survives detach() / clone() / contiguous() (a Python attribute did not, and the bytes were
    then computed on as fp32), and any torch operation on the bytes raises instead of returning garbage."""
    from eo_vae_amd import ops
    from eo_vae_amd.lib import EovaeError
    p = ops.mark_packed(torch.randn(2, 64, 4, 4))
    assert ops.is_packed(p) and not ops.is_packed(torch.zeros(1))
    for q in (p.detach(), p.clone(), p.contiguous(), p.detach().requires_grad_(True)):
        assert ops.is_packed(q) and q.shape == p.shape
    assert 'operand records' in repr(p)
    for op in (lambda a: a + 1, lambda a: a * 2.0, lambda a: torch.nn.functional.silu(a), lambda a: a.float(),
               lambda a: a.sum(), lambda a: a.cpu(), lambda a: a.view(2, -1), lambda a: a[0], lambda a: a.mul_(2),
               lambda a: torch.cat([a, a]), lambda a: a.numpy()):
        with pytest.raises(EovaeError, match='operand records'):
            op(p)


def test_groupnorm_backward_switch_and_plan_without_a_gpu():
    """eovae_gn_bwd_fused: 0 / 1 / 2 set, anything else queries; no shape is eligible for the cooperative kernel on a
    box without a GPU (the plan needs the CU count) and asking does not crash."""
    from eo_vae_amd import lib as L
    lib = L.load()
    before = lib.eovae_gn_bwd_fused(-1)
    try:
        assert lib.eovae_gn_bwd_fused(0) == 0 and lib.eovae_gn_bwd_fused(-7) == 0
        assert lib.eovae_gn_bwd_fused(2) == 2 and lib.eovae_gn_bwd_fused(99) == 2
        assert lib.eovae_gn_bwd_fused(1) == 1
        import torch
        if not torch.cuda.is_available():
            assert lib.eovae_gn_bwd_fused_eligible(16, 128, 256 * 256, 32) == 0
        assert lib.eovae_gn_bwd_fused_eligible(0, 128, 4096, 32) == 0       # malformed shapes: not eligible, no error
        assert lib.eovae_gn_bwd_fused_eligible(2, 100, 4096, 32) == 0
        # the workspace covers the cooperative kernel's granules and partials when the shape could take it
        assert lib.eovae_gn_workspace_bytes(2, 64, 4096, 32) >= 2 * 64 * 4 * (16 + 8)
    finally:
        lib.eovae_gn_bwd_fused(before if before in (0, 1, 2) else 1)


def test_zero_pool_hands_out_disjoint_zeroed_slices():
    """ops.zeros_small: slices of one pre-zeroed pool, never handed out twice, 256-byte aligned; big requests and a
    used-up pool fall back to a fresh allocation."""
    import torch
    from eo_vae_amd import ops
    dev = torch.device('cpu')
    ops._ZERO_POOL.clear()
    a = ops.zeros_small(5, torch.float32, dev)
    b = ops.zeros_small(1024, torch.float32, dev)
    c = ops.zeros_small(3, torch.float64, dev)
    assert a.shape == (5,) and b.shape == (1024,) and c.dtype == torch.float64
    assert a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() != c.untyped_storage().data_ptr()
    assert b.data_ptr() - a.data_ptr() == 64 * 4 and b.data_ptr() % 256 == a.data_ptr() % 256
    a.fill_(1.0)
    b.fill_(2.0)
    d = ops.zeros_small(7, torch.float32, dev)
    assert float(d.abs().sum()) == 0.0 and float(a.sum()) == 5.0 and float(b.sum()) == 2048.0
    big = ops.zeros_small(ops._ZERO_POOL_ELEMS, torch.float32, dev)          # too large for the pool
    assert big.untyped_storage().data_ptr() != a.untyped_storage().data_ptr() and float(big.abs().sum()) == 0.0
    first = a.untyped_storage().data_ptr()
    for _ in range(2 * ops._ZERO_POOL_ELEMS // (ops._ZERO_POOL_ELEMS // 8) + 2):   # exhaust the pool
        e = ops.zeros_small(ops._ZERO_POOL_ELEMS // 8, torch.float32, dev)
        assert float(e.abs().sum()) == 0.0
        e.fill_(3.0)
    assert e.untyped_storage().data_ptr() != first and float(a.sum()) == 5.0
    ops._ZERO_POOL.clear()


def test_batch_prep_host_logic_against_the_reference_collate(golden, monkeypatch):
    """N2 host side on the CPU: data.norm_table = the reference's NormalizerFactory tables, and data.GpuBatchPrep draws the
    modality and the D4 element in the reference's order (fixture: the reference's single_modality_collate_fn,
    terramesh_datamodule.py:418-503).  The HIP launch is replaced by the same arithmetic in torch (no GPU here); the GPU
    test (test_gpu_kernels2.py) runs the kernel against the same fixture."""
    import random
    import numpy as np
    import pytest
    import torch
    import torch.nn.functional as F
    from eo_vae_amd import data as D
    g = golden('batch_prep')

    def cpu_prepare(raw, mean=None, std=None, clamp=None, target_size=None, hflip=False, vflip=False, rot=0):
        x = raw.float()
        if clamp is not None:
            x = torch.clamp(x, clamp[0], clamp[1])
        if mean is not None:
            x = (x - mean.view(-1, 1, 1)) / std.view(-1, 1, 1)
        if target_size is not None:
            x = F.interpolate(x, size=tuple(target_size), mode='bilinear', align_corners=False)
        if hflip:
            x = torch.flip(x, dims=[-1])
        if vflip:
            x = torch.flip(x, dims=[-2])
        return torch.rot90(x, rot, dims=[-2, -1]) if rot else x
    monkeypatch.setattr(D, 'prepare_batch', cpu_prepare)
    T = lambda a: torch.from_numpy(np.asarray(a))                                                             # noqa: E731
    for mod, scheme, key_raw, key_out in (('S2L1C', 'custom', 'raw_l1c', 'norm_l1c_custom'), ('S1RTC', 'legacy', 'raw_s1rtc', 'norm_s1rtc_legacy'),
                                          ('S2L2A', 'legacy', 'raw', 'norm_l2a_legacy'), ('S1RTC', 'custom', 'raw_s1rtc', 'norm_s1rtc_legacy')):
        mean, div, clamp = D.norm_table(mod, scheme)
        got = cpu_prepare(T(g[key_raw]), mean, div, clamp)
        assert (got - T(g[key_out])).abs().max().item() < 1e-6, (mod, scheme)
    with pytest.raises(ValueError):
        D.norm_table('S1GRD', 'legacy')
    prep = D.GpuBatchPrep(['S2L2A', 'S1RTC'], normalize=True, norm_scheme='custom', target_size=(24, 32), mode='train',
                          device=torch.device('cpu'))
    raw = {'S2L2A': T(g['raw']), 'S1RTC': T(g['raw_s1rtc'])}
    for seed in range(8):
        random.seed(100 + seed)
        out = prep(raw)
        assert out['modality'] == str(g[f'collate{seed}_modality'])
        assert (out['image'] - T(g[f'collate{seed}_image'])).abs().max().item() < 2e-5


def test_wavelength_embedding_is_memoised_per_tensor_object():
    """dynamic_conv.wavelength_embedding: the same tensor object hits, an in-place edit or another object misses, a tensor
    that requires grad is never cached (reference: get_1d_sincos_pos_embed_from_grid_torch(D, wvs * 1000),
    dynamic_conv.py:466-468)."""
    from eo_vae_amd import dynamic_conv as DC
    DC._SINCOS_CACHE.clear()
    w = torch.tensor([0.665, 0.56, 0.49])
    a = DC.wavelength_embedding(64, w)
    assert DC.wavelength_embedding(64, w) is a
    assert torch.equal(a, DC.get_1d_sincos_pos_embed_from_grid_torch(64, w * 1000))
    w.mul_(2.0)                                           # in-place edit: _version moves
    b = DC.wavelength_embedding(64, w)
    assert b is not a and torch.equal(b, DC.get_1d_sincos_pos_embed_from_grid_torch(64, w * 1000))
    w2 = w.clone()
    assert DC.wavelength_embedding(64, w2) is not b       # another object, same values: its own entry
    wg = torch.tensor([0.5, 0.6], requires_grad=True)
    e = DC.wavelength_embedding(64, wg)
    assert e.requires_grad and (id(wg), 64) not in DC._SINCOS_CACHE


def test_no_register_write_directly_behind_a_16_byte_buffer_store():
    """gfx950 store-data hazard (round 6; tools/probe_store_war.hip, profiles/r06b_probe_store_war.log): a VALU write of a
    buffer_store_dwordx4's data registers issued directly behind the store reaches memory in lanes 12..15 of every 16-lane
    row, and hipcc 7.2 inserts its wait state only for stores without a register soffset.  The one-wave kernels store through
    store128_guarded(); this test compiles them and scans the compiler's output (tools/store_war_lint.py): no BUFFER store
    of 16 bytes may be followed at once by a write of its data.  (Scratch spills and loads that return much later are not the
    hazard's pattern.)"""
    import importlib.util
    import shutil
    if not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
        pytest.skip('hipcc not available')
    spec = importlib.util.spec_from_file_location('store_war_lint', os.path.join(ROOT, 'tools', 'store_war_lint.py'))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    for name in ('conv_f16c8_q16.hip', 'conv_f16c8_q16up.hip', 'conv_f16c8_q16upg.hip', 'conv_f16c8_w4.hip'):
        hits = [h for h in lint.scan(os.path.join(ROOT, 'eo_vae_amd', 'csrc', name))
                if h[0] == 1 and h[1].startswith('buffer_store_dwordx4') and h[2].split()[0].startswith('v_')]
        assert not hits, (name, hits[:3])
