"""CPU, world_size 2 over gloo: the data-parallel engine (flat buffers, bucketed async all-reduce
from post-accumulate hooks, averaged gradients, buffer broadcast) on a small torch model."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(10, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64),
                               torch.nn.Tanh(), torch.nn.Linear(64, 3))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from eo_vae_amd.optim import FlatParams
    from eo_vae_amd.parallel import BucketedGradReducer, init_distributed
    init_distributed('gloo')
    model = _make_model()
    if rank == 1:  # diverge on purpose: broadcast_parameters must repair it
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    flat = FlatParams(model.parameters())
    red = BucketedGradReducer(flat, bucket_mb=0.002)   # ~500 floats per bucket -> several buckets
    assert len(red.buckets) >= 2
    buf = torch.full((4,), float(rank))
    red.broadcast_parameters([buf])
    assert torch.all(buf == 0)
    g = torch.Generator().manual_seed(100 + rank)
    out = []
    for step in range(2):
        x = torch.randn(8, 10, generator=g)
        y = torch.randn(8, 3, generator=g)
        flat.zero_grad()
        loss = torch.nn.functional.mse_loss(model(x), y)
        loss.backward()
        red.finish()
        out.append(flat.grad.clone())
        with torch.no_grad():
            flat.data.add_(flat.grad, alpha=-0.1)
    # one unused-parameter step: bucket launched by finish()
    flat.zero_grad()
    model[0](torch.randn(2, 10, generator=g)).sum().backward()
    red.finish()
    q.put((rank, [o.numpy() for o in out], flat.data.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_allreduce_world2_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference: average of the two ranks' gradients
    from eo_vae_amd.optim import FlatParams
    model = _make_model()
    flat = FlatParams(model.parameters())
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    for step in range(2):
        grads = []
        for r in range(world):
            x = torch.randn(8, 10, generator=gens[r])
            y = torch.randn(8, 3, generator=gens[r])
            flat.zero_grad()
            torch.nn.functional.mse_loss(model(x), y).backward()
            grads.append(flat.grad.clone())
        avg = (grads[0] + grads[1]) / 2
        for r in range(world):
            assert torch.allclose(torch.from_numpy(res[r][1][step]), avg, atol=1e-6)
        with torch.no_grad():
            flat.data.add_(avg, alpha=-0.1)
    assert torch.allclose(torch.from_numpy(res[0][2]), torch.from_numpy(res[1][2]), atol=0)  # replicas agree
    assert torch.allclose(torch.from_numpy(res[0][2]), flat.data, atol=1e-6)


def test_flat_params_views_survive_zero_grad():
    from eo_vae_amd.optim import FlatParams
    m = _make_model()
    flat = FlatParams(m.parameters())
    m(torch.randn(4, 10)).sum().backward()
    assert flat.grad.abs().sum() > 0
    for p in m.parameters():
        p.grad = None                      # what zero_grad(set_to_none=True) would do
    flat.zero_grad()
    m(torch.randn(4, 10)).sum().backward()
    assert flat.grad.abs().sum() > 0 and all(p.grad.data_ptr() != 0 for p in m.parameters())
    s, e = flat.span(0)
    assert torch.equal(flat.grad[s:e].view_as(m[0].weight), m[0].weight.grad)


def test_running_stats_merge_formula():
    """Cross-rank Chan merge == feeding all batches to one accumulator (oracle)."""
    from oracle import eo_vae_cpu as O
    from eo_vae_amd.encode_latents import RunningStats
    zs = [torch.randn(4, 6, 5, 5) * (i + 1) + i for i in range(4)]
    ref = O.RunningStats(6)
    parts = []
    for half in (zs[:2], zs[2:]):
        o = O.RunningStats(6)
        for z in half:
            o.update(z)
            ref.update(z)
        parts.append(torch.stack([o.mean, o.var, o.min, o.max, o.count.expand(6)]))
    merged = RunningStats._merge(parts[0], parts[1])
    assert torch.allclose(merged[0], ref.mean, atol=1e-5)
    assert torch.allclose(merged[1], ref.var, rtol=1e-4)
    assert torch.equal(merged[2], ref.min) and torch.equal(merged[3], ref.max)
    assert merged[4, 0] == ref.count[0]


def _stats_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from eo_vae_amd.encode_latents import RunningStats
    from eo_vae_amd.parallel import init_distributed
    init_distributed('gloo')
    g = torch.Generator().manual_seed(7 + rank)
    z = torch.randn(6 + 3 * rank, 4, 5, 5, generator=g) * (1.0 + rank) + rank      # ragged per-rank sample counts
    rs = RunningStats(4, 'cpu')
    flat = z.permute(1, 0, 2, 3).reshape(4, -1)
    rs.state = torch.stack([flat.mean(1), flat.var(1, unbiased=False), flat.amin(1), flat.amax(1),
                            torch.full((4,), float(flat.shape[1]))])
    rs.merge_across_ranks()
    q.put((rank, z.numpy(), rs.state.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_running_stats_merge_across_ranks_world2():
    """cfg5's one collective: all_gather of the per-rank (mean, var, min, max, count) + ordered Chan merges
    (reference encode_latents.py:65-94 applied across ranks) must equal the statistics of the pooled data."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_stats_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    pooled = torch.cat([torch.from_numpy(r[1]).permute(1, 0, 2, 3).reshape(4, -1) for r in res], dim=1).double()
    for _, _, state in res:                      # every rank ends with the same merged state
        st = torch.from_numpy(state).double()
        assert torch.allclose(st[0], pooled.mean(1), rtol=1e-5, atol=1e-6)
        assert torch.allclose(st[1], pooled.var(1, unbiased=False), rtol=1e-4, atol=1e-6)
        assert torch.equal(st[2], pooled.amin(1)) and torch.equal(st[3], pooled.amax(1))
        assert torch.all(st[4] == pooled.shape[1])


def _fault_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from eo_vae_amd import ops
    from eo_vae_amd.optim import FlatParams
    from eo_vae_amd.parallel import BucketedGradReducer, init_distributed
    init_distributed('gloo')
    model = _make_model()
    flat = FlatParams(model.parameters())
    red = BucketedGradReducer(flat, bucket_mb=0.002)
    assert ops.GN_TIMEOUT_AGREED == red.fault_agreed
    state = {'step': -1}
    red.local_fault = lambda: rank == 1 and state['step'] == 1      # ONE rank sees a fault, in its step 1
    g = torch.Generator().manual_seed(7 + rank)
    seen = []
    for step in range(6):
        state['step'] = step
        seen.append(red.fault_agreed())          # what training_step asks before it enqueues anything
        flat.zero_grad()
        torch.nn.functional.mse_loss(model(torch.randn(4, 10, generator=g)), torch.randn(4, 3, generator=g)).backward()
        red.finish()
    red.remove()
    assert ops.GN_TIMEOUT_AGREED is None
    q.put((rank, seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_rank_local_fault_stops_every_rank_in_the_same_step():
    """a GroupNorm-backward time-out is seen by one rank; the flag rides a MAX all-reduce beside the gradient
    buckets and every rank reads the same reduced value HEALTH_LAG steps later -- both ranks agree, in the same step."""
    from eo_vae_amd.parallel import BucketedGradReducer
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_fault_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    lag = BucketedGradReducer.HEALTH_LAG
    want = [s == 1 + lag for s in range(6)]
    assert res[0] == want and res[1] == want
