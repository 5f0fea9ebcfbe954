"""CPU, world_size 2 over gloo: the data-parallel path (one flat-bucket all-reduce per step, frames sharded
contiguously, replicas stay bit-identical).  The raster ops need a GPU, so the per-rank "render" here is a small
differentiable stand-in with the same parameter structure; what is tested is dist.GradBucket + the sharding rule."""
import os
import socket

import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from fpc_diffrend_amd import dist as fdist
    r, w, _ = fdist.init(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)                       # replicated parameters
    F, K = 8, 5
    M1 = torch.zeros(F, F, requires_grad=True)
    M2 = torch.eye(K, F).requires_grad_(True)
    tex = torch.rand(4, 4).requires_grad_(True)
    frozen = torch.zeros(3, 3)                 # requires_grad False: must not travel
    params = [frozen, M1, M2, tex]
    early = bool(int(os.environ.get("FPCDR_TEST_EARLY", "0")))       # the texture's gradient reduced on its own, from an autograd hook
    bucket = fdist.GradBucket(params, "cpu", early=[tex] if early else ())
    assert bucket.nbytes == 4 * (F * F + K * F + (0 if early else 16))
    opt = torch.optim.Adam([M1, M2, tex], lr=1e-2)
    target = torch.linspace(0, 1, F * K).reshape(F, K)
    lo, hi = rank * F // world, (rank + 1) * F // world        # contiguous frame shard
    for _ in range(3):
        opt.zero_grad()
        frames = torch.arange(lo, hi)
        w_f = (M2 @ (M1[:, frames] + torch.eye(F)[:, frames])).t()          # [Fb,K]
        loss = ((w_f - target[frames]) ** 2).sum() / F + (tex ** 2).mean() / world
        loss.backward()
        bucket(params)
        opt.step()
    assert bucket.calls == 3 and (not early or bucket.early[0].fired == 3)
    ret[rank] = torch.cat([p.detach().reshape(-1) for p in (M1, M2, tex)])
    fdist.barrier()
    assert fdist.max_over_ranks(float(rank), "cpu") == world - 1
    assert fdist.sum_over_ranks(1.0, "cpu") == world
    tdist.destroy_process_group()


def _single(F=8, K=5):
    torch.manual_seed(0)
    M1 = torch.zeros(F, F, requires_grad=True)
    M2 = torch.eye(K, F).requires_grad_(True)
    tex = torch.rand(4, 4).requires_grad_(True)
    opt = torch.optim.Adam([M1, M2, tex], lr=1e-2)
    target = torch.linspace(0, 1, F * K).reshape(F, K)
    for _ in range(3):
        opt.zero_grad()
        w_f = (M2 @ (M1 + torch.eye(F))).t()
        loss = ((w_f - target) ** 2).sum() / F + (tex ** 2).mean()
        loss.backward()
        opt.step()
    return torch.cat([p.detach().reshape(-1) for p in (M1, M2, tex)])


@pytest.mark.parametrize("early", [False, True])
def test_two_rank_data_parallel_equals_single_process(early, monkeypatch):
    """early: the texture's gradient travels in its own all-reduce, launched from an autograd hook as soon as it exists
    (dist.EarlyReduce), the rest in the bucket -- same replicas, same result."""
    monkeypatch.setenv("FPCDR_TEST_EARLY", "1" if early else "0")
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert torch.equal(a, b), "replicas diverged"
    ref = _single()
    assert torch.allclose(a, ref, atol=1e-6), float((a - ref).abs().max())


def test_grad_bucket_tracks_requires_grad_changes():
    from fpc_diffrend_amd import dist as fdist
    a = torch.zeros(3, requires_grad=True)
    b = torch.zeros(2)
    bk = fdist.GradBucket([a, b], "cpu")
    assert bk.nbytes == 12
    a.grad = torch.ones(3)
    bk()
    assert torch.equal(a.grad, torch.ones(3))
    b.requires_grad = True                      # combined mode switches parameters on half-way (reference fit.py:603-608)
    b.grad = torch.full((2,), 2.0)
    bk()
    assert bk.nbytes == 20 and torch.equal(b.grad, torch.full((2,), 2.0))


def _one_rank_worker(port, ret):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from fpc_diffrend_amd import dist as fdist
    assert fdist.init(backend="gloo") == (0, 1, 0) and not tdist.is_initialized()      # a single process: no group
    assert fdist.init(backend="gloo", force_group=True) == (0, 1, 0) and tdist.is_initialized()
    p = torch.nn.Parameter(torch.arange(6.0))
    p.grad = torch.ones(6) * 3
    bucket = fdist.GradBucket([p], "cpu", always_reduce=True, timed=True)      # (timing is a GPU feature: off on the CPU)
    bucket()
    assert bucket.calls == 1 and torch.equal(p.grad, torch.ones(6) * 3) and bucket.reduce_ms() is None
    ret[0] = True
    tdist.destroy_process_group()


def test_forced_one_rank_group():
    """dist.init(force_group=True): a one-rank process group (what the GPU test drives through RCCL), here over gloo."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), ret))
    p.start()
    p.join(120)
    assert p.exitcode == 0 and ret.get(0) is True
