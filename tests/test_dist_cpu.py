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
    flags = []
    for it in range(4):
        opt.zero_grad()
        frames = torch.arange(lo, hi)
        w_f = (M2 @ (M1[:, frames] + torch.eye(F)[:, frames])).t()          # [Fb,K]
        loss = ((w_f - target[frames]) ** 2).sum() / F + (tex ** 2).mean() / world
        loss.backward()
        # the skip flag rides behind the gradients (GradBucket.flag): every step each rank WRITES it, as the pixel objective's last kernel
        # does -- here the last rank alone calls step 1 invalid -- and after the one collective every rank holds the same sum
        bucket.flag.fill_(1.0 if (it == 1 and rank == world - 1) else 0.0)
        bucket(params)
        flags.append(float(bucket.flag))
        if flags[-1] == 0.0:      # (fpcdr_adam_step does this on the device)
            opt.step()
    assert flags == [0.0, 1.0, 0.0, 0.0], flags
    assert bucket.calls == 4 and (not early or bucket.early[0].fired == 4)
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
    for _ in range(3):      # (the replicas' four iterations minus the one they all skip)
        opt.zero_grad()
        w_f = (M2 @ (M1 + torch.eye(F))).t()
        loss = ((w_f - target) ** 2).sum() / F + (tex ** 2).mean()
        loss.backward()
        opt.step()
    return torch.cat([p.detach().reshape(-1) for p in (M1, M2, tex)])


@pytest.mark.parametrize("world,early", [(2, False), (2, True), (4, False), (4, True)])
def test_data_parallel_replicas_equal_single_process(world, early, monkeypatch):
    """World size 2 and 4 over gloo.  early: the texture's gradient travels in its own all-reduce, launched from an autograd hook as
    soon as it exists (dist.EarlyReduce), the rest in the bucket -- same replicas, same result.  One rank declares one step invalid
    through the bucket's flag element: every rank skips that update, nobody hangs, the result is the run without that iteration."""
    monkeypatch.setenv("FPCDR_TEST_EARLY", "1" if early else "0")
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(1, world):
        assert torch.equal(ret[0], ret[r]), f"replica {r} diverged"
    ref = _single()
    assert torch.allclose(ret[0], ref, atol=1e-6), float((ret[0] - ref).abs().max())


def test_frame_sharding_rule_and_uneven_split_is_refused():
    """dist.shard_frames: contiguous, disjoint, covering; a frame count that does not divide over the ranks raises (every rank must
    step the same number of images), as does the Fitter's constructor through it."""
    from fpc_diffrend_amd import dist as fdist
    for F, world in ((256, 8), (32, 4), (6, 3), (5, 1)):
        spans = [fdist.shard_frames(F, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == F
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and len({hi - lo for lo, hi in spans}) == 1
    for F, world in ((10, 4), (33, 8), (1, 2)):
        with pytest.raises(ValueError, match="divide evenly"):
            fdist.shard_frames(F, 0, world)
    with pytest.raises(ValueError):
        fdist.shard_frames(8, 4, 4)


def _check_world_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from fpc_diffrend_amd import dist as fdist
    fdist.init(backend="gloo")
    info = fdist.check_world(world, "cpu")                       # CPU ranks are distinct "devices" (one per process)
    out = {"info": info}
    for name, kw in (("size", dict(expected_world=world + 1, device="cpu")), ("backend", dict(expected_world=world, device="cpu", require_backend="nccl"))):
        try:
            fdist.check_world(**kw)
            out[name] = None
        except RuntimeError as e:
            out[name] = str(e)
    ret[rank] = out
    tdist.destroy_process_group()


def test_check_world_fails_loudly_on_a_wrong_group():
    """dist.check_world (bench.py runs it before the timed region of a --gpus N > 1 run): passes on the real group, raises when the
    group's size is not the one asked for or the backend is not the required one (a gloo fallback must not produce a scaling number)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_check_world_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r]["info"] == {"backend": "gloo", "world": 2, "ranks_seen": 2}
        assert "expected 3 ranks" in ret[r]["size"] and "not 'nccl'" in ret[r]["backend"]
    from fpc_diffrend_amd import dist as fdist
    assert fdist.check_world(1, "cpu") == {"backend": None, "world": 1, "ranks_seen": 1}      # a single process passes trivially


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
