"""GPU, two processes sharing the one card: Fitter(rank, world=2) + dist.GradBucket (the shape of BASELINE configs[3],
8 x MI355X data-parallel, scaled down): replicas stay bit-identical and equal a single-process run on the union of the
frames.  Children are spawned BEFORE the parent touches the GPU and never re-exec.  RCCL refuses two ranks on one
device ("Duplicate GPU detected", probed with scripts/rccl_same_gpu_probe.py), so on this 1-GPU box the process group
is gloo carrying the GPU gradient bucket; the RCCL path is the same torch.distributed call with backend "nccl" and runs
on the driver's 8-GPU node."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import numpy as np
import torch
from fpc_diffrend_amd import dist as fdist, fit, scene
rank, world, _ = fdist.init()
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
sc = scene.cfg('cfg1', n_frames=4)
sc.q_gt[:] = (0.0, 0.0, 0.0, 1.0)
if {short}[0] >= 0:
    import fpc_diffrend_amd.ops as dr_
    dr_.SMALL_BATCH_BINS, dr_.RECORD_SLOT_MARGIN = 0, 1      # compact records for this small batch too
cfg = fit.FitConfig(max_iter=30, cam_idxs=(0, 3, 6), lr_base=5e-3, lr_t=5e-3, lr_q=1e-5, weight_laplacian=40.0,
                    weight_meshedge=0.3, init_texture='truth', hip_graph={graph})
ft = fit.Fitter(sc, cfg, device=dev, rank=rank, world=world)
ft.init_near_truth(0.8)
bucket = fdist.GradBucket(ft.params, dev, early=[ft.tex_opt] if {early} else ())      # early: dist.EarlyReduce for the texture
ft.reduce_fn = bucket if world > 1 else None
short_at, short_rank = {short}
losses, snaps = [], []
for it in range({steps}):
    h = None
    if it == short_at and rank == short_rank:      # ONE rank's record pool is too small in this iteration (tests/test_gpu_skip.py)
        import fpc_diffrend_amd.ops as dr
        h = dr._list_hints[next(k for k in dr._list_hints if k[0] == 'onepass')]
        h.poll()
        h.slots, h.frozen = 1, True
    losses.append(float(ft.step()))
    if h is not None:
        h.frozen = False
    if short_at >= 0:
        snaps.append([p.detach().cpu().clone() for p in ft.params])
if short_at >= 0:
    torch.save({{"snaps": snaps, "skipped": ft.skipped_steps, "losses": losses}}, {out!r} + f"/skip_w{{world}}_r{{rank}}.pt")
    losses = [0.0 if l != l else l for l in losses]
if world > 1:
    losses = [fdist.sum_over_ranks(l, dev) for l in losses]       # each rank reports its share of the global mean
res = ft.gather_result()
ft.save({out!r} + f"/save_w{{world}}")
torch.save({{"losses": losses, "params": [p.detach().cpu() for p in ft.params], "result": res.cpu(), "calls": bucket.calls,
            "early_fired": [e.fired for e in bucket.early]}},
           {out!r} + f"/w{{world}}_r{{rank}}.pt")
if world > 1:
    import torch.distributed as tdist
    tdist.barrier()
    tdist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(tmp_path, world, graph, steps, early=False, short=(-1, -1)):
    import subprocess
    script = tmp_path / f"child_w{world}.py"
    script.write_text(CHILD.format(root=ROOT, out=str(tmp_path), graph=graph, steps=steps, early=early, short=tuple(short)))
    procs = []
    port = _free_port()
    for r in range(world):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FPCDR_DIST_BACKEND="gloo")
        if world > 1:
            env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r))
        else:
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


@pytest.mark.parametrize("graph,early", [(False, False), (True, False), (False, True)])
def test_two_rank_fitter_equals_single_process(tmp_path, graph, early):
    """early: the texture gradient is all-reduced on its own from an autograd hook, beside the rest of the backward pass
    (dist.EarlyReduce; eager steps only), everything else in the flat bucket."""
    import torch      # imported here; nothing in this test process touches the GPU
    steps = 6 if graph else 3       # graph path: three eager steps, one that fixes the parameter set, then capture + replay
    _run(tmp_path, 2, graph, steps, early)
    _run(tmp_path, 1, graph, steps)
    a, b = (torch.load(tmp_path / f"w2_r{r}.pt") for r in (0, 1))
    one = torch.load(tmp_path / "w1_r0.pt")
    assert a["calls"] == steps and b["calls"] == steps      # exactly one bucket all-reduce per step
    assert a["early_fired"] == ([steps] if early else [])   # ... plus the texture's own, once per step
    for k, (p, q) in enumerate(zip(a["params"], b["params"])):
        assert torch.equal(p, q), (f"replicas diverged: parameter {k} {tuple(p.shape)}: max |a - b| = {float((p - q).abs().max())}, "
                                   f"non-finite a / b: {int((~torch.isfinite(p)).sum())} / {int((~torch.isfinite(q)).sum())}, losses {a['losses']} {b['losses']}")
    assert torch.equal(a["result"], b["result"])
    assert np.allclose(a["losses"], one["losses"], rtol=1e-5), (a["losses"], one["losses"])
    # two ranks against one process: the same gradients up to the ORDER of the float additions (atomics inside a rank, the all-reduce
    # across ranks).  Adam turns a gradient component that cancels to ~0 into a step of about the learning rate whose sign is that
    # order's (one texel in 260 k moved by 1.16e-3 = its learning rate in one run out of three, everything else agreed to 1e-6): all
    # but a vanishing share of every tensor's entries must agree to 1e-5, and no entry may be further off than a few steps
    # the optimiser's ten groups in order (fit.py:493-505): m1 m2 m3 M1 M2 t_opt q_opt per_frame_t per_frame_q tex; learning rates of the child
    lrs = [5e-3] * 5 + [5e-3, 1e-5, 5e-3, 1e-5, 2.5e-3]
    for k, (p, q) in enumerate(zip(a["params"], one["params"])):
        d = (p - q).abs()
        tol = 1e-5 * max(1.0, float(q.abs().max()))
        if k != 9:      # everything but the texture: EVERY entry (36 pose numbers must not hide behind a share)
            assert float(d.max()) <= tol, (k, tuple(p.shape), float(d.max()), tol)
        else:           # the texture: all but a vanishing share, and no entry further off than Adam can move it in these steps
            share = float((d > tol).float().mean())
            assert share <= 2e-4 and float(d.max()) <= 2.0 * lrs[k] * steps, (k, tuple(p.shape), share, float(d.max()))
    # every frame's final mesh is present after the gather (rank 1's rows are not left at zero) and equals the single run
    assert float(a["result"].abs().sum(dim=1).min()) > 0
    assert float((a["result"] - one["result"]).abs().max()) < 1e-4
    # rank 0 wrote all four frames
    assert sorted(os.listdir(tmp_path / "save_w2" / "result"))[:4] == ["0.obj", "1.obj", "2.obj", "3.obj"]


def test_one_rank_short_of_record_slots_makes_both_ranks_skip_the_same_step(tmp_path):
    """Rank 1's pixel objective runs out of record slots in iteration 2 (forced through its launch-hint record).  Its flag rides in the
    gradient bucket's extra element (dist.GradBucket.flag), the sum reaches both Adam launches (fpcdr_adam_params.skip_flag): BOTH ranks
    leave their parameters as they were, nobody raises between two collectives, nobody hangs, the replicas stay bit-identical and the
    run continues."""
    import torch
    steps, k = 5, 2
    _run(tmp_path, 2, False, steps, short=(k, 1))
    a, b = (torch.load(tmp_path / f"skip_w2_r{r}.pt") for r in (0, 1))
    assert a["skipped"] == 1 and b["skipped"] == 1
    assert a["losses"][k] == a["losses"][k] and b["losses"][k] != b["losses"][k]       # rank 0's share is a number, rank 1's is NaN
    for it in range(steps):
        for p, q in zip(a["snaps"][it], b["snaps"][it]):
            assert torch.equal(p, q), f"replicas diverged in iteration {it}"
    for p, q in zip(a["snaps"][k], a["snaps"][k - 1]):
        assert torch.equal(p, q)                                                       # iteration k changed nothing ...
    assert any(not torch.equal(p, q) for p, q in zip(a["snaps"][k + 1], a["snaps"][k]))   # ... and iteration k + 1 went on
    assert all(bool(torch.isfinite(p).all()) for p in a["snaps"][-1])


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO torchrun environment (the shape of the driver's N = 1 command): the process becomes a launcher --
    it touches no GPU, starts two fresh ranks with a rendezvous on 127.0.0.1 and relays rank 0's JSON line as its own last line.  Here
    both ranks share the one GPU of the box over gloo (RCCL refuses two ranks on one device), which the line must SAY."""
    import json
    import subprocess
    env = dict(os.environ, FPCDR_DIST_BACKEND="gloo", FPCDR_BENCH_ALLOW_ANY_BACKEND="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--frames-per-gpu", "2",
           "--no-cpu-baseline", "--no-reference-shaped-step"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, (out[-2000:], r.stderr.decode(errors="replace")[-3000:])
    line = json.loads(out.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["backend"] == "gloo" and line["ranks_seen"] == 1 and len(line["per_rank"]) == 2
    assert line["value"] > 0 and line["scaling"] == "weak" and line["allreduce_ms"] is not None
    assert line["config"]["frames_per_gpu"] == 2 and sorted(r_["rank"] for r_ in line["per_rank"]) == [0, 1]
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "bench_2rank_selflaunch.json"), "w") as f:
            f.write(json.dumps(line) + "\n")
    # a rank that fails takes the launcher down with it (non-zero exit, no JSON line relayed as a result): without the gloo override
    # RCCL refuses the second rank on the same device ("Duplicate GPU"), or check_world refuses the run
    env_bad = {k_: v for k_, v in env.items() if k_ not in ("FPCDR_DIST_BACKEND", "FPCDR_BENCH_ALLOW_ANY_BACKEND")}
    env_bad["FPCDR_BENCH_LAUNCH_TIMEOUT"] = "300"
    r = subprocess.run(cmd, env=env_bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode != 0


def test_rank_3_of_8_computes_its_shard_of_a_256_frame_take():
    """BASELINE configs[3] in shape -- 256 frames over 8 ranks, 32 each -- on one GPU with a small raster: Fitter(rank=3, world=8) owns
    frames 96..127 and must leave, in the GLOBAL rows / columns of the shared parameters (M1's columns, per_frame_t / q's rows), exactly
    1/8 of the loss and gradients a single-process Fitter computes when it steps those 32 frames (its mean runs over 32 frames' pixels,
    the rank's over the global batch of 256: the all-reduce sums eight such shares)."""
    import torch
    from fpc_diffrend_amd import fit, scene
    F, world, rank = 256, 8, 3
    sc = scene.cfg('cfg1', n_frames=F)
    cfg = fit.FitConfig(max_iter=100, cam_idxs=(0, 4), lr_base=5e-3, lr_t=5e-3, lr_q=1e-5, weight_laplacian=40.0, init_texture='truth',
                        resolution=(128, 128))
    sc.cams = scene.make_cameras((128, 128))
    shard = fit.Fitter(sc, cfg, device='cuda', rank=rank, world=world)
    assert (shard.frame_lo, shard.frame_hi) == (96, 128) and shard.targets.shape[0] == 32
    whole = fit.Fitter(sc, cfg, device='cuda', rank=0, world=1)
    assert torch.equal(whole.targets[96:128], shard.targets)
    for ft in (shard, whole):
        ft.init_near_truth(0.8)
    l_shard = shard.loss_and_backward(shard.pick_frames())
    assert shard.pick_frames() == slice(96, 128)
    l_whole = whole.loss_and_backward(slice(96, 128))
    assert abs(float(l_shard) * world - float(l_whole)) <= 2e-5 * abs(float(l_whole)), (float(l_shard), float(l_whole))
    names = ("m1", "m2", "m3", "M1", "M2", "t_opt", "q_opt", "per_frame_t", "per_frame_q", "tex")
    for name, p, q in zip(names, shard.params, whole.params):
        if q.grad is None:
            assert p.grad is None, name
            continue
        g, want = p.grad.double() * world, q.grad.double()
        err = float((g - want).norm() / max(float(want.norm()), 1e-30))
        assert err < 2e-5, (name, err)
    # the shard's gradient lives in the GLOBAL index space: columns 96..127 of M1, rows 96..127 of the per-frame pose, nothing elsewhere
    M1g = shard.maps['local'].grad
    assert float(M1g[:, :96].abs().max()) == 0.0 and float(M1g[:, 128:].abs().max()) == 0.0 and float(M1g[:, 96:128].abs().max()) > 0.0
    for g in (shard.per_frame_t.grad, shard.per_frame_q.grad):
        assert float(g[:96].abs().max()) == 0.0 and float(g[128:].abs().max()) == 0.0 and float(g[96:128].abs().max()) > 0.0
    # ... and its result rows are frames 96..127 of the take
    res = shard.result
    assert float(res[:96].abs().max()) == 0.0 and float(res[128:].abs().max()) == 0.0 and float(res[96:128].abs().sum(dim=1).min()) > 0.0
