"""
Data-parallel plumbing: one process per GPU, frames sharded contiguously over ranks, parameters
replicated, ONE RCCL all-reduce (sum) of a single flat f32 gradient bucket per Adam step over xGMI.

The reference is single-process / single-GPU: this is a capability of the
build, designed from the path's structure (section 8e): every (frame, view) image is an independent
render; frames interact only through the shared parameters, so the gradient sum is the one real
exchange step.  Payload in prior mode (F = 256, K = 150, 1024^2 x 1 texture): ~1.15 M floats = 4.6 MB
-- latency-bound on 7 x 153 GB/s xGMI links, so no overlap machinery is needed; weak-scaling
efficiency hinges on equal pixels per rank.

Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (world_size 2).
"""
import os

import torch
import torch.distributed as dist


def init(backend=None, force_group=False):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world, local_rank).  A single process (no env) returns (0, 1, 0) without a process group, unless
    force_group (or FPCDR_DIST_FORCE_GROUP=1) asks for a one-rank group: the collective library is then loaded and a
    communicator created exactly as on a multi-GPU node (tests/test_gpu_dist.py uses it to run RCCL on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force_group = force_group or os.environ.get("FPCDR_DIST_FORCE_GROUP", "0") == "1"
    if world <= 1 and not force_group:
        return 0, 1, 0
    if world <= 1:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if "MASTER_PORT" not in os.environ:      # a free port: concurrent one-rank runs on a box must not collide on a fixed one
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        world = 1
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = os.environ.get("FPCDR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    device_id = None
    if torch.cuda.is_available():
        # one process per GPU; the modulo only matters when a test oversubscribes a box with fewer GPUs than ranks
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        if backend == "nccl":
            # bind the communicator to this rank's device NOW: RCCL then initialises eagerly inside init_process_group instead of at
            # the first collective (where a wrong device mapping or a missing xGMI peer would surface in the middle of the timed loop)
            device_id = torch.device("cuda", torch.cuda.current_device())
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **({"device_id": device_id} if device_id is not None else {}))
    return rank, world, local_rank


def check_world(expected_world, device, require_backend=None):
    """Fail loudly, before any timed step, when a multi-rank run is not what it claims to be: the process group's size differs from
    `expected_world`, the backend is not `require_backend` (a silent gloo fallback would still produce numbers), or two ranks compute on
    the same physical device (device_identity).  Returns {"backend", "world", "ranks_seen"}.  A single process passes trivially."""
    backend = dist.get_backend() if dist.is_initialized() else None
    world = dist.get_world_size() if dist.is_initialized() else 1
    seen = len({d for d in gather_objects(device_identity(device))})
    info = {"backend": backend, "world": world, "ranks_seen": seen}
    if world != expected_world:
        raise RuntimeError(f"expected {expected_world} ranks, the process group has {world}: {info}")
    if expected_world > 1:
        if require_backend is not None and backend != require_backend:
            raise RuntimeError(f"multi-rank run over backend {backend!r}, not {require_backend!r}: {info}")
        if seen < world:
            raise RuntimeError(f"{world} ranks on {seen} distinct devices -- one process per GPU is required: {info}")
    return info


def shard_frames(n_frames, rank, world):
    """The contiguous frame range of `rank`: [rank * F / world, (rank + 1) * F / world).  Frames must divide
    evenly -- every rank steps the same number of images (weak scaling, identical launch shapes, one all-reduce per step in lockstep);
    an uneven split is refused, not rounded."""
    if n_frames <= 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad sharding request: {n_frames} frames, rank {rank} of {world}")
    if n_frames % world != 0:
        raise ValueError(f"{n_frames} frames do not divide evenly over {world} ranks: choose a frame count that is a multiple of the world size")
    return rank * n_frames // world, (rank + 1) * n_frames // world


class EarlyReduce:
    """All-reduce ONE parameter's gradient the moment autograd has produced it -- on the collective's own stream, beside the rest
    of the backward pass -- instead of inside the flat bucket at the end.  The fit loop's texture gradient (4 MB of the 4.4 MB
    payload in prior mode) is final right after the objective's backward kernel, while ~0.15 ms of small backward kernels
    (transform_clip, MVP chain, blend) are still to run; GradBucket(early=[...]) leaves the parameter out of the bucket and waits
    for the handle before the optimiser step.  Not for HIP-graph capture (the collective is launched from an autograd hook)."""

    def __init__(self, param, always=False, before=None):
        # before: called right before the collective is issued (e.g. ops.join_texture_flush: with the texel windows' flush on a side
        # stream the gradient handed to autograd is complete only once that stream has been joined)
        self.param, self.work, self.always, self.fired, self.before = param, None, always, 0, before
        self._hook = param.register_post_accumulate_grad_hook(self._fire)

    def _fire(self, p):
        if dist.is_initialized() and (dist.get_world_size() > 1 or self.always):
            if self.before is not None:
                self.before()
            if p.grad.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("EarlyReduce cannot run inside a HIP-graph capture (the collective is issued from an autograd hook on "
                                   "its own stream): use FitConfig(hip_graph=False) with early reduction, or leave the parameter in the bucket")
            self.work = dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True)
            self.fired += 1

    def wait(self):
        if self.work is not None:
            self.work.wait()        # the current stream waits for the collective's stream
            self.work = None

    def remove(self):
        self._hook.remove()


class GradBucket:
    """Flat gradient bucket.  `bucket(params)` packs every existing .grad into one contiguous f32 buffer,
    all-reduces it (sum) and unpacks -- exactly one collective per optimisation step.
    `bucket.flag`: ONE more float behind the gradients, a view into the same buffer, that rides in the same collective.  The fit loop lets
    the pixel objective's last kernel write it (1 = this rank's gradients of this step are invalid: fpcdr_objective_params.skip_out) and
    hands it, summed over the ranks, to the Adam launch (fpcdr_adam_params.skip_flag): every rank skips the same step, no rank raises
    between two collectives.  Nobody else writes it; it starts at 0."""

    def __init__(self, params, device, always_reduce=False, timed=False, early=(), early_before=None):
        # early: parameters whose gradient is reduced on its own as soon as it exists (EarlyReduce); the bucket skips them
        self.early = [EarlyReduce(p, always=always_reduce, before=early_before) for p in early]
        early_ids = {id(p) for p in early}
        self.all_params = [p for p in params if id(p) not in early_ids]
        self.device = device
        self.calls = 0
        self.always_reduce = always_reduce     # issue the collective even in a one-rank group (RCCL smoke test)
        # timed: HIP events around the collective alone (bench.py reports allreduce_ms / bucket_bytes per rank)
        self.timed = bool(timed) and torch.device(device).type == 'cuda'
        self._events = []
        self._layout(None)

    def _layout(self, sig):
        # only parameters that are being optimised travel (combined mode enables more half-way, fit.py:603-608)
        self.sig = tuple(p.requires_grad for p in self.all_params) if sig is None else sig
        self.params = [p for p in self.all_params if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]
        self.n_grad = max(sum(self.sizes), 1)
        self.flat = torch.zeros(self.n_grad + 1, dtype=torch.float32, device=self.device)
        self.flag = self.flat[self.n_grad:]

    def __call__(self, params=None):
        sig = tuple(p.requires_grad for p in self.all_params)
        if sig != self.sig:
            old_flag = self.flag
            self._layout(sig)
            self.flag.copy_(old_flag)      # (what this step's objective wrote moves into the new buffer)
        # pack / unpack with ONE multi-tensor copy each (a copy per tensor is ~5 us of launch in the serial tail of the step)
        views, off = [], 0
        for p, n in zip(self.params, self.sizes):
            views.append(self.flat[off:off + n])
            off += n
        have = [(v, p) for v, p in zip(views, self.params) if p.grad is not None]
        for v, p in zip(views, self.params):
            if p.grad is None:
                v.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [p.grad.reshape(-1) for _, p in have])
        if dist.is_initialized() and (dist.get_world_size() > 1 or self.always_reduce):
            if self.timed and not torch.cuda.is_current_stream_capturing():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                e1.record()
                self._events.append((e0, e1))
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.calls += 1
        for e in self.early:
            e.wait()
        if have:
            torch._foreach_copy_([p.grad.view(-1) if p.grad.is_contiguous() else p.grad for _, p in have],
                                 [v if p.grad.is_contiguous() else v.view_as(p.grad) for v, p in have])
        for v, p in zip(views, self.params):
            if p.grad is None:
                p.grad = v.view_as(p).clone()

    @property
    def nbytes(self):
        """Bytes of gradient in the bucket (the collective carries four more: `flag`)."""
        return self.n_grad * 4

    def reduce_ms(self, reset=True):
        """Mean HIP-event time of the timed collectives since the last call (None if none were timed).  Synchronises."""
        if not self._events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._events]
        if reset:
            self._events = []
        return sum(ms) / len(ms)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def gather_objects(obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank (a single process: [obj])."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, obj)
        return out
    return [obj]


def device_identity(device):
    """A string that names the physical GPU a rank computes on (uuid, else PCI address): two ranks reporting the same one share a
    GPU -- a launch that did not pin one process per device -- which a per-rank throughput figure alone would not show."""
    device = torch.device(device)
    if device.type != 'cuda':
        return f"{device.type}:{os.getpid()}"
    p = torch.cuda.get_device_properties(device)
    uid = getattr(p, "uuid", None)
    if uid is not None:
        return str(uid)
    return "pci-%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
