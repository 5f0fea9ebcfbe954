#!/usr/bin/env python3
"""
bench.py -- fit-loop throughput of the MI355X-native differentiable-raster path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg1|cfg5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no torchrun environment: the process is a LAUNCHER (below, in front of the torch import) -- it starts N fresh
ranks of itself with a rendezvous on 127.0.0.1, relays rank 0's JSON line and exits non-zero if any rank fails.

A "step" = one optimisation step of the fit loop over this rank's batch: blend (MFMA) -> MVP chain ->
transform_clip -> rasterize -> interpolate -> texture -> antialias -> background + pixel loss, the whole
backward, the gradient all-reduce and Adam (reference src/torch/fit.py:524-618, batched).  The default
workload is BASELINE.json's configs[2] (9 views x 1080p, ~150 blendshapes, ~30k triangles, textured +
antialias forward/backward, 32 frames per GPU, Adam on weights + pose + texture): the configuration the
metric "fit-loop frames/sec (9-view 1080p)" and the 8-GPU config (32 frames/GPU) are quoted on.
`--workload cfg2` runs configs[1] (single frame, rasterize + interpolate only).  One "frame" = all 9
views of one time-frame.  Inputs (mesh, blendshapes, texture, 8-bit reference images) are synthetic and
resident in HBM before the timed region.  Weak scaling: every rank owns 32 frames.

Rank 0 prints ONE JSON line; see DESIGN.md section 6 for the roofline / cpu_baseline accounting.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def requested_gpus(argv):
    """The N of `--gpus N` / `--gpus=N` on a command line (1 when absent)."""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    return n


def wants_self_launch(argv, env=None):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a torchrun environment -- no WORLD_SIZE, or a wrapper's WORLD_SIZE=1 -- : the
    process is then a launcher, not a rank.  (Its children carry WORLD_SIZE = N: they are ranks.)"""
    env = os.environ if env is None else env
    try:
        world = int(env.get("WORLD_SIZE", "1") or "1")
    except ValueError:
        world = 1
    return requested_gpus(argv) > 1 and world <= 1


def self_launch(argv, timeout_s=None):
    """One process per GPU, started from here: this parent touches no GPU (it does not even import torch), picks a free rendezvous port on
    127.0.0.1 and starts N FRESH children `python bench.py <argv>` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set -- no exec of a
    process that initialised a device.  Rank 0's stdout is relayed as this process's stdout (its JSON line is the last line); the other
    ranks' stdout and every stderr go to stderr.  A child that fails, or the timeout (FPCDR_BENCH_LAUNCH_TIMEOUT seconds, default 1500),
    ends the others by their own PIDs and makes the launcher exit non-zero."""
    import socket
    import subprocess
    import threading
    n = requested_gpus(argv)
    timeout_s = float(os.environ.get("FPCDR_BENCH_LAUNCH_TIMEOUT", "1500")) if timeout_s is None else timeout_s
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, relays, last_line = [], [], [None]

    def relay(stream, to_stdout):
        for raw in iter(stream.readline, b""):
            line = raw.decode(errors="replace")
            if to_stdout:
                if line.strip():
                    last_line[0] = line.strip()
                sys.stdout.write(line)
                sys.stdout.flush()
            else:
                sys.stderr.write(line)
                sys.stderr.flush()
        stream.close()

    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdin=subprocess.DEVNULL,
                             stdout=subprocess.PIPE, stderr=None)
        procs.append(p)
        t = threading.Thread(target=relay, args=(p.stdout, r == 0), daemon=True)
        t.start()
        relays.append(t)
    deadline = time.monotonic() + timeout_s
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            failed = f"timeout after {timeout_s:.0f} s"
            break
        time.sleep(0.2)
    if failed:
        for p in procs:              # exactly the PIDs started above
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    for t in relays:
        t.join(timeout=5.0)
    if failed:
        sys.stderr.write(f"bench.py launcher (--gpus {n}): {failed}; the other ranks were stopped\n")
        return 1
    if last_line[0] is None or not last_line[0].startswith("{"):
        sys.stderr.write(f"bench.py launcher (--gpus {n}): rank 0 printed no JSON line\n")
        return 1
    return 0


if __name__ == "__main__" and wants_self_launch(sys.argv[1:]):
    sys.exit(self_launch(sys.argv[1:]))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling

# algorithmic bytes per pixel, C = colour channels, fp32, op-boundary tensors read / written once
# (SURVEY.md section 8d / BASELINE.md section 2).  rasterize forward writes rast only (rast_db is not
# materialised when mip-mapping is off); pixel_loss: colour + rast + 8-bit ref in, grad out.
def algorithmic_bytes_per_px(C, with_db):
    return {
        "fpcdr_rasterize_fwd": 16 + (16 if with_db else 0),
        "fpcdr_rasterize_bwd": 32 + (16 if with_db else 0),
        "fpcdr_interpolate_fwd": 24 + (32 if with_db else 0),
        "fpcdr_interpolate_bwd": 40 + (48 if with_db else 0),
        "fpcdr_texture_fwd": 8 + 4 * C + (16 if with_db else 0),
        "fpcdr_texture_bwd": 16 + 4 * C + (32 if with_db else 0),
        "fpcdr_antialias_fwd": 16 + 8 * C,
        "fpcdr_antialias_bwd": 8 * C,
        "fpcdr_pixel_loss": 4 * C + 16 + 1 + 4 * C,
        # fused objective (dense-equivalent: every pixel counted, although the sparse mode skips empty regions)
        "fpcdr_render_fwd": 16 + 4 * C,                 # rast + colour written
        "fpcdr_aa_loss_fwd": 4 * C + 16 + 1 + 4 * C,    # colour + rast + 8-bit ref in, d loss / d aa out
        "fpcdr_render_aa_bwd": 4 * C + 16,              # d loss / d aa + rast in, scatter only
        "fpcdr_render_loss_fwd": 16 + 4 * C + 1 + 4 * C,  # rast + colour + d loss / d aa written, 8-bit ref read
        # one-pass objective (value + gradient in one call): the id plane written by the rasteriser and read by the shading kernel,
        # the 8-bit reference read; nothing else of the image exists in HBM (deferred-pixel records: a few per cent, not counted)
        "fpcdr_objective_fwd": 4 + 4 + 1,
    }


COUNTERS_TAG = "r06"


def counters_file(workload, mip=False, channels=1):
    """The committed PMC passes of a workload (scripts/measure_round.sh): profiles/r06_counters_<workload>[_mip][_c3].json"""
    return os.path.join(ROOT, "profiles", f"{COUNTERS_TAG}_counters_{workload}{'_mip' if mip else ''}{f'_c{channels}' if channels != 1 else ''}.json")


def workload_config(workload, mip=False):
    """(FitConfig, frames per GPU, use a HIP graph) of a bench workload -- ONE definition for bench.py and for scripts/prof_objective.py,
    whose rocprofv3 passes the bench pairs with its own timings."""
    from fpc_diffrend_amd import fit
    cfg = fit.FitConfig(max_iter=80000, enable_mip=mip, frames_per_step=0, init_texture="random")
    if workload == "cfg2":       # BASELINE configs[1]: single frame, rasterize + interpolate only, no texture
        cfg.optimize_texture = False
        cfg.shading = "vertex"
    if workload == "ref":        # the reference's loop: one random (camera, frame) image per iteration
        cfg.frames_per_step = 1
        cfg.views_per_step = 1
    if workload == "cfg5":       # BASELINE configs[4]: 4K, per-vertex free-form offsets on top of the blendshapes
        cfg.mode = "combined"
        cfg.max_iter = 2              # the reference enables the free-form basis after max_iter / 2 (fit.py:603-608)
    fpg = {"cfg1": 4, "cfg2": 1, "cfg3": 32, "cfg5": 4, "ref": 8}[workload]
    return cfg, fpg, workload in ("cfg2", "ref")


def kernel_source_sha16():
    """sha256 over the kernel sources (csrc/*.hip, *.h, Makefile, include/fpcdr.h): the counters file records the one it was measured
    on (scripts/make_counters_json.py), and counters of other sources are not paired with this run's times."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "fpc_diffrend_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "fpc_diffrend_amd", "csrc", "*.h"))
                   + [os.path.join(ROOT, "fpc_diffrend_amd", "csrc", "Makefile"), os.path.join(ROOT, "include", "fpcdr.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
# kernels behind each C-ABI entry point (the PMC passes are per kernel)
# kernels of each entry point, by name prefix (template arguments vary with the instantiation the launch picked)
ENTRY_KERNELS = {
    # the nvdiffrast-style operators (cfg2's timed step; the stand-alone sweep of cfg3)
    "fpcdr_rasterize_fwd": ["k_init_ibox", "k_setup<false, false", "k_setup_clip<false", "k_bins<", "k_hint_dilate"],
    "fpcdr_rasterize_bwd": ["k_grad<"],
    "fpcdr_interpolate_fwd": ["k_interp_fwd"],
    "fpcdr_interpolate_bwd": ["k_interp_bwd"],
    "fpcdr_texture_fwd": ["k_tex_fwd", "k_tex_empty"],
    "fpcdr_texture_bwd": ["k_tex_bwd"],
    "fpcdr_antialias_fwd": ["k_sil", "k_aa_fill", "k_aa_fwd<"],
    "fpcdr_pixel_loss": ["k_pixel_loss"],
    "fpcdr_render_loss_fwd": ["k_sil2", "k_init_queue", "k_setup", "k_list_count<", "k_list_scan", "k_list_write<",
                              "k_bins_list<false, true, true", "k_bins_queue<false, true, true", "k_aa_fix_list<", "k_aa_fix_queue<"],
    "fpcdr_render_aa_bwd": ["k_render_aa_bwd<", "k_render_aa_bwd_list<", "k_render_aa_bwd_queue<"],
    "fpcdr_objective_fwd": ["k_init_objective", "k_setup<true, true", "k_setup_clip<true", "k_list_count<", "k_list_write_sum<",
                            "k_bins_list<false, false, false", "k_bins_queue<false, false, false", "k_shade_list<", "k_shade_queue<",
                            "k_shade_mip_list<", "k_shade_mip_queue<", "k_fix_list<", "k_fix_queue<", "k_fix_mip_list<", "k_fix_mip_queue<",
                            "k_objective_finish<"],
    "fpcdr_antialias_bwd": ["k_copy_f4_chunk", "k_aa_bwd_fix<"],
    "fpcdr_blend_fwd": ["k_blend_fwd_lds"],
}
# entry points that move their algorithmic bytes for EVERY pixel of the batch whatever the region hints say: rasterize forward writes
# every pixel, antialias backward is a full copy, the pixel loss reads every pixel.  The other operators skip the reads of empty bins:
# bytes/px x all pixels / time is then a dense-EQUIVALENT rate (it can exceed the HBM peak), not a roofline figure
MOVES_ALL = ("fpcdr_rasterize_fwd", "fpcdr_antialias_bwd", "fpcdr_pixel_loss")
OBJECTIVE_CALLS = ("fpcdr_objective_fwd", "fpcdr_render_loss_fwd", "fpcdr_render_aa_bwd")   # one-pass form / two-call form
N_SIMD = 1024           # 256 CUs x 4 SIMDs
F32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X dense f32 matrix peak (MI355X_MICROARCH.md)


def measured_counters(name, workload, n_images, C, t_ms=None, mip=False):
    """Per-launch PMC figures of one entry point from the committed rocprofv3 passes (COUNTERS_FILE: separate --pmc passes of
    scripts/prof_objective.py, scripts/measure_round.sh).  Returns (figures, None) or (None, reason): the figures are used only
    if the file describes this workload, was measured on THESE kernel sources (kernel_source_sha16) and -- t_ms given -- the
    kernels' durations in the passes agree with this run's HIP-event time of the call to 10 %.
    HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE: on gfx950 FETCH_SIZE reports exactly half the bytes of a coalesced stream of ANY
    width per lane (1, 2, 4, 8, 16 B: profiles/r02_fetch_calibration.txt); the kernels' gathers (vertices, texels) hit L2 and do
    not reach the counter."""
    path = counters_file(workload, mip, C)
    src = os.path.relpath(path, ROOT)
    try:
        with open(path) as f:
            t = json.load(f)
    except Exception as e:
        return None, f"no counters file for this workload ({src}: {e!r})"
    if name not in ENTRY_KERNELS:
        return None, f"no kernel list for {name}"
    if t.get("workload") != workload or t.get("images") != n_images or t.get("channels") != C or bool(t.get("mip", 0)) != bool(mip):
        return None, (f"{src} describes {t.get('workload')} / {t.get('images')} images / C = {t.get('channels')} / mip = {t.get('mip', 0)}, "
                      "not this run")
    if t.get("kernel_source_sha16") != kernel_source_sha16():
        return None, (f"{src} was measured on kernel sources {t.get('kernel_source_sha16')} (commit {t.get('git_head')}), this run's are "
                      f"{kernel_source_sha16()}: counters of other kernels are not paired with these times")
    tot = {"fetch_kb": 0.0, "write_kb": 0.0, "valu_insts": 0.0, "gui_active": 0.0, "mfma_busy": 0.0, "lds_conflict": 0.0, "lds_active": 0.0,
           "thread_cycles_valu": 0.0, "pass_ms": 0.0}
    for kname, entry in t["kernels"].items():
        short = kname[5:] if kname.startswith("void ") else kname
        c = entry.get("counters")
        if not c or not any(short.startswith(pre) for pre in ENTRY_KERNELS[name]):
            continue
        tot["fetch_kb"] += c.get("FETCH_SIZE", 0.0)
        tot["write_kb"] += c.get("WRITE_SIZE", 0.0)
        tot["valu_insts"] += c.get("SQ_INSTS_VALU", 0.0)
        tot["thread_cycles_valu"] += c.get("SQ_THREAD_CYCLES_VALU", 0.0)
        tot["gui_active"] += c.get("GRBM_GUI_ACTIVE", 0.0)
        tot["mfma_busy"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        tot["lds_conflict"] += c.get("SQ_LDS_BANK_CONFLICT", 0.0)
        tot["lds_active"] += c.get("SQ_LDS_IDX_ACTIVE", 0.0)
        d = list(entry.get("duration_us", {}).values())
        tot["pass_ms"] += (sum(d) / len(d) / 1e3) if d else 0.0
    if t_ms is not None and tot["pass_ms"] > 0 and abs(tot["pass_ms"] - t_ms) > 0.10 * t_ms:
        return None, (f"{src}: the kernels of {name} took {tot['pass_ms']:.3f} ms in the PMC passes, {t_ms:.3f} ms in this run "
                      "(> 10 % apart): counters not paired with these times")
    tot["hbm_bytes"] = (2.0 * tot["fetch_kb"] + tot["write_kb"]) * 1024.0
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of the launches = / 8; a VALU wave-instruction occupies its SIMD 4 cycles
    # (measured: scripts/micro/valu_rate_bench.hip -- every f32 / i32 / f64 / conversion instruction alike, v_rcp_f32 8)
    cyc = tot["gui_active"] / 8.0
    tot["valu_issue_frac"] = (tot["valu_insts"] * 4.0 / (N_SIMD * cyc)) if cyc else None
    tot["mfma_busy_frac"] = (tot["mfma_busy"] / (N_SIMD * cyc)) if cyc else None
    tot["lane_utilisation"] = (tot["thread_cycles_valu"] / (64.0 * tot["valu_insts"])) if tot["valu_insts"] and tot["thread_cycles_valu"] else None
    tot["source"] = {"file": src, "git_head": t.get("git_head"), "kernel_source_sha16": t.get("kernel_source_sha16")}
    return tot, None


def standalone_op_sweep(fitter, reps=3):
    """HIP-event time of every nvdiffrast-style operator (the separate C-ABI calls) at the bench's batch size, outside
    the timed region: the fit loop itself runs the fused objective, so this is where the per-operator roofline
    figures (incl. antialias backward, the kernel BASELINE.json's north star names) come from."""
    from fpc_diffrend_amd import _lib, camera, fit
    import fpc_diffrend_amd.ops as dr
    ft = fitter
    ids = slice(ft.frame_lo, ft.frame_hi)
    timer = _lib.KernelTimer()
    ctx = dr.RasterizeGLContext(output_db=False, device=ft.device)
    for rep in range(reps + 1):
        if rep == 1:
            torch.cuda.synchronize()
            _lib.TIMER = timer
        verts = ft.vertices(ids).reshape(ft.frame_hi - ft.frame_lo, -1, 3).detach()
        pos = camera.transform_clip(ft.mvp(ids).detach(), verts).requires_grad_(True)
        tex = ft.tex_opt.detach().clone().requires_grad_(True)
        rast, _ = dr.rasterize(ctx, pos, ft.pos_idx, ft.resolution)
        texc, _ = dr.interpolate(ft.uv[None], rast, ft.uv_idx)
        col = dr.texture(tex[None], texc, filter_mode='linear')
        aa = dr.antialias(col, rast, pos, ft.pos_idx)
        ref = ft.targets.reshape(-1, *ft.resolution)
        _, g = fit.pixel_loss_fused(aa, rast, ref)
        torch.autograd.backward([aa], [g])
        del rast, texc, col, aa, g, pos, tex
    summ = timer.summary()
    _lib.TIMER = None
    return summ


def count_launches(step_fn, steps=3):
    """GPU kernel launches per step (torch profiler over `steps` eager steps)."""
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(steps):
            step_fn()
        torch.cuda.synchronize()
    n = 0
    for e in prof.events():
        if str(getattr(e, "device_type", "")).endswith("CUDA") and not e.name.startswith(("Memcpy", "Memset")):
            n += 1
    return n / steps


def reference_shaped_step(device, steps=40, frames=8):
    """The reference's OWN run shape (main.py:28-30, fit.py:525-526): ONE random (camera, frame) image of 1600 x 1200 per
    iteration, a 1024^2 x 1 texture, 30k triangles -- what a user who only switches the import gets.  Two surfaces:
      drop_in    rasterize / interpolate / texture / antialias as four operators + the reference's torch.where / mean loss
      objective  ops.pixel_objective (the one-pass objective: value and gradient from one call), eager and replayed as HIP graphs
    Wall-clock ms per step between synchronisations, images/s, and GPU launches per step of the eager form."""
    from fpc_diffrend_amd import fit, scene
    sc = scene.cfg("ref", n_frames=frames)
    out = {"resolution": list(sc.resolution), "texture": list(sc.texture.shape), "triangles": int(sc.pos_idx.shape[0]),
           "frames_in_take": frames, "cameras": 9, "images_per_step": 1,
           "what": "one random (camera, frame) image per Adam step as reference fit.py:525-526; wall clock over %d steps" % steps}
    targets = None
    surfaces = (("drop_in", dict(fused_objective=False, fused_render=False, fused_loss=False), False),
                ("objective", {}, False), ("objective_hip_graph", {}, True))
    for name, kw, graph in surfaces:
        try:
            cfg = fit.FitConfig(max_iter=80000, frames_per_step=1, views_per_step=1, init_texture="random", hip_graph=graph, **kw)
            ft = fit.Fitter(sc, cfg, device=device, targets=targets)
            targets = ft.targets
            for _ in range(fit.Fitter.GRAPH_WARMUP + 3):
                ft.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                ft.step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            row = {"ms": 1e3 * dt, "images_per_s": 1.0 / dt}
            if not graph:
                row["launches_per_step"] = count_launches(ft.step)
            else:
                row["launches_per_step"] = "one graph replay (forward + backward + Adam in one launch each) + one staged host-to-device copy"
            out[name] = row
            del ft
        except Exception as e:   # never take the measurement down
            out[name] = {"error": repr(e)}
    return out


def cpu_baseline(sc, seconds_budget=30.0):
    """The oracle fit step (PyTorch-CPU software raster, cpu_baseline.kind = 'port') on the host cores, on a
    bounded sample of the same workload: ONE 1080p image (1 frame x 1 view) per step."""
    from oracle import fit as ofit
    from oracle import ops as oops
    oops.build()
    t0 = time.perf_counter()
    # torch's CPU kernels stop scaling long before the box's core count (r1: one step took 65 s on 256 threads, 1.7 s on 8):
    # time one step at a few thread counts and keep the fastest -- `cores` reports the threads actually used
    ncpu = os.cpu_count() or 1
    best = None
    for threads in sorted({min(ncpu, n) for n in (8, 16, 32, 64)}):
        dt_t, _, used_t = ofit.timed_steps(sc, cams=[3], frame_ids=[0], steps=1, threads=threads)
        if best is None or dt_t < best[0]:
            best = (dt_t, threads)
        if dt_t > 2.0 * best[0]:
            break
    threads = best[1]
    torch.set_num_threads(threads)
    dt, n_img, used, first = ofit.timed_steps(sc, cams=[3], frame_ids=[0], steps=1, threads=threads, keep_first=True)
    steps = 1
    # the same 1080p image through the HIP operators, from the oracle's own clip-space positions: the "L2 pixel err vs
    # ref" half of BASELINE.json's metric, with the oracle as the checker (integer buffer must match exactly)
    parity = None
    try:
        import fpc_diffrend_amd.ops as dr
        from fpc_diffrend_amd import fit as gfit
        dev = torch.device("cuda", torch.cuda.current_device())
        st = first["state"]
        ctx = dr.RasterizeGLContext(device=dev)
        pos = first["pos_clip"].to(dev).contiguous()
        tri = st.pos_idx.to(dev).to(torch.int32).contiguous()
        rast, _ = dr.rasterize(ctx, pos, tri, sc.resolution)
        texc, _ = dr.interpolate(st.uv.to(dev)[None].contiguous(), rast, st.uv_idx.to(dev).to(torch.int32).contiguous())
        col = dr.antialias(dr.texture(first["tex"].to(dev)[None].contiguous(), texc, filter_mode='linear'), rast, pos, tri)
        img = torch.where(rast[..., 3:] > 0, col, torch.tensor(gfit.BACKGROUND, device=dev)).cpu()
        mism = int((rast[..., 3].cpu() != first["ids"]).sum())
        err = float((img.double() - first["image"].double()).norm() / first["image"].double().norm())
        parity = {"id_mismatches": mism, "image_rel_l2": err, "pixels": int(first["ids"].numel()),
                  "covered_pixels": int((first["ids"] > 0).sum()),
                  "what": "rasterize / interpolate / texture / antialias / background on the oracle's clip positions vs the oracle image"}
    except Exception as e:   # never take the measurement down
        parity = {"error": repr(e)}
    if dt < seconds_budget / 3:
        k = max(1, int(seconds_budget / 3 / dt))
        dt, n_img, used = ofit.timed_steps(sc, cams=[3], frame_ids=[0], steps=k, threads=threads)
        steps = k
    n_views = 9
    return {"value": (n_img / n_views) / dt, "unit": "frames/s", "cores": used, "kind": "port", "parity_1080p": parity,
            "sample": f"{steps} full fit step(s) (forward + backward + Adam) of 1 frame x 1 view at "
                      f"{sc.resolution[1]}x{sc.resolution[0]} on the same mesh/rig, {dt:.2f} s per image, scaled linearly to "
                      f"{n_views} views per frame ({used} of the box's {ncpu} hardware threads: the fastest of 8 / 16 / 32 / 64); "
                      f"wall {time.perf_counter() - t0:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # (a timed region of ~0.15 s at cfg3: long enough for a GPU-busy sampler to land in)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=["cfg1", "cfg2", "cfg3", "cfg5", "ref"],
                    help="ref: the reference's own run shape -- ONE random (camera, frame) image of 1600 x 1200 per step (main.py:28-30, "
                         "fit.py:525-526), replayed as HIP graphs")
    ap.add_argument("--early-tex-reduce", action="store_true",
                    help="all-reduce the texture gradient on its own as soon as the backward kernel has produced it (dist.EarlyReduce), "
                         "beside the rest of the backward pass; eager steps only")
    ap.add_argument("--no-reference-shaped-step", action="store_true", help="skip the reference_shaped_step extra of the default run")
    ap.add_argument("--frames-per-gpu", type=int, default=0)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--mip", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--time-all-calls", action="store_true", help="HIP-event timing of every C-ABI call, not only the pixel kernels")
    ap.add_argument("--fill", type=float, default=0.0,
                    help="head height as a fraction of the image height (default: the scene generator's 0.6 = 11.5 %% of a 1080p frame covered)")
    ap.add_argument("--graph", type=int, default=-1,
                    help="1: replay each step as two HIP graphs (FitConfig.hip_graph); default: on for cfg2 (launch-bound), off otherwise")
    args = ap.parse_args()

    from fpc_diffrend_amd import _lib, dist as fdist, fit, scene

    rank, world, local_rank = fdist.init()
    assert world == args.gpus, (f"--gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` (it launches "
                                "its own ranks) or through torch.distributed.run with --nproc-per-node equal to --gpus")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    _lib.load()
    # a multi-GPU line is only worth printing when it is what it claims: N ranks, RCCL (backend "nccl"), N distinct devices.  Checked
    # HERE, before any work, and loudly (FPCDR_DIST_BACKEND=gloo rehearsals on one GPU set FPCDR_BENCH_ALLOW_ANY_BACKEND=1)
    if os.environ.get("FPCDR_BENCH_ALLOW_ANY_BACKEND", "0") != "1":
        fdist.check_world(args.gpus, device, require_backend="nccl" if world > 1 else None)

    cfg, fpg_default, graph_default = workload_config(args.workload, args.mip)
    fpg = args.frames_per_gpu or fpg_default
    n_frames = fpg * world
    sc = scene.cfg(args.workload, n_frames=n_frames)
    if args.fill:
        sc.cams = scene.make_cameras(sc.resolution, fill=args.fill)
    if args.channels != 1:
        import numpy as np
        sc.texture = np.repeat(sc.texture, args.channels, axis=2)[:, :, :args.channels].copy()
    use_graph = bool(args.graph) if args.graph >= 0 else graph_default
    cfg.hip_graph = use_graph
    bucket = None
    fitter = fit.Fitter(sc, cfg, device=device, rank=rank, world=world)
    early = [fitter.tex_opt] if (args.early_tex_reduce and world > 1 and not use_graph) else ()
    bucket = fdist.GradBucket(fitter.params, device, timed=True, early=early)     # HIP events around the collective alone
    fitter.reduce_fn = bucket if world > 1 else None
    H, W = fitter.resolution
    n_cam = len(fitter.cam_idxs)
    C = sc.texture.shape[2]
    with torch.no_grad():       # covered share of the frame at the starting pose (the sparse objective's speed depends on it)
        import fpc_diffrend_amd.ops as _dr
        ids0 = slice(fitter.frame_lo, fitter.frame_lo + 1)
        pos0 = fit.transform_clip_batched(fitter.mvp(ids0), fitter.vertices(ids0).reshape(1, -1, 3))
        rast0, _ = _dr.rasterize(_dr.RasterizeGLContext(output_db=False, device=device), pos0, fitter.pos_idx, fitter.resolution)
        coverage = float((rast0[..., 3] > 0).float().mean())
        del rast0, pos0

    timer = None
    if not args.no_kernel_timer:
        # inside the timed region only the entry points with per-pixel traffic are bracketed by HIP events (an event pair
        # costs the stream ~15 us; eleven calls per step would add 3 % to the step being measured); the small calls'
        # durations are in the rocprofv3 summaries under profiles/
        timer = _lib.KernelTimer(names=None if args.time_all_calls else algorithmic_bytes_per_px(1, False).keys())
    timer_steps = args.steps
    if use_graph:
        # HIP events cannot be read back from inside a graph: the per-kernel durations come from an eager pass of the
        # same steps BEFORE the timed region (same kernels, same arguments); the timed region then replays the graphs
        fitter.use_graph = False
        timer_steps = max(args.steps, fitter.GRAPH_WARMUP)
        _lib.TIMER = timer
        for _ in range(timer_steps):
            fitter.step()
        _lib.TIMER = None
        fitter.use_graph = True
        for _ in range(max(args.warmup, 2)):      # one eager step that fixes the parameter set, then capture + replay
            fitter.step()
    else:
        for _ in range(args.warmup):
            fitter.step()
    torch.cuda.synchronize()
    fdist.barrier()
    torch.cuda.synchronize()
    _lib.TIMER = None if use_graph else timer
    t0 = time.perf_counter()
    loss = None
    for _ in range(args.steps):
        loss = fitter.step()
    torch.cuda.synchronize()
    fdist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _lib.TIMER = None
    elapsed_local = elapsed
    elapsed = fdist.max_over_ranks(elapsed, device)

    # a "frame" = all views of one time-frame; the ref workload renders ONE image (one view of one frame) per step and rank
    frames_step = 1 if args.workload == "ref" else fpg
    n_views_step = 1 if args.workload == "ref" else n_cam
    frames_total = frames_step * world * args.steps
    value = frames_total / elapsed
    allreduce_ms = bucket.reduce_ms() if world > 1 else None
    out = {
        "metric": "fit-loop frames/sec (9-view 1080p)" if args.workload in ("cfg2", "cfg3") else
                  ("fit-loop images/sec (reference run shape: 1 view x 1 frame of 1600x1200 per step)" if args.workload == "ref"
                   else f"fit-loop frames/sec ({args.workload})"),
        "value": value, "unit": "images/s" if args.workload == "ref" else "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n_cam}-view {W}x{H}, K={sc.blendshapes.shape[1]} blendshapes, "
                               f"T={sc.pos_idx.shape[0]} triangles, {frames_step} frames/GPU/step ({frames_step * n_views_step} images), "
                               + ("rasterize + interpolate only (per-vertex grey), Adam on weights+pose"
                                  if args.workload == "cfg2" else
                                  f"textured C={C} + antialias fwd/bwd{' + mip' if args.mip else ''}, Adam on weights+pose+texture"
                                  + (" + free-form vertex offsets" if args.workload == "cfg5" else "")),
                   "coverage": coverage, "frames_per_gpu": fpg, "views": n_cam, "resolution": [H, W], "triangles": int(sc.pos_idx.shape[0]),
                   "blendshapes": int(sc.blendshapes.shape[1]), "texture": list(sc.texture.shape),
                   "parallelism": (f"dp{world} (frames sharded, one RCCL all-reduce of {bucket.nbytes / 1e6:.1f} MB per step)" if world > 1
                                   else "one GPU, no collective")},
        # the data-parallel exchange of one step: HIP-event time of the RCCL all-reduce of the flat gradient bucket on this rank
        # (mean over the timed steps; includes waiting for the slowest rank) and the bucket's size; null on one GPU
        "allreduce_ms": allreduce_ms,
        # this rank's step time with the collective's own HIP-event time taken out: what a rank spends computing.  A multi-GPU line whose
        # ms_per_step exceeds the one-GPU line by more than allreduce_ms lost the rest to waiting for the slowest rank (load imbalance:
        # per_rank[*].occupied_bins) or to launch-side effects, not to xGMI
        "step_ms_without_allreduce": (1000.0 * elapsed_local / args.steps - allreduce_ms) if allreduce_ms is not None else 1000.0 * elapsed_local / args.steps,
        "allreduce_ms_max_over_ranks": fdist.max_over_ranks(allreduce_ms, device) if allreduce_ms is not None else None,
        "bucket_bytes": bucket.nbytes,
        "early_tex_reduce_bytes": sum(p.numel() * 4 for p in early) if early else 0,
        "final_loss": float(loss) if loss is not None else None,
        "hbm_allocated_peak_GB": torch.cuda.max_memory_allocated(device) / 1e9,
    }
    # what a reader of a multi-GPU line needs to trust it: the collective backend, the physical device of every rank (two ranks on one
    # GPU, or a gloo fallback, must be visible), each rank's own time per step (load imbalance) and the bins its frames occupied
    import torch.distributed as tdist_
    import fpc_diffrend_amd.ops as dr_ops_
    my_bins = None
    for key_, h_ in dr_ops_._list_hints.items():
        if key_[0] == 'onepass' and key_[2] == fpg * n_cam:
            my_bins = {"rasteriser": int(h_.host[2]), "shaded": int(h_.host[3]), "deferred": int(h_.host[0])}
    per_rank = fdist.gather_objects({"rank": rank, "device": fdist.device_identity(device), "ms_per_step": 1000.0 * elapsed_local / args.steps,
                                     "occupied_bins": my_bins, "allreduce_ms": allreduce_ms})
    out["backend"] = tdist_.get_backend() if (tdist_.is_available() and tdist_.is_initialized()) else None
    out["ranks_seen"] = len({r["device"] for r in per_rank})
    out["ms_per_step_rank_min_max"] = [min(r["ms_per_step"] for r in per_rank), max(r["ms_per_step"] for r in per_rank)]
    out["per_rank"] = per_rank
    if rank == 0 and timer is not None:
        bpp = algorithmic_bytes_per_px(C, args.mip)
        npix = frames_step * n_views_step * H * W

        def table_of(summ):
            """Per entry point: calls, mean HIP-event ms and its per-pixel rate.  Only the calls that move their bytes for every pixel
            of the batch (MOVES_ALL) carry it as algorithmic_GBps; the operators that skip the reads of empty bins through the region
            hints, and the objective calls that run over bin lists, carry a dense_equivalent_GBps (it may exceed the HBM peak)."""
            table = {}
            for name, (calls, ms) in summ.items():
                per = ms / max(calls, 1)
                row = {"calls": calls, "avg_ms": per}
                if name in bpp:
                    key = "algorithmic_GBps" if name in MOVES_ALL else "dense_equivalent_GBps"
                    row[key] = bpp[name] * npix / (per * 1e-3) / 1e9
                    row["bytes_per_px"] = bpp[name]
                    if name not in MOVES_ALL and name not in OBJECTIVE_CALLS:
                        row["note"] = "region hints skip the reads of empty bins: not a roofline figure"
                table[name] = row
            return table

        table = table_of(timer.summary())
        out["kernels"] = table
        out["fpcdr_ms_per_step"] = sum(v["avg_ms"] * v["calls"] for v in table.values()) / timer_steps
        out["hip_graph"] = use_graph
        if use_graph:
            out["kernels_note"] = "per-kernel HIP-event durations from an eager pass before the timed region; the timed steps replay two HIP graphs"
        px_ops = {k: v for k, v in table.items() if "bytes_per_px" in v}
        # pixels the SPARSE objective has to touch: 1024 per bin on its lists (counts of the last step, ops._ListHints)
        import fpc_diffrend_amd.ops as dr_ops
        sparse_px = {}
        for key, h in dr_ops._list_hints.items():
            if key[0] == 'onepass' and key[2] == fpg * n_cam:      # one-pass objective: live bins of the rasteriser, occupied bins
                _, _, n_bins, n_occ = (int(v) for v in h.host[:4].tolist())
                sparse_px = {"fpcdr_objective_fwd": n_occ * 1024}
                out["config"]["occupied_bins"] = {"rasteriser": n_bins, "shaded": n_occ,
                                                  "of": fpg * n_cam * ((H + 31) // 32) * ((W + 31) // 32)}
            elif key[0] != 'onepass' and key[1] == fpg * n_cam:
                n_bwd, _, n_bins, n_fix = (int(v) for v in h.host.tolist())
                sparse_px = {"fpcdr_render_loss_fwd": n_bins * 1024, "fpcdr_render_aa_bwd": n_bwd * 1024}
                out["config"]["occupied_bins"] = {"rasteriser": n_bins, "antialias_fix": n_fix, "backward": n_bwd,
                                                  "of": fpg * n_cam * ((H + 31) // 32) * ((W + 31) // 32)}
        if px_ops:
            def hbm_roofline(name_):
                """HBM roofline of one per-pixel entry point: algorithmic bytes / HIP-event time against the 8 TB/s peak, the PMC
                traffic of its kernels beside it, and what the counters say bounds it.  `achieved` / `frac` exist only where the
                algorithmic bytes are known: the calls of MOVES_ALL (every pixel of the batch) and the sparse objective (1 024 pixels x
                the bins on its list, counted live).  A hinted operator skips the reads of an unknown share of the batch: null, with
                the dense-equivalent rate and the measured traffic beside it."""
                t_s_ = px_ops[name_]["avg_ms"] * 1e-3
                pmc_, why_ = measured_counters(name_, args.workload, fpg * n_cam, C, t_ms=px_ops[name_]["avg_ms"], mip=args.mip)
                known = name_ in MOVES_ALL or name_ in sparse_px
                alg_ = bpp[name_] * sparse_px.get(name_, npix) if known else None
                a_ = alg_ / t_s_ / 1e9 if known else None
                # what the counters say limits the call: the larger of the two shares -- vector issue slots, HBM bytes against the peak --
                # if it is at least half; a call that fills neither (a one-frame batch, a chain of dependent gathers) is bound by
                # latency / occupancy and says so
                hbm_share_ = (pmc_["hbm_bytes"] / t_s_ / 1e9 / HBM_PEAK_GBS) if pmc_ else 0.0
                valu_share_ = (pmc_.get("valu_issue_frac") or 0.0) if pmc_ else 0.0
                bound_ = None if not pmc_ else ("valu-issue" if valu_share_ >= 0.5 and valu_share_ > hbm_share_ else
                                                ("hbm" if hbm_share_ >= 0.5 else "latency (neither vector issue nor HBM is half used)"))
                return {"kernel": name_,
                        # what the counters say limits the call; achieved / peak / frac are the HBM figures the metric asks for
                        # (algorithmic bytes against the 8 TB/s peak), whatever the bound
                        # (no counters for this workload / these sources: no evidence either way -> null, not "hbm")
                        "bound": bound_,
                        "achieved": a_, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (a_ / HBM_PEAK_GBS) if known else None,
                        "ms": px_ops[name_]["avg_ms"], "algorithmic_bytes": alg_,
                        "traffic": pmc_["hbm_bytes"] if pmc_ else None,
                        "traffic_GBps": (pmc_["hbm_bytes"] / t_s_ / 1e9) if pmc_ else None,
                        "traffic_frac": (pmc_["hbm_bytes"] / t_s_ / 1e9 / HBM_PEAK_GBS) if pmc_ else None,
                        "valu_issue_frac": pmc_.get("valu_issue_frac") if pmc_ else None,
                        "traffic_source": pmc_["source"] if pmc_ else None,
                        "traffic_unavailable": why_,
                        "dense_equivalent_GBps": bpp[name_] * npix / t_s_ / 1e9}, pmc_

            dom = max(px_ops, key=lambda k: px_ops[k]["avg_ms"] * px_ops[k]["calls"])
            t_s = px_ops[dom]["avg_ms"] * 1e-3
            out["roofline"], pmc = hbm_roofline(dom)
            if dom == "fpcdr_objective_fwd" and sparse_px.get(dom):
                # continuity with rounds 1-3, whose two calls wrote and re-read rast / colour / d loss/d colour: the SAME work priced at
                # their 25 + 20 algorithmic bytes per pixel of the listed bins.  The one-pass call does not move those bytes -- that is
                # the point of it -- so this is a rate of work done, comparable with the earlier rounds' frac, not traffic
                eq_bytes = (16 + 4 * C + 1 + 4 * C + 4 * C + 16) * sparse_px[dom]
                out["roofline"]["two_call_equivalent"] = {"bytes_per_px": 16 + 4 * C + 1 + 4 * C + 4 * C + 16, "bytes": eq_bytes,
                                                          "GBps": eq_bytes / t_s / 1e9, "frac": eq_bytes / t_s / 1e9 / HBM_PEAK_GBS,
                                                          "note": "the work of fpcdr_render_loss_fwd + fpcdr_render_aa_bwd (round 3: 45 B/px "
                                                                  "on the listed bins, frac 0.14-0.15 per call) done in this call's time"}
            if dom == "fpcdr_objective_fwd":
                out["roofline"]["note"] = ("fpcdr_objective_fwd computes value AND gradient in one call and moves 9 B/px (id plane out and in, "
                                           "8-bit reference): it is bound by vector issue, not by HBM -- see roofline_valu.  "
                                           "achieved = algorithmic bytes of the SPARSE call (B/px x 1024 px x the bins on its list, counted "
                                           "live) / HIP-event time inside the timed region; traffic = 2 x FETCH_SIZE + WRITE_SIZE of the "
                                           "call's kernels from the committed PMC passes named in traffic_source (null, with the reason in "
                                           "traffic_unavailable, when they were measured on other kernel sources or their durations differ "
                                           "from this run's by more than 10 %); dense_equivalent counts every pixel of the batch although "
                                           "80 % are never touched")
            elif dom in MOVES_ALL:
                out["roofline"]["note"] = (f"{dom}: achieved = bytes/px x every pixel of the batch / HIP-event time (the call moves them "
                                           "whatever the region hints say); traffic = 2 x FETCH_SIZE + WRITE_SIZE of its kernels from the "
                                           "committed PMC passes named in traffic_source")
            else:
                out["roofline"]["note"] = (f"{dom} skips the reads of the bins the region hint calls empty: its algorithmic bytes are not "
                                           "bytes/px x the batch, so achieved / frac are null; dense_equivalent_GBps prices every pixel "
                                           "(not a roofline figure, it may exceed the peak); traffic / traffic_frac are the measured HBM "
                                           "bytes of its kernels (committed PMC passes) against the 8 TB/s peak")
            # the two calls of the objective take nearly the same time and swap places from run to run: both are reported
            out["roofline_objective_calls"] = {n_: hbm_roofline(n_)[0] for n_ in OBJECTIVE_CALLS if n_ in px_ops}
            if pmc and pmc.get("valu_issue_frac") is not None:
                out["roofline_valu"] = {"kernel": dom, "bound": "valu-issue", "achieved": pmc["valu_insts"] / t_s / 1e9,
                                        "unit": "G wave-instructions/s", "peak": N_SIMD * (pmc["gui_active"] / 8.0) / 4.0 / t_s / 1e9 if t_s else None,
                                        "frac": pmc["valu_issue_frac"], "lane_utilisation": pmc.get("lane_utilisation"),
                                        "lds_bank_conflict_share": (pmc["lds_conflict"] / pmc["lds_active"]) if pmc["lds_active"] else None,
                                        "source": pmc["source"],
                                        "note": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): share of the chip's vector "
                                                "issue slots (every VALU instruction holds its SIMD for 4 cycles on gfx950 -- f32, i32 "
                                                "multiplies, f64, conversions alike; v_rcp_f32 8; packed f32 4.6 for two results: "
                                                "profiles/r03_valu_rate_bench.txt); lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x "
                                                "SQ_INSTS_VALU), the share of lanes doing work in an issued instruction"}
            # vector wave-instructions per 64 pixels of the two objective calls (the figure an HBM-bound kernel could afford ~120 of)
            vpp = {}
            for name_ in OBJECTIVE_CALLS:
                if name_ in table and sparse_px.get(name_):
                    pm, _ = measured_counters(name_, args.workload, fpg * n_cam, C, t_ms=table[name_]["avg_ms"], mip=args.mip)
                    if pm and pm["valu_insts"]:
                        vpp[name_] = pm["valu_insts"] / (sparse_px[name_] / 64.0)
            if vpp:
                out["valu_insts_per_px"] = dict(vpp, note="SQ_INSTS_VALU of the call's kernels / (pixels on its bin list / 64): vector "
                                                          "wave-instructions issued per 64 pixels")
        bl, _ = measured_counters("fpcdr_blend_fwd", args.workload, fpg * n_cam, C, mip=args.mip)
        if bl and bl.get("mfma_busy_frac") is not None:
            Mrows, Kb = 3 * (sc.v_base.shape[0] // 3), sc.blendshapes.shape[1]
            out["roofline_blend_mfma"] = {"kernel": "k_blend_fwd_lds (V = v_base + W B^T, v_mfma_f32_32x32x2_f32)", "bound": "mfma",
                                          "flops": 2.0 * Mrows * Kb * fpg, "peak": F32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "mfma_busy_frac": bl["mfma_busy_frac"],
                                          "hbm_bytes": bl["hbm_bytes"], "source": bl["source"],
                                          "note": "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) from "
                                                  "the committed PMC pass; the contraction is 0.43 GFLOP against one 27 MB read of the "
                                                  "blendshape basis: it is bounded by that read and by per-tile latency, not by the "
                                                  "matrix cores"}
        if not args.mip and args.workload in ("cfg1", "cfg3") and world == 1:   # (scaling runs: all ranks leave together)
            try:
                st = table_of(standalone_op_sweep(fitter))
                out["kernels_standalone_ops"] = st
                if "fpcdr_antialias_bwd" in st:
                    a = st["fpcdr_antialias_bwd"]["algorithmic_GBps"]
                    ab, ab_why = measured_counters("fpcdr_antialias_bwd", args.workload, fpg * n_cam, C, t_ms=st["fpcdr_antialias_bwd"]["avg_ms"])
                    out["roofline_antialias_bwd"] = {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                     "frac": a / HBM_PEAK_GBS,
                                                     "traffic": ab["hbm_bytes"] if ab else None, "traffic_source": ab["source"] if ab else None,
                                                     "traffic_unavailable": ab_why,
                                                     "note": "stand-alone dr.antialias backward at the same batch, outside the timed region"}
            except Exception as e:   # the sweep must never take the measurement down
                out["kernels_standalone_ops"] = {"error": repr(e)}
            if world == 1:
                # the same step through the DROP-IN surface only: the reference's render() on the four nvdiffrast-style
                # operators and its torch loss (no fused objective) -- what a user gets by switching the import alone
                try:
                    _lib.TIMER = None
                    cfg_d = fit.FitConfig(max_iter=80000, frames_per_step=0, init_texture="random", fused_objective=False,
                                          fused_render=False, fused_loss=False)
                    ft_d = fit.Fitter(sc, cfg_d, device=device, targets=fitter.targets)
                    for _ in range(2):
                        ft_d.step()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        ft_d.step()
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / 3
                    out["drop_in_path"] = {"ms_per_step": 1e3 * dt, "frames_per_s": fpg / dt,
                                           "what": "rasterize / interpolate / texture / antialias as four separate operators + "
                                                   "the reference's torch.where / mean loss, same batch, outside the timed region"}
                    del ft_d
                except Exception as e:
                    out["drop_in_path"] = {"error": repr(e)}
    if rank == 0 and world == 1 and args.workload in ("cfg3", "ref") and not args.no_reference_shaped_step:
        del fitter
        torch.cuda.empty_cache()
        out["reference_shaped_step"] = reference_shaped_step(device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(sc)
        except Exception as e:  # the baseline must never take the measurement down
            out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {e!r}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as tdist
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
