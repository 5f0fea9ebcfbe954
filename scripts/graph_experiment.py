"""Experiment: whole Adam step captured in one HIP graph (torch.cuda.CUDAGraph) vs eager launches."""
import sys, time
import torch
from fpc_diffrend_amd import fit, scene

def run(workload, nf):
    sc = scene.cfg(workload, n_frames=nf)
    cfg = fit.FitConfig(max_iter=80000, frames_per_step=0, init_texture="random")
    if workload == "cfg2":
        cfg.optimize_texture = False
        cfg.shading = "vertex"
    ft = fit.Fitter(sc, cfg, device='cuda')
    # capturable optimiser with tensor learning rates
    groups = [{"params": g["params"], "lr": torch.tensor(float(g["lr"]), device='cuda')} for g in ft.optimizer.param_groups]
    ft.optimizer = torch.optim.Adam(groups, lr=torch.tensor(cfg.lr_base, device='cuda'), capturable=True)
    def step_body():
        loss = ft.loss_and_backward(ft.pick_frames())
        ft.optimizer.step()
        with torch.no_grad():
            ft.q_opt /= torch.sum(ft.q_opt ** 2) ** 0.5
            ft.per_frame_q /= torch.sum(ft.per_frame_q ** 2) ** 0.5
        return loss
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step_body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        l = step_body()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 10
    g = torch.cuda.CUDAGraph()
    ft.optimizer.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        loss = step_body()
    torch.cuda.synchronize()
    ls = []
    for _ in range(3):
        g.replay(); ls.append(float(loss))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / 10
    print(workload, nf, "eager %.3f ms  graphed %.3f ms" % (eager * 1e3, graphed * 1e3), "losses", ls, float(l), flush=True)

run("cfg2", 1)
run("cfg3", 32)
