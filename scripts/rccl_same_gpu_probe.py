"""Does RCCL accept two ranks on ONE device (the shape of tests/test_gpu_dist.py on the 1-GPU box)?  Always exits 0;
prints / writes the outcome."""
import os
import socket
import subprocess
import sys

CHILD = r"""
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend=os.environ['PROBE_BACKEND'], rank=int(os.environ['RANK']), world_size=2)
t = torch.full((4,), float(int(os.environ['RANK']) + 1), device='cuda')
dist.all_reduce(t)
torch.cuda.synchronize()
print('rank', os.environ['RANK'], os.environ['PROBE_BACKEND'], 'sum', t.tolist(), flush=True)
dist.destroy_process_group()
"""


def main():
    out = {}
    for backend in ("nccl", "gloo"):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = []
        for r in range(2):
            env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="2", PROBE_BACKEND=backend,
                       HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        ok = True
        texts = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=120)
            except subprocess.TimeoutExpired:
                p.kill(); o = b"TIMEOUT"
            texts.append(o.decode(errors="replace")[-1500:])
            ok = ok and p.returncode == 0
        out[backend] = ok
        print(backend, "OK" if ok else "FAILED", texts)
    print("PROBE", out)


if __name__ == "__main__":
    main()
