"""Two gloo ranks on one GPU through tests/helpers/ddp_gpu_worker.py: do the replicas of every model stay bit-identical?
usage: python tools/ddp_probe.py WORKLOAD VARIANT   (environment switches are inherited by the ranks)"""
import os, subprocess, sys, tempfile, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
worker = os.path.join(root, "tests", "helpers", "ddp_gpu_worker.py")
out = tempfile.mkdtemp()
port = str(29700 + os.getpid() % 200)
procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, out, sys.argv[1], sys.argv[2]]) for r in range(2)]
rc = [p.wait(timeout=900) for p in procs]
a, b = (torch.load(os.path.join(out, f"rank{r}.pt")) for r in range(2))
for name in a["all"]:
    d = (a["all"][name] - b["all"][name]).abs()
    print(f"{sys.argv[2]:10s} {name:8s} identical {torch.equal(a['all'][name], b['all'][name])}  max diff {d.max().item():.3e}  differing {int((d > 0).sum())} of {d.numel()}")
