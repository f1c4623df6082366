cd /root/repo
mkdir -p /tmp/ddpout
for r in 0 1; do python tests/helpers/ddp_gpu_worker.py $r 2 29871 /tmp/ddpout full > /tmp/ddpout/log$r.txt 2>&1 & done
wait
python - <<'PY'
import torch, sys
sys.path.insert(0, "/root/repo")
a, b = torch.load("/tmp/ddpout/rank0.pt"), torch.load("/tmp/ddpout/rank1.pt")
from graphecho_amd.models.fpnseg import FPN
net = FPN([2,4,23,3], 4, 3)
off = 0
fa, fb = a["all"]["Net"], b["all"]["Net"]
bad = []
for n, p in net.named_parameters():
    k = p.numel()
    d = (fa[off:off+k] - fb[off:off+k]).abs().max().item()
    if d > 0: bad.append((n, off, k, d))
    off += k
print("total params", off, fa.numel(), "differing tensors:", len(bad))
for x in bad[:40]: print(x)
for name in a["all"]:
    print(name, torch.equal(a["all"][name], b["all"][name]))
print(a["losses"], b["losses"])
PY
