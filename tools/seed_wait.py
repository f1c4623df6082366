"""How long does the temporal step block on the seed bank's spectral-clustering fits (graphecho_amd/cluster_pool.py)?
Config 5 in its stated dtype; prints ms/step and the host time spent inside ClusterPool.result per step.
usage: GE_CLUSTER_WORKERS=n python tools/seed_wait.py [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import cluster_pool
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
waited = [0.0, 0]
orig = cluster_pool.ClusterPool.result
def timed(self, jid):
    t0 = time.perf_counter(); r = orig(self, jid); waited[0] += time.perf_counter() - t0; waited[1] += 1; return r
cluster_pool.ClusterPool.result = timed
tr = GraphEchoTrainer(dev, workload="temporal", back_bone="VGG16", in_channel=1, num_classes=4, image_size=256, seed=0,
                      conv_precision=os.environ.get("PREC", "f16s"), clip_len=16, transport_method="sinkhorn_distance", seg_loss="cardiac", graphs="auto")
xs, ms = synthetic_batch(8, 1, 4, 256, dev, 1234); xt, _ = synthetic_batch(8, 1, 4, 256, dev, 4321)
def clip(seed, t=16):
    f, mk = synthetic_batch(t, 1, 4, 256, dev, seed)
    return (f.reshape(1, t, 1, 256, 256).permute(0, 2, 3, 4, 1).contiguous(), mk.reshape(1, t, 4, 256, 256).permute(0, 2, 3, 4, 1).contiguous())
cs, cm = clip(77); ct, _ = clip(78)
clips = {"source": cs, "target": ct, "masks": cm}
for _ in range(8):
    tr.step(xs, ms, xt, clips)
torch.cuda.synchronize()
waited[:] = [0.0, 0]
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(xs, ms, xt, clips)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"workers {os.environ.get('GE_CLUSTER_WORKERS', '4')}, GM_FIRST {os.environ.get('GE_GM_FIRST', '1')}: {1e3 * dt:.2f} ms/step, blocked on the fits {1e3 * waited[0] / steps:.2f} ms/step over {waited[1] / steps:.1f} results per step")
