"""Host + device time of one attention block of GModule (MultiHeadAttention, one head, 256 features) forward + backward, fused call
(csrc/ge_attention.hip) vs the composed ops: wall time per iteration over 200 back-to-back iterations, one sync at the end."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd.models import transformer as T
from graphecho_amd import functional as GF
from graphecho_amd.optim import FlatParams
dev = torch.device("cuda:0")
for N in (100, 300, 600):
    for fused in (True, False):
        T.FUSED_MHA = fused
        torch.manual_seed(0)
        mod = T.MultiHeadAttention(256, 1, dropout=0.1, version="v2").to(dev).train()
        fp = FlatParams([mod])
        x = torch.randn(N, 256, device=dev, requires_grad=True)
        GF.DIRECT_GRAD_ACCUM = True
        def it():
            o, a = mod(x, x, x)
            (o.sum() + a.sum()).backward()
        for _ in range(20): it()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): it()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"N={N} fused={fused}: host {t_host / 200 * 1e6:.0f} us / iteration, host+device {t_all / 200 * 1e6:.0f} us")
