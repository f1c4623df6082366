import sys, torch
sys.path.insert(0, "/root/repo")
from graphecho_amd import functional as GF
from graphecho_amd.trainer import GraphEchoTrainer, synthetic_batch
dev = torch.device("cuda:0")
for (B, S) in [(4, 128), (8, 256), (32, 256)]:
    x, m = synthetic_batch(B, 3, 4, S, dev, 5)
    tr = GraphEchoTrainer(dev, workload="fpn", image_size=S, seed=3)
    net = tr.network
    with torch.no_grad():
        l32, p32 = net(x)
        GF.CONV_PRECISION = "f16"
        l16, p16 = net(x)
        GF.CONV_PRECISION = "f32"
        # reference: fp32 path on a slightly perturbed input (same relative size as fp16 rounding) -> conditioning
        xp = x * (1 + 4.9e-4 * torch.randn_like(x))
        lp, pp = net(xp)
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    rms = lambda a, b: ((a - b).norm() / b.norm()).item()
    print(f"B{B} {S}: logits f16 max-rel {rel(l16,l32):.3e} rms-rel {rms(l16,l32):.3e} | perturbed-input fp32: max-rel {rel(lp,l32):.3e} rms {rms(lp,l32):.3e}")
    print("   pyramid f16 rms-rel", [f"{rms(a,b):.2e}" for a, b in zip(p16, p32)], " perturbed", [f"{rms(a,b):.2e}" for a, b in zip(pp, p32)])
