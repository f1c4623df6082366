"""The 1x1 conv layers of the config-2 step one by one: forward (+ fused BatchNorm moments) and data gradient.
Prints us per launch, TFLOP/s, and the launch time a perfect kernel would need -- max(flops / 147 TFLOP/s sustained fp32 MFMA,
algorithmic bytes / 5.5 TB/s) -- over the measured time ("of roof").  usage: bench_conv1x1.py [batch [nostats]]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
dev = torch.device("cuda:0")
_big = torch.randn(8192, 8192, device=dev)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2): torch.mm(_big, _big)       # ~20 ms of GPU work queued first: the host runs ahead, so launches of a few
    s.record()                                    # tens of us are not timed at the host's pace
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
STATS = not (len(sys.argv) > 2 and sys.argv[2] == "nostats")
tot = [0.0, 0.0]
for (Cin, H, Cout) in [(64, 64, 64), (64, 64, 256), (256, 64, 64), (256, 64, 128), (128, 32, 512), (512, 32, 128), (256, 64, 256),
                       (512, 32, 256), (512, 64, 256), (256, 32, 512), (256, 16, 1024), (1024, 16, 256), (512, 16, 1024), (1024, 16, 512)]:
    x = torch.randn(B, Cin, H, H, device=dev, requires_grad=True)
    w = (torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05)
    cache = GF.PackCache()
    t_f = timeit(lambda: GF.conv2d(x.detach(), w, None, 1, 0, 1, cache, STATS))
    y = GF.conv2d(x, w, None, 1, 0, 1, cache)
    g = torch.randn_like(y)
    t_d = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
    fl = 2.0 * B * H * H * Cout * Cin
    byt = 4.0 * B * H * H * (Cin + Cout)
    roof = max(fl / 147e12, byt / 5.5e12)
    tot[0] += t_f; tot[1] += t_d
    print(f"B{B} {Cin:4d}->{Cout:4d} @{H}x{H} ({GF.lib.ge_last_conv_kernel().decode()[:24]}): fwd+stats {t_f*1e6:6.1f} us {fl/t_f/1e12:6.1f} TF ({roof/t_f:4.2f} of roof) | dgrad {t_d*1e6:6.1f} us {fl/t_d/1e12:6.1f} TF ({roof/t_d:4.2f})")
print(f"sum fwd {tot[0]*1e3:.3f} ms, dgrad {tot[1]*1e3:.3f} ms")
