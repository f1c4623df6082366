"""Phase split of the bf16x3 conv kernel (GE_CONV_DEBUG bits: 8 no global loads, 16 no staging, 32 no MFMA phase) --
wrong results by design; run one configuration per process (the bits are read once)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
dev = torch.device("cuda:0")
GF.BX3_HYBRID = False
B, Cin, H, Cout, k = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (32, 256, 64, 256, 3))]
x = torch.randn(B, Cin, H, H, device=dev)
w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
GF.CONV_PRECISION = os.environ.get("PREC", "bf16x3")
cache = GF.PackCache()
for _ in range(3):
    GF.conv2d(x, w, None, 1, k // 2, 1, cache)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    GF.conv2d(x, w, None, 1, k // 2, 1, cache)
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 20
y = GF.conv2d(x, w, None, 1, k // 2, 1, cache)
GF.CONV_PRECISION = "f32"
yr = GF.conv2d(x, w, None, 1, k // 2, 1, GF.PackCache())
err = ((y - yr).abs().max() / yr.abs().max()).item()
print(f"err vs f32 kernel {err:.1e} ", end="")
print(f"dbg={os.environ.get('GE_CONV_DEBUG','0'):>3s} stages={os.environ.get('GE_X3_STAGES','1')} {t*1e3:8.1f} us  {2.0*B*H*H*Cout*Cin*k*k/t/1e9:7.1f} TF")
