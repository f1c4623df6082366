"""One Winograd layer, forward only, timed (tuning builds: GE_LIB_PATH=graphecho_amd/csrc/variants/lib_<tag>.so; PMC passes).
usage: bench_wino_one.py [B Cin Cout S [iters]]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd._lib import lib, check

dev = torch.device("cuda:0")
B, Cin, Cout, S = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 256, 256, 64))]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 30
p = lambda t: None if t is None else t.data_ptr()
torch.manual_seed(0)
x = torch.randn(B, Cin, S, S, device=dev)
w = torch.randn(Cout, Cin, 3, 3, device=dev) / (3 * Cin ** 0.5)
bias = torch.randn(Cout, device=dev)
u = torch.empty(lib.ge_wino3x3_weight_floats(Cin, Cout), device=dev)
check(lib.ge_wino3x3_pack_weight(p(w), p(u), Cout, Cin, 0, None), "pack")
ws = torch.empty(max(1, lib.ge_wino3x3_workspace(B, Cin, Cout, S, S)), device=dev)
y = torch.empty(B, Cout, S, S, device=dev)
fw = lambda: check(lib.ge_wino3x3_fwd(p(x), p(u), p(bias), None, p(y), None, p(ws), B, Cin, Cout, S, S, None), "fwd")
for _ in range(5):
    fw()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fw()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / iters * 1e-3
fl = 2.0 * B * S * S * Cout * Cin * 9
print(f"{os.path.basename(os.environ.get('GE_LIB_PATH', 'production'))}: {Cin}->{Cout} @{S}x{S}x{B} splits {lib.ge_wino3x3_splits(B, Cin, Cout, S, S)}: "
      f"{t * 1e3:.3f} ms  {fl / t / 1e12:.0f} TF direct-equivalent, {fl * 16 / 36 / t / 1e12:.1f} TF executed = {fl * 16 / 36 / t / 157.3e12:.3f} of peak")
