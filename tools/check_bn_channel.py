"""ge_bn_fwd_channel / ge_bn_bwd_channel against the three-launch path and an fp64 reference (accuracy of the moments)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphecho_amd import functional as GF
from graphecho_amd._lib import lib, check
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, C, H) in ((3, 64, 64), (32, 256, 16), (8, 512, 8)):
    HW = H * H
    x = (torch.randn(B, C, H, H, device=dev) * 2.5 + torch.randn(1, C, 1, 1, device=dev) * 3).contiguous()
    w = torch.randn(C, C, 1, 1, device=dev) * 0.1
    ref_mean = x.double().mean((0, 2, 3)); ref_var = x.double().var((0, 2, 3), unbiased=False)
    st = GF._stream()
    res = {}
    for name in ("channel", "three"):
        mean = torch.empty(C, device=dev); invstd = torch.empty(C, device=dev); y = torch.empty_like(x)
        rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
        if name == "channel":
            check(lib.ge_bn_fwd_channel(x.data_ptr(), None, 0, 3, 0, None, None, None, y.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                        rm.data_ptr(), rv.data_ptr(), B, C, HW, 1e-5, 0.1, 0, st))
        else:
            nb = lib.ge_bn_num_partials(B, HW); own = torch.empty(C * nb * 3, device=dev)
            check(lib.ge_bn_stats_partial(x.data_ptr(), own.data_ptr(), B, C, HW, st))
            check(lib.ge_bn_finalize(own.data_ptr(), nb * 3, 3, nb, C, 1e-5, 0.1, None, mean.data_ptr(), invstd.data_ptr(), rm.data_ptr(), rv.data_ptr(), st))
            check(lib.ge_bn_apply(x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), None, None, None, y.data_ptr(), B, C, HW, 0, st))
        torch.cuda.synchronize()
        em = (mean.double() - ref_mean).abs().max().item()
        ev = (1 / invstd.double() ** 2 - 1e-5 - ref_var).abs().max().item() / ref_var.max().item()
        res[name] = (em, ev, y)
        print(f"B{B} C{C} H{H} {name:8s} mean err {em:.2e} (scale {ref_mean.abs().max().item():.2f})  var rel err {ev:.2e}")
    print("   y max diff", (res["channel"][2] - res["three"][2]).abs().max().item())
