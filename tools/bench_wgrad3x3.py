"""3x3 / s1 / p1 weight gradient: patch-staged kernel vs the per-tap kernel (GE_WGRAD_PATCH=0, separate process) on the
shapes of config 2, GPU time per launch (wgrad kernel and slab reduce timed apart) and bit-equality of the results."""
import json, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(32, 256, 64, 64, 256), (32, 256, 64, 64, 128), (32, 64, 64, 64, 64), (32, 128, 32, 32, 128),
          (32, 256, 32, 32, 256), (32, 256, 16, 16, 256), (32, 512, 8, 8, 512), (32, 256, 8, 8, 256)]


def run():
    from graphecho_amd import functional as GF
    from graphecho_amd._lib import lib, check
    dev = torch.device("cuda:0")
    out = {}
    for B, Ci, H, W, Co in SHAPES:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, Ci, H, W, generator=g).to(dev)
        dy = torch.randn(B, Co, H, W, generator=g).to(dev)
        dw = torch.empty(Co, Ci, 3, 3, device=dev)
        ws = torch.empty(lib.ge_conv2d_wgrad_workspace(B, Ci, Co, H, W, 3, 3, 1), device=dev)
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: check(lib.ge_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, Ci, H, W, Co,
                                                 H, W, 3, 3, 1, 1, 1, 0, st))
        for _ in range(3):
            call()
        mid = torch.cuda.Event(enable_timing=True); mid.record()
        lib.ge_set_wgrad_split_event(mid.cuda_event)
        tk = tr = 0.0
        n = 20
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); call(); b.record(); torch.cuda.synchronize()
            tk += a.elapsed_time(mid); tr += mid.elapsed_time(b)
        lib.ge_set_wgrad_split_event(None)
        fl = 2.0 * B * H * W * Co * Ci * 9
        out[f"{Ci}->{Co}@{H}x{W}"] = {"kernel": lib.ge_last_conv_kernel().decode(), "us": round(1e3 * tk / n, 1),
                                      "tflops": round(fl / (tk / n * 1e-3) / 1e12, 1), "reduce_us": round(1e3 * tr / n, 1),
                                      "sum": dw.double().sum().item(), "abs": dw.double().abs().sum().item()}
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run()
    else:
        res = {}
        for flag, db in (("1", "1"), ("1", "0"), ("0", "0")):
            r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, GE_WGRAD_PATCH=flag, GE_WGRAD_DB=db),
                               capture_output=True, text=True)
            res[flag + db] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        for k in res["11"]:
            a, a1, b = res["11"][k], res["10"][k], res["00"][k]
            print(f"{k:18s} patch 2-stage {a['us']:7.1f} us {a['tflops']:6.1f} TF | patch 1-stage {a1['us']:7.1f} us {a1['tflops']:6.1f} TF | "
                  f"per-tap {b['us']:7.1f} us {b['tflops']:6.1f} TF | reduce {a['reduce_us']:5.1f} us | "
                  f"bit-equal {a['sum'] == b['sum'] and a['abs'] == b['abs'] and a1['sum'] == b['sum']}")
