"""Idle gaps of the GPU inside one training step, from a rocprofv3 kernel trace (csv with Start/End timestamps).
Steps are delimited by `adam_kernel` launches (one per step: the FPN's optimizer)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
adam = [i for i, k in enumerate(ks) if k[2].startswith("adam_kernel")]
a, b = adam[-3], adam[-2]
step = ks[a + 1:b + 1]
t0, t1 = step[0][0], step[-1][1]
busy, cur_end = 0, t0
gaps = []
for i, (s, e, n) in enumerate(step):
    if s > cur_end:
        gaps.append((s - cur_end, step[i - 1][2] if i else "", n, (cur_end - t0) / 1e6))
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
print(f"step {(t1 - t0) / 1e6:.2f} ms, busy (union) {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms, kernels {len(step)}")
hist = {}
for g, *_ in gaps:
    k = "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"
    h = hist.setdefault(k, [0, 0]); h[0] += 1; h[1] += g
for k, (n, t) in hist.items():
    print(f"  gaps {k:8s}: {n:5d}  total {t / 1e6:.2f} ms")
for g, prev, nxt, at in sorted(gaps, key=lambda t: -t[0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    print(f"  {g / 1e3:8.1f} us at {at:6.2f} ms  after {prev[:50]:50s} before {nxt[:50]}")
