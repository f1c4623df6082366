"""GPU idle time of a steady-state run from a rocprofv3 --kernel-trace CSV: union of the kernel intervals over all queues against the
wall span of the kept part (the last two thirds of the dispatches), and the distribution of the gaps in that union.
usage: gpu_idle.py kernel_trace.csv [STEPS_IN_KEPT_PART]"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
iv = iv[len(iv) // 3:]
span = (max(e for _s, e, _n in iv) - iv[0][0]) * 1e-6
busy, gaps, cur_s, cur_e, last = 0.0, [], iv[0][0], iv[0][1], iv[0][2]
for s, e, n in iv[1:]:
    if s > cur_e:
        busy += (cur_e - cur_s) * 1e-6
        gaps.append(((s - cur_e) * 1e-3, last, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last = n
busy += (cur_e - cur_s) * 1e-6
ksum = sum(e - s for s, e, _n in iv) * 1e-6
print(f"kept {len(iv)} dispatches over {span:.2f} ms: union busy {busy:.2f} ms ({busy / span:.3f}), idle {span - busy:.2f} ms, kernel-time sum {ksum:.2f} ms (overlap x{ksum / busy:.2f})")
if len(sys.argv) > 2:
    st = float(sys.argv[2])
    print(f"per step ({st:g} steps): span {span / st:.2f} ms, busy {busy / st:.2f}, idle {(span - busy) / st:.2f}, gaps {len(gaps) / st:.0f}")
for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1e9)):
    g = [x for x in gaps if lo <= x[0] < hi]
    print(f"  gaps {lo:>4}-{hi:<6g} us: {len(g):6d}, total {sum(x[0] for x in g) * 1e-3:8.2f} ms")
short = lambda n: n.replace("void ", "")[:60]
for g in sorted(gaps, key=lambda x: -x[0])[:12]:
    print(f"  {g[0]:8.1f} us after {short(g[1])}  before {short(g[2])}")
