"""Per-queue kernel census of a rocprofv3 --kernel-trace CSV: launches and kernel time per hardware queue, top kernels by launch
count and by time (names shortened), per step.  usage: queue_census.py kernel_trace.csv STEPS [top]"""
import collections, csv, re, sys

path, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
short = lambda n: re.sub(r"\(.*", "", re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", n))[:70]
q = collections.defaultdict(lambda: [0, 0.0, collections.Counter(), collections.Counter()])
rows = list(csv.DictReader(open(path)))
# the steady-state part: drop the first third of the trace (construction, warm-up, captures)
t_lo = sorted(int(r["Start_Timestamp"]) for r in rows)[len(rows) // 3]
kept = [r for r in rows if int(r["Start_Timestamp"]) >= t_lo]
frac = len(kept) / max(1, len(rows))
for r in kept:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    e = q[r["Queue_Id"]]
    e[0] += 1
    e[1] += d
    e[2][short(r["Kernel_Name"])] += 1
    e[3][short(r["Kernel_Name"])] += d
print(f"# {len(kept)} of {len(rows)} dispatches kept (after the first third); per-step figures assume {steps:g} steps in the kept part")
for k, (n, ms, cnt, tim) in sorted(q.items(), key=lambda kv: -kv[1][1]):
    print(f"queue {k}: {n / steps:.0f} kernels/step, {ms / steps:.2f} ms of kernel time/step")
    print("  by launches: " + ", ".join(f"{a} {b / steps:.0f}" for a, b in cnt.most_common(top)))
    print("  by time (ms/step): " + ", ".join(f"{a} {b / steps:.3f}" for a, b in tim.most_common(top)))
