"""Summarise the PMC passes of tools/collect_profile.sh: per-kernel mean counters, the gfx950 FETCH/WRITE_SIZE
calibration from the known-byte kernels, corrected HBM traffic per launch of the conv kernels, MFMA-busy fraction.
Writes <dir>/traffic.json (read by bench.py to fill roofline.traffic)."""
import csv, json, os, sys
from collections import defaultdict

d = sys.argv[1]


def load(path):
    acc = defaultdict(lambda: defaultdict(list))
    if not os.path.exists(path):
        return acc
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def mean(v):
    return sum(v) / len(v) if v else float("nan")


fetch, write, sq = load(f"{d}/pmc_FETCH_SIZE.csv"), load(f"{d}/pmc_WRITE_SIZE.csv"), load(f"{d}/pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv")
CAL_BYTES = 32 * 256 * 128 * 128 * 4


def find(acc, key):
    for k in acc:
        if k.startswith(key):
            return k
    return None


def counter(acc, kname, cname):
    k = find(acc, kname)
    return mean(acc[k][cname]) if k else float("nan")


# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like units of the derived metric; establish bytes-per-unit from
# the calibration kernels instead of trusting the nominal unit.
out = {"calibration": {}, "kernels": {}}
for width, kname in (("4B_per_lane", "act_fwd_kernel"), ("16B_per_lane", "bn_apply_kernel")):
    f, w = counter(fetch, kname, "FETCH_SIZE"), counter(write, kname, "WRITE_SIZE")
    out["calibration"][width] = {"kernel": kname, "known_read_bytes": CAL_BYTES, "known_write_bytes": CAL_BYTES,
                                 "FETCH_SIZE": f, "WRITE_SIZE": w,
                                 "read_bytes_per_unit": CAL_BYTES / f if f == f and f > 0 else None,
                                 "write_bytes_per_unit": CAL_BYTES / w if w == w and w > 0 else None}
c4 = out["calibration"]["4B_per_lane"]
print("calibration:", json.dumps(out["calibration"], indent=1))
ALG = {"conv_gemm_kernel<TileCfg<2, 2, 2, 2, 18>, 3, 3, false, false, true>": (32 * 256 * 64 * 64 * 4 + 256 * 256 * 9 * 4, 32 * 256 * 64 * 64 * 4),
       "conv_gemm_kernel<TileCfg<2, 2, 2, 2, 18>, 3, 3, true, false, true>": (32 * 256 * 64 * 64 * 4 + 256 * 256 * 9 * 4, 32 * 256 * 64 * 64 * 4),
       "conv_wgrad_kernel<TileCfg<2, 2, 2, 2, 32>, 3, 3>": (2 * 32 * 256 * 64 * 64 * 4, 256 * 256 * 9 * 4),
       "conv_wgrad3x3_kernel<TileCfg<2, 2, 2, 2, 32>, 32, false>": (2 * 32 * 256 * 64 * 64 * 4, 256 * 256 * 9 * 4),
       # Winograd kernels on the same layer: forward / data gradient read the input and the 16 C M transformed filters, write the
       # output; the weight gradient reads x and dy and writes its 16 K-split slabs of 9 C M floats
       "wino3x3_kernel<16, false>": (32 * 256 * 64 * 64 * 4 + 16 * 256 * 256 * 4, 32 * 256 * 64 * 64 * 4),
       "wino3x3_wgrad_kernel": (2 * 32 * 256 * 64 * 64 * 4, 16 * 256 * 256 * 9 * 4)}
for k in sorted(set(list(fetch) + list(write) + list(sq))):
    if not ("conv_gemm" in k or "conv_wgrad" in k or "slab_reduce" in k or "wino3x3" in k or "wnw_reduce" in k):
        continue
    f, w = mean(fetch[k]["FETCH_SIZE"]) if k in fetch else float("nan"), mean(write[k]["WRITE_SIZE"]) if k in write else float("nan")
    rec = {"FETCH_SIZE": f, "WRITE_SIZE": w}
    if c4["read_bytes_per_unit"] and f == f:
        rec["read_bytes"] = f * c4["read_bytes_per_unit"]
    if c4["write_bytes_per_unit"] and w == w:
        rec["write_bytes"] = w * c4["write_bytes_per_unit"]
    if "read_bytes" in rec and "write_bytes" in rec:
        rec["hbm_bytes_per_launch"] = rec["read_bytes"] + rec["write_bytes"]
    name = k.split("(")[0].replace("void ", "")
    if name in ALG:
        rec["algorithmic_bytes"] = sum(ALG[name])
        if "hbm_bytes_per_launch" in rec:
            rec["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes"]
    if k in sq:
        mf, bz, gui = mean(sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"]), mean(sq[k]["SQ_BUSY_CU_CYCLES"]), mean(sq[k]["GRBM_GUI_ACTIVE"])
        rec.update({"SQ_VALU_MFMA_BUSY_CYCLES": mf, "SQ_BUSY_CU_CYCLES": bz, "GRBM_GUI_ACTIVE": gui})
        if gui == gui and gui > 0:
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over 256 CUs x 4 SIMDs (= 64 cycles per 32x32x2 fp32 MFMA);
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs, so gui / 8 = wall cycles of the dispatch.
            rec["mfma_busy_frac"] = mf / (gui / 8.0 * 256 * 4)
    out["kernels"][name] = rec
    print(name, json.dumps(rec))
# the same counters over the bench's own launches: average HBM bytes per launch of each conv kernel instantiation
bf = json.load(open(f"{d}/pmc_bench_FETCH_SIZE.json")) if os.path.exists(f"{d}/pmc_bench_FETCH_SIZE.json") else {}
bw = json.load(open(f"{d}/pmc_bench_WRITE_SIZE.json")) if os.path.exists(f"{d}/pmc_bench_WRITE_SIZE.json") else {}
bs = json.load(open(f"{d}/pmc_bench_SQ_VALU_MFMA_BUSY_CYCLES.json")) if os.path.exists(f"{d}/pmc_bench_SQ_VALU_MFMA_BUSY_CYCLES.json") else {}
out["bench_launch_average"] = {}
for k in sorted(bf):
    if k not in bw or not c4["read_bytes_per_unit"]:
        continue
    rb, wb = bf[k]["FETCH_SIZE"] * c4["read_bytes_per_unit"], bw[k]["WRITE_SIZE"] * c4["write_bytes_per_unit"]
    rec = {"read_bytes": rb, "write_bytes": wb, "hbm_bytes_per_launch": rb + wb, "launches_sampled": bf[k]["n"]}
    if k in bs and bs[k].get("GRBM_GUI_ACTIVE", 0) > 0:
        rec["mfma_busy_frac"] = bs[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (bs[k]["GRBM_GUI_ACTIVE"] / 8.0 * 1024)
    out["bench_launch_average"][k] = rec
print("bench launch averages (top by bytes):")
for k, v in sorted(out["bench_launch_average"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])[:14]:
    print(f"  {k[:78]:78s} {v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch x{v['launches_sampled']:4d}  mfma_busy {v.get('mfma_busy_frac', float('nan')):.3f}")
# what the counters are valid for: the kernel sources of this snapshot (bench.py compares and prints `stale`)
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
try:
    from bench import csrc_sha16
    out["collected_on"] = {"csrc_sha16": csrc_sha16()}
except Exception as exc:      # noqa: BLE001
    out["collected_on"] = {"csrc_sha16": None, "error": str(exc)[:100]}
json.dump(out, open(f"{d}/traffic.json", "w"), indent=1)
