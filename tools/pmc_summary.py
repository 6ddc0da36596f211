#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/profile_round.sh: mean per launch of every counter per kernel,
and HBM traffic per launch corrected as MI355X_MICROARCH.md (HBM / rocprofv3 PMC slots) prescribes:
FETCH_SIZE and WRITE_SIZE are in KB; their relation to real bytes depends on the access width, so both are
calibrated on tools/calib_traffic.hip (8 B per lane coalesced reads/writes of a known size, the width the
transport kernels use).    usage: tools/pmc_summary.py <pmc dir> [traffic.json]"""
import collections
import csv
import glob
import json
import sys

base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof/pmc"


def load(d):
    fs = glob.glob(f"{base}/{d}/*/*_counter_collection.csv")
    # a kernel launched with several grids in one step (pass A and B run once for T,S on the side stream and once for
    # the other tracers on the main stream) is listed per grid: the plain name is the largest grid, the others "name#<grid>"
    rows = []
    for f in fs:
        for r in csv.DictReader(open(f)):
            rows.append((r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
    biggest = {}
    for name, grid, _, _ in rows:
        biggest[name] = max(biggest.get(name, 0), grid)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, grid, cname, val in rows:
        agg[name if grid == biggest[name] else f"{name}#{grid}"][cname].append(val)
    return agg


def mean(v):
    return sum(v) / len(v)


print("rocprofv3 --pmc passes (tools/profile_round.sh), bench.py c30 102x102x19 nt=30; mean per launch")
for d in ["sq1", "sq2", "fetch", "write", "tcc"]:
    agg = load(d)
    for k in sorted(agg):
        if k.startswith("k_"):
            print(f"{d:6s} {k:20s} " + " ".join(f"{c}={mean(v):.4g}" for c, v in sorted(agg[k].items())))

# calibration: known bytes / counter (KB)
CAL_BYTES = 64 * (1 << 20)
cf, cw = load("calib_fetch"), load("calib_write")
cal = {}
try:
    cal["fetch_read8"] = CAL_BYTES / (mean(cf["k_calib_read8"]["FETCH_SIZE"]) * 1024.0)
    cal["fetch_copy8"] = CAL_BYTES / (mean(cf["k_calib_copy8"]["FETCH_SIZE"]) * 1024.0)
    cal["write_copy8"] = CAL_BYTES / (mean(cw["k_calib_copy8"]["WRITE_SIZE"]) * 1024.0)
except (KeyError, ZeroDivisionError) as e:
    print("calibration missing:", e)
print("calibration (true bytes / counter bytes, 8 B per lane coalesced, 64 MiB): " + " ".join(f"{k}={v:.3f}" for k, v in cal.items()))
fs, ws = load("fetch"), load("write")
SQ = load("sq1")
BIGGEST = {}
for f in glob.glob(f"{base}/sq1/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        BIGGEST[n] = max(BIGGEST.get(n, 0), int(r["Grid_Size"]))
traffic = {"unit": "bytes per launch", "calibration": cal, "kernels": {}}
kf = cal.get("fetch_read8", 1.0)
kw = cal.get("write_copy8", 1.0)
for k in sorted(set(fs) | set(ws)):
    if not k.startswith("k_"):
        continue
    fr = mean(fs[k]["FETCH_SIZE"]) * 1024.0 if k in fs and "FETCH_SIZE" in fs[k] else None
    wr = mean(ws[k]["WRITE_SIZE"]) * 1024.0 if k in ws and "WRITE_SIZE" in ws[k] else None
    ent = {"fetch_raw": fr, "write_raw": wr,
           "fetch": None if fr is None else fr * kf, "write": None if wr is None else wr * kw}
    ent["total"] = None if fr is None or wr is None else ent["fetch"] + ent["write"]
    # wave-level VALU instructions per launch: the SQ counters see a share of the launch's waves (SQ_WAVES of grid/64)
    sq = SQ.get(k)
    if sq and "SQ_INSTS_VALU" in sq and "SQ_WAVES" in sq and mean(sq["SQ_WAVES"]) > 0:
        grid = int(k.split("#")[1]) if "#" in k else BIGGEST.get(k.split("#")[0], 0)
        if grid:
            ent["valu_wave_insts"] = mean(sq["SQ_INSTS_VALU"]) * (grid / 64.0) / mean(sq["SQ_WAVES"])
    traffic["kernels"][k] = ent
    print(f"traffic {k:20s} fetch={ent['fetch'] and ent['fetch']/1e6:.2f} MB write={ent['write'] and ent['write']/1e6:.2f} MB per launch (corrected)")
if len(sys.argv) > 2:
    json.dump(traffic, open(sys.argv[2], "w"), indent=1)
