#!/usr/bin/env python3
"""Merged timeline (kernels and host<->device copies) of the last steps of tools/overlay_time.py under
rocprofv3 --kernel-trace --memory-copy-trace: where a resident overlay step spends its time.
usage: python tools/overlay_tl.py DIR [nsteps]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Direction", "copy")
        size = r.get("Size") or r.get("Bytes") or ""
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"COPY {name} {size}", "dma"))
rows.sort()
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 4
idx = [n for n, r in enumerate(rows) if r[2] == "k_colfct"]
first = idx[-nstep - 1]
last = idx[-1]
t0 = rows[first][0]
print(f"{nstep} steps in {(rows[last][0] - t0) / 1e3 / nstep:.1f} us per step (bulk pass A to bulk pass A)")
for s, e, name, q in rows[first:last]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {q:>4}  {name}")
