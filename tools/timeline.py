#!/usr/bin/env python3
"""Print the kernel timeline (start, duration, queue) of the last few time steps of a rocprofv3 --kernel-trace run."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "?")))
rows.sort()
# four steps from the middle of the first (uninstrumented) timed loop of bench.py: counted by the bulk pass A launches
idx = [n for n, r in enumerate(rows) if r[2] in ("k_colfct", "k_colfct_dif", "k_colfct2", "k_colfct_sh", "k_colfct_sha", "k_colfct_sh_y", "k_colfct_sha_y")]
first = idx[12] if len(idx) >= 17 else idx[0]
last = idx[16] if len(idx) >= 17 else len(rows)
t0 = rows[first][0]
for s, e, name, q in rows[first:last]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  q{q}  {name}")
