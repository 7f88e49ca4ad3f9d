#!/usr/bin/env python3
"""Do consecutive render_kernel launches overlap?  Reads a rocprofv3 kernel_trace.csv:
   tools/trace_overlap.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "render_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
last = rows[-40:]
ov = 0
for a, b in zip(last, last[1:]):
    if int(b["Start_Timestamp"]) < int(a["End_Timestamp"]):
        ov += 1
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in last]
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6
print(f"last {len(last)} render kernels: {ov} of {len(last)-1} start before their predecessor ends; "
      f"duration avg {sum(dur)/len(dur):.3f} ms, {span/len(last):.3f} ms per kernel overall; queues {sorted(set(r['Queue_Id'] for r in last))}")
for r in last[-6:]:
    print("   q", r["Queue_Id"], "start", (int(r["Start_Timestamp"]) - t0) / 1e6, "end", (int(r["End_Timestamp"]) - t0) / 1e6, "grid", r.get("Grid_Size_X", r.get("Grid_Size", "?")))
