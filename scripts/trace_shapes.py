"""Per-(kernel, grid) duration table from a rocprofv3 kernel trace: python scripts/trace_shapes.py <kernel_trace.csv>"""
import collections
import csv
import sys

agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Kernel_Name"][:48], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    v2 = sorted(v)
    print(f"{k[0]:50s} grid {k[1]:>8s} {k[2]:>3s} {k[3]:>5s}  n={len(v):4d}  med {v2[len(v2)//2]:8.1f} us  tot {sum(v)/1e3:7.1f} ms")
