"""Kernel timeline (start, gap to the previous kernel, duration) of the last N ms of a rocprofv3 kernel trace csv, or from the last
launch of a marker kernel:  python benchmarks/timeline.py trace.csv [marker-kernel-substring] [min-duration-ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else None
mind = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
if marker:
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    rows = rows[idx[-1]:]
t0 = int(rows[0]["Start_Timestamp"]); prev = t0; tot = 0.0
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d, gap = (en - st) / 1e6, (st - prev) / 1e6
    tot += d
    if d >= mind or gap > 0.2:
        print(f"{(st - t0) / 1e6:8.3f} gap={gap:6.3f} {d:7.3f} {r['Kernel_Name'].split('(')[0].replace('bppp::', '')[:34]:34s} grid={r['Grid_Size_X']} wg={r['Workgroup_Size_X']}")
    prev = max(prev, en)
print(f"gpu {tot:.3f} ms, span {(prev - t0) / 1e6:.3f} ms")
