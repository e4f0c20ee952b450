"""GPU busy time (union of kernel intervals) over the last prove batch of a rocprofv3 kernel trace; kernels by summed duration.
   python benchmarks/busy.py trace.csv [marker]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_rpp_draws"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
# the last batch may launch the marker twice (two halves): start from the second-to-last marker if the two are within 5 ms
start = idx[-1]
if len(idx) >= 2 and int(rows[idx[-1]]["Start_Timestamp"]) - int(rows[idx[-2]]["Start_Timestamp"]) < 5e6: start = idx[-2]
sel = rows[start:]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e for _, e in iv) - iv[0][0]
acc = collections.defaultdict(float)
for r in sel: acc[r["Kernel_Name"].split("(")[0].replace("bppp::", "")] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print(f"span {span / 1e6:.1f} ms, busy (union) {busy / 1e6:.1f} ms, sum of kernel durations {sum(acc.values()):.1f} ms")
print(", ".join(f"{k} {v:.1f}" for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:10]))
