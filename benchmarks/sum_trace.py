"""Sum of kernel durations of the LAST prove batch in a rocprofv3 kernel trace (csv), by kernel."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_rpp_draws" in r["Kernel_Name"]]
sel = rows[idx[-1]:] if idx else rows
acc = collections.defaultdict(float)
for r in sel:
    acc[r["Kernel_Name"].split("(")[0].replace("bppp::", "")] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(acc.values())
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
top = sorted(acc.items(), key=lambda kv: -kv[1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]
print(f"gpu {tot:.1f} ms, span {span:.1f} ms; " + ", ".join(f"{k} {v:.1f}" for k, v in top))
