"""Union of kernel-busy intervals in the LAST call of a rocprofv3 kernel trace (from the last launch of a marker kernel): how much of the span has
some kernel running, how much has one of the comb kernels running, and the kernel-time sums.   python benchmarks/busy_union.py trace.csv [marker]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_rpp_draws"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
# the last call of a split batch starts with TWO marker launches (one per half): go back to the first of the last group
i0 = idx[-1]
while len(idx) > 1 and idx[-1] - idx[-2] < 50 and int(rows[idx[-1]]["Start_Timestamp"]) - int(rows[idx[-2]]["Start_Timestamp"]) < 5e6:
    idx.pop(); i0 = idx[-1]
rows = rows[i0:]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
allv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
comb = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_comb_msm" in r["Kernel_Name"]]
span = max(e for _, e in allv) - min(s for s, _ in allv)
print(f"span {span / 1e6:.2f} ms; some kernel running {union(allv) / 1e6:.2f} ms; a comb MSM kernel running {union(comb) / 1e6:.2f} ms; "
      f"sum of kernel times {sum(e - s for s, e in allv) / 1e6:.2f} ms, of comb MSM kernels {sum(e - s for s, e in comb) / 1e6:.2f} ms; {len(rows)} launches")
