"""Turn the rocprofv3 CSVs under gpurun_out/ into the committed summaries under profiles/.

  python benchmarks/summarise_profiles.py <round-tag>

Inputs (written on the GPU box, see DESIGN.md section 4 for the exact commands):
  gpurun_out/prof_msm/msm_kernel_stats.csv        rocprofv3 --kernel-trace --stats, bench.py --headline-only --no-cpu-baseline (the 2^20 MSM alone)
  gpurun_out/prof_bench/bench_kernel_stats.csv    same, whole default bench.py (MSM + verify + prove legs)
  gpurun_out/pmc_fetch/fetch_counter_collection.csv   rocprofv3 --pmc FETCH_SIZE   (own pass)
  gpurun_out/pmc_write/write_counter_collection.csv   rocprofv3 --pmc WRITE_SIZE   (own pass)
Outputs: profiles/<tag>_*.csv (copies / per-kernel reductions) and profiles/traffic.json (read by bench.py).
HBM bytes follow MI355X_MICROARCH.md's HBM section: FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts each
128-B request as 64 B, so fetch bytes = FETCH_SIZE * 1024 * 2.
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def per_kernel_counter(path, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            acc[name] += float(r["Counter_Value"])
            cnt[name] += 1
    return {k: (acc[k] / cnt[k], cnt[k]) for k in acc}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(P, exist_ok=True)
    for src, dst in (("prof_msm/msm_kernel_stats.csv", f"{tag}_msm_2p20_kernel_stats.csv"),
                     ("prof_bench/bench_kernel_stats.csv", f"{tag}_bench_default_kernel_stats.csv")):
        s = os.path.join(G, src)
        if os.path.exists(s):
            shutil.copy(s, os.path.join(P, dst))
            print("copied", dst)
    out = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        s = os.path.join(G, f"pmc_{kind}", f"{kind}_counter_collection.csv")
        if not os.path.exists(s):
            continue
        pk = per_kernel_counter(s, counter)
        with open(os.path.join(P, f"{tag}_pmc_{kind}_size_per_kernel.csv"), "w") as f:
            f.write(f"kernel,launches,mean_{counter}_KB_per_launch\n")
            for k, (v, n) in sorted(pk.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
                f.write(f"{k},{n},{v:.3f}\n")
        out[counter] = pk
    if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
        name = "bppp::k_acc_points"
        fk, wk = out["FETCH_SIZE"][name][0], out["WRITE_SIZE"][name][0]
        per_kernel = {}
        for k in out["FETCH_SIZE"]:
            if k.startswith("bppp::") and k in out["WRITE_SIZE"]:
                per_kernel[k] = out["FETCH_SIZE"][k][0] * 1024 * 2 + out["WRITE_SIZE"][k][0] * 1024
        traffic = {
            "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --headline-only, 2^20 pairs, auto window (c = 16)",
            "k_acc_points_FETCH_SIZE_KB_raw": fk,
            "k_acc_points_WRITE_SIZE_KB": wk,
            "correction": "gfx950 FETCH_SIZE counts 128-B requests at 64 B: x2 (MI355X_MICROARCH.md, HBM section); the accumulate kernel's point gathers are "
                          "64-B requests, for which the x2 may overstate (raw: 17.8 M entries x ~82 B)",
            "k_acc_points_bytes_per_launch": fk * 1024 * 2 + wk * 1024,
            "bytes_per_launch_by_kernel": per_kernel,
        }
        json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)
        print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
