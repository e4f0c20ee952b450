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


def find_one(d, suffix):
    """the one file under gpurun_out/<d> whose name ends in `suffix` (rocprofv3 nests its output under host / pid directories)"""
    for dp, _, files in os.walk(os.path.join(G, d)):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(dp, f)
    return None


def reduce_on_box():
    """Runs on the GPU box right after the passes (benchmarks/profile_round.sh): every counter file -> gpurun_out/<dir>.reduced.csv
    (kernel, counter, launches, mean per launch) and every *_kernel_stats.csv copied up one level, so that only kilobytes travel back."""
    for d in sorted(os.listdir(G)):
        if not os.path.isdir(os.path.join(G, d)):
            continue
        cc = find_one(d, "counter_collection.csv")
        if cc:
            acc, cnt = defaultdict(float), defaultdict(int)
            with open(cc) as f:
                for r in csv.DictReader(f):
                    key = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
                    acc[key] += float(r["Counter_Value"]); cnt[key] += 1
            with open(os.path.join(G, d + ".reduced.csv"), "w") as f:
                f.write("kernel,counter,launches,mean_per_launch\n")
                for (k, c), v in sorted(acc.items(), key=lambda kv: -kv[1]):
                    f.write(f"{k},{c},{cnt[(k, c)]},{v / cnt[(k, c)]:.3f}\n")
        ks = find_one(d, "kernel_stats.csv")
        if ks:
            shutil.copy(ks, os.path.join(G, d + ".kernel_stats.csv"))


def reduced(d):
    """{kernel: {counter: (mean per launch, launches)}} from gpurun_out/<d>.reduced.csv"""
    out = defaultdict(dict)
    path = os.path.join(G, d + ".reduced.csv")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        next(f)
        for line in f:                                    # kernel names may hold commas (template arguments): split from the right
            k, c, n, v = line.rstrip("\n").rsplit(",", 3)
            out[k][c] = (float(v), int(n))
    return out


def round3(tag):
    """profiles/<tag>_*: kernel-time summaries (2^20 MSM alone, whole bench, the 4096-proof prove command), FETCH / WRITE per kernel for
    the MSM (traffic.json, read by bench.py) and for the prover, and the prover's SQ counters for k_comb_msm"""
    for d, dst in (("prof_msm", f"{tag}_msm_2p20_kernel_stats.csv"), ("prof_bench", f"{tag}_bench_default_kernel_stats.csv"),
                   ("prof_prove", f"{tag}_prove_4096_kernel_stats.csv")):
        s = os.path.join(G, d + ".kernel_stats.csv")
        if os.path.exists(s):
            shutil.copy(s, os.path.join(P, dst)); print("copied", dst)
    fe, wr = reduced("pmc_fetch"), reduced("pmc_write")
    if fe and wr:
        for kind, counter, data in (("fetch", "FETCH_SIZE", fe), ("write", "WRITE_SIZE", wr)):
            with open(os.path.join(P, f"{tag}_pmc_{kind}_size_per_kernel.csv"), "w") as f:
                f.write(f"kernel,launches,mean_{counter}_KB_per_launch\n")
                for k, cs in sorted(data.items(), key=lambda kv: -kv[1].get(counter, (0, 0))[0] * kv[1].get(counter, (0, 0))[1]):
                    if counter in cs:
                        f.write(f"{k},{cs[counter][1]},{cs[counter][0]:.3f}\n")
        name = "bppp::k_acc_points"
        fk, wk = fe[name]["FETCH_SIZE"][0], wr[name]["WRITE_SIZE"][0]
        per_kernel = {k: fe[k]["FETCH_SIZE"][0] * 1024 * 2 + wr[k]["WRITE_SIZE"][0] * 1024 for k in fe if "bppp::" in k and k in wr}
        traffic = {
            "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --headline-only, 2^20 pairs, auto window (c = 16); " + tag,
            "k_acc_points_FETCH_SIZE_KB_raw": fk, "k_acc_points_WRITE_SIZE_KB": wk,
            "correction": "gfx950 FETCH_SIZE counts 128-B requests at 64 B: x2 (MI355X_MICROARCH.md, HBM section); the accumulate kernel's point gathers are "
                          "64-B requests, for which the x2 may overstate",
            "k_acc_points_bytes_per_launch": fk * 1024 * 2 + wk * 1024, "bytes_per_launch_by_kernel": per_kernel,
        }
        pf, pw, ps = reduced("pmc_prove_fetch"), reduced("pmc_prove_write"), reduced("pmc_prove_sq")
        if pf and pw:
            prover = {}
            for k in pf:
                if "bppp::" in k and k in pw:
                    prover[k] = {"launches": pf[k]["FETCH_SIZE"][1], "fetch_bytes_per_launch_x2": pf[k]["FETCH_SIZE"][0] * 1024 * 2,
                                 "fetch_bytes_per_launch_raw": pf[k]["FETCH_SIZE"][0] * 1024, "write_bytes_per_launch": pw[k]["WRITE_SIZE"][0] * 1024}
                    if ps and k in ps:
                        prover[k]["sq_per_launch"] = {c: v[0] for c, v in ps[k].items()}
            traffic["prover_4096_proofs_64by64"] = {"source": "same counters, separate passes, BPPP_RP_NO_SPLIT=1 python3 benchmarks/prove_timing.py 4096 (one context)",
                                                    "by_kernel": prover}
            with open(os.path.join(P, f"{tag}_pmc_prover_per_kernel.csv"), "w") as f:
                f.write("kernel,launches,FETCH_SIZE_bytes_raw,FETCH_SIZE_bytes_x2,WRITE_SIZE_bytes," + ",".join(sorted(next(iter(ps.values())).keys()) if ps else []) + "\n")
                for k, v in sorted(prover.items(), key=lambda kv: -kv[1]["fetch_bytes_per_launch_raw"] * kv[1]["launches"]):
                    sq = v.get("sq_per_launch", {})
                    f.write(f"{k},{v['launches']},{v['fetch_bytes_per_launch_raw']:.0f},{v['fetch_bytes_per_launch_x2']:.0f},{v['write_bytes_per_launch']:.0f}," +
                            ",".join(f"{sq.get(c, 0):.0f}" for c in sorted(sq)) + "\n")
        json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)
        print("wrote traffic.json")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--reduce":
        return reduce_on_box()
    if len(sys.argv) > 1 and sys.argv[1] >= "r03":
        return round3(sys.argv[1])
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
