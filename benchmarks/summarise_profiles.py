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


def kernel_stats(d):
    """{kernel: (calls, mean ns)} from gpurun_out/<d>.kernel_stats.csv"""
    path = os.path.join(G, d + ".kernel_stats.csv")
    if not os.path.exists(path):
        return None
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            out[r["Name"].split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]))
    return out


def verifier_table(tag, name, stats_dir, fetch_dir, write_dir, sq_dir, calls, what):
    """Per-kernel table of ONE verifier call: launches, ms, HBM bytes (FETCH x2 per the guide, raw beside it; WRITE), wait fraction and the
    VALU-busy estimate.  `calls` = verifier calls in the profiled command.  Returns the dict stored in traffic.json."""
    ks, fe, wr, sq = kernel_stats(stats_dir), reduced(fetch_dir), reduced(write_dir), reduced(sq_dir)
    if not (ks and fe and wr):
        return None
    rows = []
    for k, (n, mean_ns) in ks.items():
        if "bppp::" not in k or k not in fe or k not in wr:
            continue
        per_call = n / calls
        f_raw = fe[k]["FETCH_SIZE"][0] * 1024 * per_call
        w = wr[k]["WRITE_SIZE"][0] * 1024 * per_call
        row = {"kernel": k, "launches_per_call": per_call, "ms_per_call": mean_ns * per_call / 1e6, "fetch_bytes_raw": f_raw, "fetch_bytes_x2": 2 * f_raw, "write_bytes": w}
        if sq and k in sq:
            c = {cn: v[0] for cn, v in sq[k].items()}
            if c.get("SQ_WAVE_CYCLES"):
                row["wait_inst_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
            # SQ_ACTIVE_INST_VALU counts quad-cycles of VALU execution summed over the chip's 1024 SIMDs (it equals SQ_INSTS_VALU for plain
            # 4-cycle instructions); busy = 4 x that / (1024 SIMDs x kernel cycles at 2.4 GHz)
            row["valu_busy_est"] = 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024.0 * mean_ns * 2.4)
            row["valu_insts"] = c.get("SQ_INSTS_VALU", 0.0) * per_call
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_call"])
    with open(os.path.join(P, f"{tag}_pmc_{name}_per_kernel.csv"), "w") as f:
        f.write(f"# {what}; one verifier call; FETCH_SIZE / WRITE_SIZE / SQ_* in separate rocprofv3 --pmc passes, kernel times from a --kernel-trace --stats pass of the same command\n")
        f.write("kernel,launches_per_call,ms_per_call,fetch_bytes_raw,fetch_bytes_x2,write_bytes,wait_inst_frac,valu_busy_est,valu_insts\n")
        for r in rows:
            f.write("%s,%.2f,%.4f,%.0f,%.0f,%.0f,%s,%s,%.0f\n" % (r["kernel"], r["launches_per_call"], r["ms_per_call"], r["fetch_bytes_raw"], r["fetch_bytes_x2"], r["write_bytes"],
                                                               "%.3f" % r["wait_inst_frac"] if "wait_inst_frac" in r else "", "%.3f" % r["valu_busy_est"] if "valu_busy_est" in r else "",
                                                               r.get("valu_insts", 0.0)))
    tot = {"ms_of_kernels": sum(r["ms_per_call"] for r in rows), "fetch_bytes_x2": sum(r["fetch_bytes_x2"] for r in rows), "fetch_bytes_raw": sum(r["fetch_bytes_raw"] for r in rows),
           "write_bytes": sum(r["write_bytes"] for r in rows)}
    tot["bytes_per_call"] = tot["fetch_bytes_x2"] + tot["write_bytes"]
    print("wrote", f"{tag}_pmc_{name}_per_kernel.csv", tot)
    return {"source": what + "; " + tag, "per_call": tot, "by_kernel": {r["kernel"]: {k: v for k, v in r.items() if k != "kernel"} for r in rows}}


def round4(tag):
    """round 3's summaries plus the verifier's: per-kernel counters of one 4096-proof 64by64 verify call and one 1024-proof binary 64 x 64-bit one,
    kernel-time summaries of the new provers"""
    round3(tag)
    for d, dst in (("prof_verify", f"{tag}_verify_4096_kernel_stats.csv"), ("prof_binv", f"{tag}_binary_verify_1024_kernel_stats.csv"),
                   ("prof_binp", f"{tag}_binary_prove_verify_1024_kernel_stats.csv"), ("prof_ipp", f"{tag}_ip_prove_verify_16384_kernel_stats.csv")):
        s_ = os.path.join(G, d + ".kernel_stats.csv")
        if os.path.exists(s_):
            shutil.copy(s_, os.path.join(P, dst)); print("copied", dst)
    tpath = os.path.join(P, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    v = verifier_table(tag, "verify", "prof_verify", "pmc_verify_fetch", "pmc_verify_write", "pmc_verify_sq", 6,
                       "VERIFY_REPS=6 python3 benchmarks/verify_timing.py 4096 (4096 distinct 64by64 proofs read from a file, bppp_rp_verify_batch_device)")
    if v:
        traffic["verify_4096_64by64"] = v
    b = verifier_table(tag, "binary_verify", "prof_binv", "pmc_binv_fetch", "pmc_binv_write", "pmc_binv_sq", 4,
                       "python3 benchmarks/binary_64by64.py 1024 (1024 distinct 64 x 64-bit binary proofs read from a file, 4 calls of bppp_rp_verify_batch_device)")
    if b:
        traffic["verify_binary_1024_64x64bit"] = b
    json.dump(traffic, open(tpath, "w"), indent=1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--reduce":
        return reduce_on_box()
    if len(sys.argv) > 1 and sys.argv[1] >= "r04":
        return round4(sys.argv[1])
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
