"""The static objects bench.py builds from the committed profiles (profiles/traffic.json, profiles/r04_*_kernel_stats.csv): they must exist and be
consistent with each other, or the bench line silently loses its roofline / stage tables (no GPU needed)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_prover_rows_roofline_comes_from_the_committed_profiles():
    import bench
    r = bench.prover_rows_roofline()
    assert r is not None and r["kernel"] == "k_comb_msm_rows" and r["bound"] == "hbm" and r["unit"] == "GB/s"
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the counters and the duration are of the same command: VALU-busy is a fraction, and the raw fetch count is within 25 % of the algorithmic bytes
    assert 0.3 < r["valu_busy_est"] < 1.0 and 0 < r["wait_inst_frac"] < 1
    alg = r["achieved"] * 1e9 * r["ms_per_launch"] * 1e-3
    assert 0.75 < r["fetch_bytes_raw"] / alg < 1.25
    assert 1500 < r["valu_insts_per_addition"] < 3000


def test_verifier_stage_tables_exist():
    import bench
    for key in ("verify_4096_64by64", "verify_binary_1024_64x64bit"):
        tab = bench.profile_table(key)
        rows = bench.stage_rows(tab)
        assert rows and all(x["ms"] > 0 and x["hbm_bytes"] >= 0 for x in rows)
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert t["k_acc_points_bytes_per_launch"] > 2**20 * 96          # more than the algorithmic 96 B per pair: every window re-gathers the points
