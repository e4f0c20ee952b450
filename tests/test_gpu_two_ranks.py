"""Multi-GPU readiness on ONE card: bench.py's N > 1 path with two ranks on cuda:0 (gloo for the 64-byte exchange), every kernel the
real HIP one.  `python bench.py --gpus 2` is started PLAINLY, as the driver starts it: the parent launches its two ranks itself before
anything touches the GPU (bench.py launch_ranks).  The MSM legs shard the terms, the verify legs shard the proofs; in both the ranks
all-gather their partial points and add them (bulletproofspp_amd/dist.py).  Checked: the combined MSM point equals the single-rank
MSM over all terms (--check-combined); the verify leg reports BOTH scaling modes — strong (BASELINE configs 4 / 5: one job of proofs
split over the ranks, bppp_rp_verify_shard_device with the job-wide seed and per-rank offsets; the summed rank points are the
identity, asserted inside bench.py) and weak (a full batch per rank) — and the JSON line reports n_gpus = 2.  The torch.distributed.run
route is kept working too (second test, headline leg only)."""
import json
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(p):
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_two_ranks_plain_start_both_scaling_modes():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--log2n", "17", "--verify-batch", "256", "--backend", "gloo",
           "--same-device", "--check-combined", "--no-cpu-baseline", "--msm-streams", "1"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = _json_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT))
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["combined_check"].startswith("sum of 2 rank-local MSMs == single-rank MSM")
    ms = out["msm_strong_scaling"]
    assert ms["scaling"] == "strong" and ms["pairs_per_gpu"] == (1 << 17) // 2 and ms["value"] > 0
    v = out["verify"]
    assert v["scaling"] == "strong" and v["job_proofs"] == 256 and v["batch_per_gpu"] == 128 and v["value"] > 0        # config 5's layout
    assert v["weak"]["scaling"] == "weak" and v["weak"]["batch_per_gpu"] == 256 and v["weak"]["job_proofs"] == 512 and v["weak"]["value"] > 0
    v4 = v["other_shapes"][0]                                                                                          # config 4's layout
    assert v4["scaling"] == "strong" and v4["job_proofs"] == 128 and v4["batch_per_gpu"] == 64 and v4["weak"]["batch_per_gpu"] == 128
    assert out["prove"]["replicas"] == 2
    assert "rank 1 of 2" in v["tamper_check"] and "rank 1 of 2" in v4["tamper_check"]


def _plain_start(n, extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--log2n", "14", "--verify-batch", str(16 * n), "--ip-batch", "0",
           "--binary-batch", "0", "--backend", "gloo", "--same-device", "--check-combined", "--no-cpu-baseline", "--msm-streams", "1"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return _json_line(subprocess.run(cmd, capture_output=True, text=True, timeout=1100, env=env, cwd=ROOT))


@pytest.mark.parametrize("n,tamper", [(4, 2), (5, 4)])
def test_bench_four_and_five_ranks_on_one_card(n, tamper):
    """BASELINE config 4 (4 GPUs) and the widest launch a one-GPU box admits beside the test runner itself (at most 6 processes may hold the
    card, and the pytest process is one of them; the 8-rank layout of
    config 5 is rehearsed shard by shard in test_gpu_native_verify.py::test_eight_shards_of_one_job and rank by rank on the CPU in
    tests/test_dist_gloo.py): `python bench.py --gpus N` started plainly — N children spawned before any GPU call — every rank on cuda:0,
    the exchange over gloo, small batches (no rank builds a prover table).  The job shards, the weak and strong figures and the rejection
    of a job with ONE corrupted proof on a middle / the last rank are asserted."""
    out = _plain_start(n, ["--tamper-rank", str(tamper)])
    assert out["n_gpus"] == n and out["scaling"] == "weak"
    assert out["combined_check"].startswith("sum of %d rank-local MSMs == single-rank MSM" % n)
    assert out["msm_strong_scaling"]["pairs_per_gpu"] in ((1 << 14) // n, (1 << 14) // n + 1)
    v = out["verify"]
    assert v["scaling"] == "strong" and v["job_proofs"] == 16 * n and v["batch_per_gpu"] == 16 and v["value"] > 0
    assert v["weak"]["batch_per_gpu"] == 16 * n and v["weak"]["job_proofs"] == 16 * n * n
    assert "rank %d of %d" % (tamper, n) in v["tamper_check"]
    v4 = v["other_shapes"][0]
    assert v4["scaling"] == "strong" and v4["job_proofs"] == 8 * n and v4["batch_per_gpu"] == 8 and "rank %d of %d" % (tamper, n) in v4["tamper_check"]
    assert out["prove"]["replicas"] == n


def test_bench_two_ranks_under_torch_distributed_run():
    port = 29600 + random.randrange(1500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log2n", "16", "--verify-batch", "0", "--backend", "gloo",
           "--same-device", "--no-cpu-baseline", "--msm-streams", "1"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = _json_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT))
    assert out["n_gpus"] == 2 and out["msm_strong_scaling"]["pairs_per_gpu"] == (1 << 16) // 2
