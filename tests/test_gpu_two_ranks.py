"""Multi-GPU readiness on ONE card: bench.py's N > 1 path with two ranks on cuda:0 (gloo for the 64-byte exchange, processes
started by torch.distributed.run before anything touches the GPU), every kernel the real HIP one.  The MSM leg shards the terms,
the verify leg shards the proofs; in both the ranks all-gather their partial points and add them (bulletproofspp_amd/dist.py).
Checked: the combined MSM point equals the single-rank MSM over all terms (--check-combined), every rank's batch verifies and the
combined verification point is the identity (asserted inside bench.py), and the JSON line reports n_gpus = 2."""
import json
import os
import random
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    port = 29600 + random.randrange(1500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--log2n", "17", "--verify-batch", "256", "--backend", "gloo",
           "--same-device", "--check-combined", "--no-cpu-baseline", "--msm-streams", "1"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["combined_check"].startswith("sum of 2 rank-local MSMs == single-rank MSM")
    assert out["verify"]["batch_per_gpu"] == 256 and out["verify"]["value"] > 0
    assert out["prove"]["replicas"] == 2
