"""The single-launch route (k_msm_small over windows x slices, csrc/msm.hip) against the general pipeline, for one MSM of n terms.

    python benchmarks/sweep_small_msm.py --n 858 4096 22016 65536 --c 6 7 8 --len 512 1024 2048

Prints the wall-clock time per MSM for the general pipeline and for each (c, slice length); MSM_SMALL_DEFAULT_MAX and the defaults in
msm_run_ex are read off this table.  Every setting must return the same point.
"""
import argparse, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[858, 4096, 22016, 65536])
    ap.add_argument("--c", type=int, nargs="+", default=[6, 7, 8])
    ap.add_argument("--len", type=int, nargs="+", default=[1024])
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    import torch
    import bench
    from bulletproofspp_amd.capi import Bppp
    dev = torch.device("cuda:0")
    gpu = Bppp(0)
    dsc, dpts = bench.make_inputs(gpu, torch, dev, max(args.n), 7)
    keys = ["BPPP_MSM_NO_SMALL", "BPPP_MSM_SMALL_C", "BPPP_MSM_SMALL_LEN", "BPPP_MSM_SMALL_MAX"]

    def run(n, env):
        nonlocal gpu
        for k in keys: os.environ.pop(k, None)
        os.environ.update(env)
        gpu.close(); gpu = Bppp(0)                # the tuning overrides are read when a context is created
        for _ in range(3): out = gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps): gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, 0)
        return out, (time.perf_counter() - t0) / args.reps * 1e3

    for n in args.n:
        ref, ms = run(n, {"BPPP_MSM_NO_SMALL": "1"})
        print(f"n={n} general pipeline {ms:.4f} ms", flush=True)
        best = None
        for c in args.c:
            for ln in args.len:
                out, ms = run(n, {"BPPP_MSM_SMALL_C": str(c), "BPPP_MSM_SMALL_LEN": str(ln), "BPPP_MSM_SMALL_MAX": str(1 << 30)})
                assert out == ref, (n, c, ln)
                print(f"n={n} sliced c={c} len={ln} {ms:.4f} ms", flush=True)
                if best is None or ms < best[0]: best = (ms, c, ln)
        print(f"best n={n}: c={best[1]} len={best[2]} {best[0]:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
