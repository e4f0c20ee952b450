"""Window width x slice length sweep for one MSM (bppp_msm_device), in one process.

    python benchmarks/sweep_window.py --log2n 12 13 14 --windows 9 10 11 12 13 --slices 4 8 16 32

Prints the wall-clock time per MSM for each (n, c, L) and the best setting per n; the plan heuristics in csrc/msm.hip
(choose_window, make_plan's slice length) are fitted to this table.  L is passed through the BPPP_LACC override.
"""
import argparse, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, nargs="+", default=[16])
    ap.add_argument("--windows", type=int, nargs="+", default=[0])
    ap.add_argument("--slices", type=int, nargs="+", default=[0])
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import bench
    from bulletproofspp_amd.capi import Bppp
    dev = torch.device("cuda:0")
    gpu = Bppp(0)
    nmax = 1 << max(args.log2n)
    dsc, dpts = bench.make_inputs(gpu, torch, dev, nmax, 7)
    for ln in args.log2n:
        n = 1 << ln
        best = None
        ref = None
        for c in args.windows:
            for L in args.slices:
                if L: os.environ["BPPP_LACC"] = str(L)
                else: os.environ.pop("BPPP_LACC", None)
                gpu.close(); gpu = Bppp(0)            # the tuning overrides are read when a context is created
                for _ in range(3): out = gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, c)
                if ref is None: ref = out
                assert out == ref, (ln, c, L)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.reps): gpu.msm_device(dsc.data_ptr(), dpts.data_ptr(), n, c)
                ms = (time.perf_counter() - t0) / args.reps * 1e3
                print(f"n=2^{ln} c={c} L={L} {ms:.4f} ms", flush=True)
                if best is None or ms < best[0]: best = (ms, c, L)
        print(f"best n=2^{ln}: c={best[1]} L={best[2]} {best[0]:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
