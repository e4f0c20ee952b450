#!/usr/bin/env python3
"""Generates tests/golden/*.json from the oracle (oracle/pyoracle.py + oracle/bppp_oracle.c).

The reference ships NO golden vectors, KATs or runnable tests for this path and cannot be built
here (no GHC), so these fixtures are outputs of the build's own restatement,
cross-checked (in this script) between the pure-Python big-int restatement and the independent C
restatement, and against OpenSSL for the MSM cases.  They pin regressions and give the GPU tests
data that does not depend on the oracle being importable; they do not pin the oracle itself
("parity unpinned" by the reference — see oracle/bppp_oracle.c header).

Run:  python tests/golden/make_golden.py      (deterministic; rewrites the JSON files)
"""
import json
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as O  # noqa: E402

H = lambda v: None if v is None else ([hex(v[0]), hex(v[1])] if isinstance(v, tuple) else hex(v))


def openssl_msm(sgs):
    exe = os.path.join(ROOT, "oracle", "_build", "openssl_check")
    if not os.path.exists(exe):
        return "skip"
    inp = f"{len(sgs)}\n" + "".join(f"{s:x} {(p or (0, 0))[0]:x} {(p or (0, 0))[1]:x}\n" for s, p in sgs)
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.split()
    x, y = int(out[0], 16), int(out[1], 16)
    return None if x == 0 and y == 0 else (x, y)


def main():
    ec, py = O.CEC(), O.PyEC()
    rnd = random.Random(0xB9B9)
    # ---- MSM (innerProduct, Commitment.hs:325-335)
    msm = []
    for n, seed in [(1, b"g1"), (2, b"g2"), (3, b"g3"), (17, b"g17"), (64, b"g64"), (200, b"g200")]:
        pts = O.hash_points(seed, n)
        sc = [rnd.randrange(O.N) for _ in range(n)]
        if n >= 3:
            sc[1] = 0
            pts[2] = None
        if n >= 17:
            sc[5] = O.N - 1
            sc[6] = (O.N + 1) // 2
            sc[7] = (O.N - 1) // 2
            pts[9] = pts[8]                      # repeated point
            sc[9] = (O.N - sc[8]) % O.N          # cancels term 8
        sgs = list(zip(sc, pts))
        want = ec.inner_product(sgs)
        if n <= 64:
            assert py.inner_product(sgs) == want
        chk = openssl_msm(sgs)
        assert chk == "skip" or chk == want
        msm.append({"scalars": [hex(s) for s in sc], "points": [H(p) for p in pts], "result": H(want)})
    json.dump({"doc": "sum_i s_i * P_i as canonical affine (null = infinity)", "cases": msm}, open(os.path.join(HERE, "msm.json"), "w"), indent=0)

    # ---- rationalReduceScalar (Commitment.hs:242-255) and collapsePoints (Bulletproof.hs:213-214)
    rr = []
    for x in [0, 1, 2, O.N - 1, O.N - 2, 2**128, 2**129 + 12345, (O.N - 1) // 2, (O.N + 1) // 2] + [rnd.randrange(O.N) for _ in range(40)]:
        a, b = O.rational_reduce_scalar(x)
        assert ec.rational_reduce(x) == (a, b) and (a - b * x) % O.N == 0
        rr.append({"x": hex(x), "a": str(a), "b": str(b)})
    folds = []
    for n, seed in [(1, b"f1"), (2, b"f2"), (5, b"f5"), (16, b"f16")]:
        pts = O.hash_points(seed, n)
        if n >= 5:
            pts[3] = None
        e = rnd.randrange(O.N)
        a, b = O.rational_reduce_scalar(e)
        out = [ec.pair_ip(b, pts[2 * j], a, pts[2 * j + 1] if 2 * j + 1 < n else None) for j in range((n + 1) // 2)]
        assert out == [py.pair_ip(b, pts[2 * j], a, pts[2 * j + 1] if 2 * j + 1 < n else None) for j in range((n + 1) // 2)]
        folds.append({"e": hex(e), "a": str(a), "b": str(b), "points": [H(p) for p in pts], "out": [H(p) for p in out]})
    json.dump({"doc": "rationalReduceScalar x -> (a, b); fold: out[j] = b*P[2j] + a*P[2j+1]", "rational_reduce": rr, "folds": folds},
              open(os.path.join(HERE, "fold.json"), "w"), indent=0)

    # ---- one full norm-linear argument transcript (proveBPM / verifyBPM, Bulletproof.hs:346-378)
    nl, ll = 13, 6
    g, *rest = O.hash_points(b"golden-basis", 1 + nl + ll)
    gs, hs = rest[:nl], rest[nl:]
    xs = [rnd.randrange(O.N) for _ in range(nl)]
    ls = [rnd.randrange(O.N) for _ in range(ll)]
    cs = [rnd.randrange(O.N) for _ in range(ll)]
    q = rnd.randrange(1, O.N)
    body = O.NormLinear.make(1, q, cs, xs, gs, ls, hs)
    wit = O.PSV(body.eval_scalar(), g, body)
    C = O.commit(wit.open_terms(), ec)
    rounds, (fn, fl) = O.optimal_witness_size_nl(nl, ll)
    rounds_log = []
    com = wit
    tr = O.Transcript(O.sha_oracle_fn())
    resps, es = [], []
    for _ in range(rounds):
        c = com.body
        sX, xw, sR, rw = c.make_scalars_coms()
        com, (X, R), e = O.prove_round(com, tr, ec)
        resps.insert(0, (X, R)); es.insert(0, e)
        rounds_log.append({"sX": hex(sX), "sR": hex(sR), "X": H(X), "R": H(R), "e": hex(e), "s_next": hex(com.sc),
                           "norm_x": [hex(v) for v, _ in com.body.norm.body], "norm_n": hex(com.body.norm.n), "norm_q": hex(com.body.norm.q),
                           "norm_g": [H(p) for _, p in com.body.norm.body],
                           "lin_c": [hex(c_) for c_, _, _ in com.body.lin.body], "lin_x": [hex(v) for _, v, _ in com.body.lin.body],
                           "lin_n": hex(com.body.lin.n), "lin_h": [H(p) for _, _, p in com.body.lin.body]})
    basis = O.PSV(0, g, O.NormLinear.make(1, q, [0] * ll, [0] * nl, gs, [0] * ll, hs))
    pub = O.PSV(0, g, O.NormLinear.make(1, q, cs, [0] * nl, [None] * nl, [0] * ll, [None] * ll))
    witb = O.NormLinear.make(1, 1, [], com.body.norm.get_witness(), [], com.body.lin.get_witness(), [])
    terms = O.verify_terms([(1, C)], es, resps, pub, basis, witb)
    assert O.commit(terms, ec) is None
    assert O.verify_bp([(1, C)], resps, pub, basis, witb, O.Transcript(O.sha_oracle_fn()), ec)
    json.dump({"doc": "one NormLinear bulletproof (NL flavour), oracle = pyoracle.sha_oracle_fn(b'bppp')",
               "q": hex(q), "g": H(g), "gs": [H(p) for p in gs], "hs": [H(p) for p in hs], "xs": [hex(v) for v in xs], "ls": [hex(v) for v in ls],
               "cs": [hex(v) for v in cs], "s": hex(wit.sc), "commitment": H(C), "n_rounds": rounds, "final_lens": [fn, fl],
               "rounds": rounds_log, "final_norm_witness": [hex(v) for v in com.body.norm.get_witness()],
               "final_lin_witness": [hex(v) for v in com.body.lin.get_witness()],
               "verifier_scalars": [hex(s) for s, _ in terms], "verifier_points": [H(p) for _, p in terms]},
              open(os.path.join(HERE, "bp_transcript.json"), "w"), indent=0)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
