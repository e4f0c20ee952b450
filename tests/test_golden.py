"""Committed golden fixtures (tests/golden/*.json, made by tests/golden/make_golden.py): the oracle must
reproduce them on CPU; the GPU tests check the HIP path against the same files."""
import json
import os

import numpy as np
import pytest

import pyoracle as O
from bulletproofspp_amd.capi import points_to_array, scalars_to_array, array_to_point, array_to_scalars

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
pt = lambda v: None if v is None else (int(v[0], 16), int(v[1], 16))
load = lambda name: json.load(open(os.path.join(G, name)))


def test_oracle_reproduces_msm_golden(oracle_lib):
    for case in load("msm.json")["cases"]:
        sgs = [(int(s, 16), pt(p)) for s, p in zip(case["scalars"], case["points"])]
        assert oracle_lib.inner_product(sgs) == pt(case["result"])


def test_oracle_reproduces_fold_golden(oracle_lib):
    d = load("fold.json")
    for r in d["rational_reduce"]:
        assert O.rational_reduce_scalar(int(r["x"], 16)) == (int(r["a"]), int(r["b"]))
    for f in d["folds"]:
        pts = [pt(p) for p in f["points"]]
        a, b = int(f["a"]), int(f["b"])
        out = [oracle_lib.pair_ip(b, pts[2 * j], a, pts[2 * j + 1] if 2 * j + 1 < len(pts) else None) for j in range((len(pts) + 1) // 2)]
        assert out == [pt(p) for p in f["out"]]


def test_oracle_reproduces_transcript_golden(oracle_lib):
    d = load("bp_transcript.json")
    terms = [(int(s, 16), pt(p)) for s, p in zip(d["verifier_scalars"], d["verifier_points"])]
    assert oracle_lib.inner_product(terms) is None            # the verifier's single MSM is infinity
    gs, hs = [pt(p) for p in d["gs"]], [pt(p) for p in d["hs"]]
    body = O.NormLinear.make(1, int(d["q"], 16), [int(v, 16) for v in d["cs"]], [int(v, 16) for v in d["xs"]], gs,
                             [int(v, 16) for v in d["ls"]], hs)
    wit = O.PSV(int(d["s"], 16), pt(d["g"]), body)
    assert O.commit(wit.open_terms(), oracle_lib) == pt(d["commitment"])
    fin, resps, es = O.prove_bp(d["n_rounds"], wit, O.Transcript(O.sha_oracle_fn()), oracle_lib)
    assert [hex(e) for e in reversed(es)] == [r["e"] for r in d["rounds"]]
    assert fin.body.norm.get_witness() == [int(v, 16) for v in d["final_norm_witness"]]


@pytest.mark.gpu
def test_gpu_matches_msm_golden(gpu):
    for case in load("msm.json")["cases"]:
        sc = [int(s, 16) for s in case["scalars"]]
        pts = [pt(p) for p in case["points"]]
        assert gpu.msm(scalars_to_array(sc), points_to_array(pts)) == pt(case["result"])


@pytest.mark.gpu
def test_gpu_matches_fold_golden(gpu):
    d = load("fold.json")
    for r in d["rational_reduce"]:
        assert gpu.rational_reduce(int(r["x"], 16)) == (int(r["a"]), int(r["b"]))
    for f in d["folds"]:
        pts = [pt(p) for p in f["points"]]
        got = gpu.fold_points(int(f["b"]), int(f["a"]), points_to_array(pts))
        assert [array_to_point(got[j]) for j in range(len(f["out"]))] == [pt(p) for p in f["out"]]


@pytest.mark.gpu
def test_gpu_verifier_msm_of_golden_transcript_is_infinity(gpu):
    d = load("bp_transcript.json")
    sc = [int(s, 16) for s in d["verifier_scalars"]]
    pts = [pt(p) for p in d["verifier_points"]]
    assert gpu.msm(scalars_to_array(sc), points_to_array(pts)) is None
    sc[0] = (sc[0] + 1) % O.N
    assert gpu.msm(scalars_to_array(sc), points_to_array(pts)) is not None
