"""proveBPM / verifyBPM with device-resident vectors (bppp_nl_*) vs the oracle's restatement of
src/Bulletproof.hs:346-378 and src/Bulletproof/NormArgument.hs, round by round: bit-exact scalars and points."""
import json
import os
import random

import pytest

import pyoracle as O
from bulletproofspp_amd.bulletproof import NormLinearBP, proveBPM, verifyBPM

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
pt = lambda v: None if v is None else (int(v[0], 16), int(v[1], 16))


def test_golden_transcript_round_by_round(gpu):
    d = json.load(open(os.path.join(G, "bp_transcript.json")))
    h = lambda k: [int(v, 16) for v in d[k]]
    gs, hs = [pt(p) for p in d["gs"]], [pt(p) for p in d["hs"]]
    com = NormLinearBP(gpu, int(d["s"], 16), pt(d["g"]), int(d["q"], 16), h("cs"), h("xs"), gs, h("ls"), hs)
    for r in d["rounds"]:
        sX, X, sR, R = com.makeScalarsComs()
        assert (hex(sX), hex(sR)) == (r["sX"], r["sR"])
        assert X == pt(r["X"]) and R == pt(r["R"])
        com.collapse(int(r["e"], 16))
        st = com.download()
        assert [hex(v) for v in st["norm_x"]] == r["norm_x"] and st["norm_g"] == [pt(p) for p in r["norm_g"]]
        assert [hex(v) for v in st["lin_c"]] == r["lin_c"] and [hex(v) for v in st["lin_x"]] == r["lin_x"]
        assert st["lin_h"] == [pt(p) for p in r["lin_h"]]
        assert hex(st["s"]) == r["s_next"] and hex(st["norm_n"]) == r["norm_n"] and hex(st["norm_q"] if "norm_q" in st else st["q"]) == r["norm_q"]
        assert hex(st["lin_n"]) == r["lin_n"]
    nw, lw = com.getWitness()
    assert [hex(v) for v in nw] == d["final_norm_witness"] and [hex(v) for v in lw] == d["final_lin_witness"]
    com.close()


def _instance(nl, ll, seed):
    rnd = random.Random(seed)
    g, *rest = O.hash_points(b"gpubp%d" % seed, 1 + nl + ll)
    return g, rest[:nl], rest[nl:], [rnd.randrange(O.N) for _ in range(nl)], [rnd.randrange(O.N) for _ in range(ll)], \
        [rnd.randrange(O.N) for _ in range(ll)], rnd.randrange(1, O.N)


@pytest.mark.parametrize("nl,ll", [(5, 1), (8, 5), (7, 3), (33, 6), (64, 7), (1, 1), (0, 6), (9, 0)])
def test_prove_on_gpu_equals_oracle_and_verifies(gpu, oracle_lib, nl, ll):
    g, gs, hs, xs, ls, cs, q = _instance(nl, ll, 31 * nl + ll)
    body = O.NormLinear.make(1, q, cs, xs, gs, ls, hs)
    wit = O.PSV(body.eval_scalar(), g, body)
    C = O.commit(wit.open_terms(), oracle_lib)
    rounds, (fn, fl) = O.optimal_witness_size_nl(nl, ll)
    rounds = max(rounds, 1)
    fin, resps_o, es_o = O.prove_bp(rounds, wit, O.Transcript(O.sha_oracle_fn()), oracle_lib)
    # GPU prover with the same injected oracle
    com = NormLinearBP(gpu, wit.sc, g, q, cs, xs, gs, ls, hs)
    tr = O.Transcript(O.sha_oracle_fn())
    resps, es = proveBPM(rounds, com, tr.oracle)
    assert resps == resps_o and es == es_o
    nw, lw = com.getWitness()
    assert nw == fin.body.norm.get_witness() and lw == fin.body.lin.get_witness()
    assert com.download()["s"] == fin.sc
    com.close()
    # GPU verifier: expandChallenges + the single MSM
    zeros_n, zeros_l = [0] * nl, [0] * ll
    ok = verifyBPM(gpu, q, 0, g, zeros_n, gs, cs, zeros_l, hs, es, resps, nw, lw, [(1, C)])
    assert ok
    # the verifier's term list equals the oracle's (scalars bit-for-bit), checked through the result of a perturbed run
    if nw:
        bad = [(nw[0] + 1) % O.N] + nw[1:]
        assert not verifyBPM(gpu, q, 0, g, zeros_n, gs, cs, zeros_l, hs, es, resps, bad, lw, [(1, C)])
    assert not verifyBPM(gpu, q, 1, g, zeros_n, gs, cs, zeros_l, hs, es, resps, nw, lw, [(1, C)])
    if len(resps) > 1:
        swapped = [resps[1], resps[0]] + resps[2:]
        assert not verifyBPM(gpu, q, 0, g, zeros_n, gs, cs, zeros_l, hs, es, swapped, nw, lw, [(1, C)])


def test_verify_with_nonzero_public_vectors(gpu, oracle_lib):
    """pub (publicCs) non-zero as the range proofs use it (TypedReciprocal.hs:356-357): witness = pub + hidden part."""
    nl, ll = 12, 5
    g, gs, hs, xs, ls, cs, q = _instance(nl, ll, 77)
    rnd = random.Random(5)
    pub_n = [rnd.randrange(O.N) for _ in range(nl)]
    pub_l = [rnd.randrange(O.N) for _ in range(ll)]
    body = O.NormLinear.make(1, q, cs, xs, gs, ls, hs)
    wit = O.PSV(body.eval_scalar(), g, body)
    # initCom commits to the hidden part: total - pub on every basis element, and s on g minus sp
    sp = rnd.randrange(O.N)
    hidden = [((x - p) % O.N, G_) for x, p, G_ in zip(xs, pub_n, gs)] + [((x - p) % O.N, H_) for x, p, H_ in zip(ls, pub_l, hs)] + [((wit.sc - sp) % O.N, g)]
    C = O.commit(hidden, oracle_lib)
    rounds, _ = O.optimal_witness_size_nl(nl, ll)
    com = NormLinearBP(gpu, wit.sc, g, q, cs, xs, gs, ls, hs)
    tr = O.Transcript(O.sha_oracle_fn())
    resps, es = proveBPM(rounds, com, tr.oracle)
    nw, lw = com.getWitness()
    com.close()
    assert verifyBPM(gpu, q, sp, g, pub_n, gs, cs, pub_l, hs, es, resps, nw, lw, [(1, C)])
    # same through the oracle's verifier
    basis = O.PSV(0, g, O.NormLinear.make(1, q, [0] * ll, [0] * nl, gs, [0] * ll, hs))
    pub = O.PSV(sp, g, O.NormLinear.make(1, q, cs, pub_n, [None] * nl, pub_l, [None] * ll))
    witb = O.NormLinear.make(1, 1, [], nw, [], lw, [])
    assert O.commit(O.verify_terms([(1, C)], es, resps, pub, basis, witb), oracle_lib) is None


def test_examples_64by64_shape(gpu, oracle_lib):
    """BASELINE config 3 shape: nrmLen 512, linLen 261, 8 rounds — prover rounds on the GPU,
    final opening and every response equal to the oracle's."""
    nl, ll = 512, 261
    g, gs, hs, xs, ls, cs, q = _instance(nl, ll, 6464)
    body = O.NormLinear.make(1, q, cs, xs, gs, ls, hs)
    wit = O.PSV(body.eval_scalar(), g, body)
    rounds, (fn, fl) = O.optimal_witness_size_nl(nl, ll)
    assert (rounds, fn, fl) == (8, 2, 2)
    fin, resps_o, es_o = O.prove_bp(rounds, wit, O.Transcript(O.sha_oracle_fn()), oracle_lib)
    com = NormLinearBP(gpu, wit.sc, g, q, cs, xs, gs, ls, hs)
    resps, es = proveBPM(rounds, com, O.Transcript(O.sha_oracle_fn()).oracle)
    assert resps == resps_o and es == es_o
    nw, lw = com.getWitness()
    assert (nw, lw) == (fin.body.norm.get_witness(), fin.body.lin.get_witness())
    com.close()
    C = O.commit(wit.open_terms(), oracle_lib)
    assert verifyBPM(gpu, q, 0, g, [0] * nl, gs, cs, [0] * ll, hs, es, resps, nw, lw, [(1, C)])


def test_batch_verifier(gpu, oracle_lib):
    """SURVEY.md 8(c) pins for the batch verifier: (ii) each valid proof's MSM is infinity, (iii) the random linear
    combination is infinity iff all are valid, and (i) the combined result equals the oracle's sum_k rho_k * MSM(T_k)
    when one proof is corrupted (a non-infinity point that depends on every per-proof scalar)."""
    from bulletproofspp_amd.bulletproof import verifyBatch
    nl, ll, B = 12, 5, 7
    g, gs, hs, _, _, _, _ = _instance(nl, ll, 4242)
    rnd = random.Random(99)
    proofs, oracle_terms = [], []
    for b in range(B):
        xs = [rnd.randrange(O.N) for _ in range(nl)]
        ls = [rnd.randrange(O.N) for _ in range(ll)]
        cs = [rnd.randrange(O.N) for _ in range(ll)]
        q = rnd.randrange(1, O.N)
        pub_n = [rnd.randrange(O.N) for _ in range(nl)]
        pub_l = [rnd.randrange(O.N) for _ in range(ll)]
        sp = rnd.randrange(O.N)
        body = O.NormLinear.make(1, q, cs, xs, gs, ls, hs)
        wit = O.PSV(body.eval_scalar(), g, body)
        hidden = [((x - p) % O.N, G_) for x, p, G_ in zip(xs, pub_n, gs)] + [((x - p) % O.N, H_) for x, p, H_ in zip(ls, pub_l, hs)] + [((wit.sc - sp) % O.N, g)]
        C = O.commit(hidden, oracle_lib)
        t = rnd.randrange(1, O.N)          # initCom as a 2-term opening: t * (t^-1 C1) + 1 * C2 with C1 + C2 = C
        C1 = oracle_lib.mul(rnd.randrange(1, O.N), g)
        C2 = oracle_lib.add(C, O.PyEC.neg(C1))
        init = [(t, oracle_lib.mul(O.inv_mod(t, O.N), C1)), (1, C2)]
        rounds, _ = O.optimal_witness_size_nl(nl, ll)
        fin, resps, es = O.prove_bp(rounds, wit, O.Transcript(O.sha_oracle_fn(b"p%d" % b)), oracle_lib)
        nw, lw = fin.body.norm.get_witness(), fin.body.lin.get_witness()
        proofs.append({"q": q, "sp": sp, "pub_norm": pub_n, "pub_lin_c": cs, "pub_lin_x": pub_l, "es": es, "responses": resps,
                       "wit_norm": nw, "wit_lin": lw, "init_terms": init})
        basis = O.PSV(0, g, O.NormLinear.make(1, q, [0] * ll, [0] * nl, gs, [0] * ll, hs))
        pub = O.PSV(sp, g, O.NormLinear.make(1, q, cs, pub_n, [None] * nl, pub_l, [None] * ll))
        oracle_terms.append(lambda nw=nw, lw=lw, es=es, resps=resps, pub=pub, basis=basis, init=init:
                            O.verify_terms(init, es, resps, pub, basis, O.NormLinear.make(1, 1, [], nw, [], lw, [])))
        assert O.commit(oracle_terms[-1](), oracle_lib) is None
    rhos = [1] + [rnd.randrange(1, O.N) for _ in range(B - 1)]
    assert verifyBatch(gpu, proofs, g, gs, hs, rhos)
    assert verifyBatch(gpu, proofs[:1], g, gs, hs, [1])
    # corrupt one proof's opening: batch must reject
    bad = [dict(p) for p in proofs]
    bad[3]["wit_lin"] = [(bad[3]["wit_lin"][0] + 5) % O.N] + bad[3]["wit_lin"][1:]
    assert not verifyBatch(gpu, bad, g, gs, hs, rhos)
    bad = [dict(p) for p in proofs]
    bad[B - 1]["sp"] = (bad[B - 1]["sp"] + 1) % O.N
    assert not verifyBatch(gpu, bad, g, gs, hs, rhos)


# ----------------------------------------------------------------------------- inner-product flavour (a12)
@pytest.mark.parametrize("nl,ll", [(11, 6), (16, 6), (8, 5), (62, 24), (10, 3), (5, 0), (0, 7)])
def test_ip_flavour_prove_and_verify(gpu, oracle_lib, nl, ll):
    """src/Bulletproof/InnerProductArgument.hs on the GPU vs the oracle's restatement, round by round.
    Shapes include examples/32bit (11, 6), examples/64bit (16, 6) and examples/rec_test (62, 24)."""
    from bulletproofspp_amd.bulletproof import NormLinearIP, verifyBPM_IP
    ec = oracle_lib
    g, gs, hs, xs, ls, cs, r = _instance(nl, ll, 500 + 31 * nl + ll)
    body = O.NormLinearIP.make(1, r, cs, xs, gs, ls, hs, ec)
    s = body.eval_scalar()
    com_o = O.PSV(s, g, body)
    C = O.commit(com_o.open_terms(), ec)
    rounds, (fn, fl) = O.optimal_witness_size_ip(nl, ll)
    rounds = max(rounds, 1)
    dev = NormLinearIP(gpu, s, g, r, cs, xs, gs, ls, hs)
    tr = O.Transcript(O.sha_oracle_fn())
    resps, es = [], []
    for _ in range(rounds):
        c = com_o.body
        sL, a, sR, b = c.make_scalars_coms()
        ac = O.commit(O.PSV(sL, g, a).open_terms(), ec)
        bc = O.commit(O.PSV(sR, g, b).open_terms(), ec)
        dsL, dL, dsR, dR = dev.makeScalarsComs()
        assert (dsL, dL, dsR, dR) == (sL, ac, sR, bc)
        e = tr.oracle([ac, bc])
        e0, e1 = c.make_es(e)
        com_o = O.PSV((com_o.sc + e0 * sL + e1 * sR) % O.N, g, c.collapse(e, ec))
        dev.collapse(e)
        resps.insert(0, (ac, bc)); es.insert(0, e)
        nw_d, lw_d, s_d = dev.getWitness()
        assert nw_d == O.ip_norm_get_witness(com_o.body.norm) and lw_d == com_o.body.lin.get_witness() and s_d == com_o.sc
    nw, lw, _ = dev.getWitness()
    dev.close()
    assert verifyBPM_IP(gpu, r, 0, g, [0] * nl, gs, cs, [0] * ll, hs, es, resps, nw, lw, [(1, C)])
    if nw:
        assert not verifyBPM_IP(gpu, r, 0, g, [0] * nl, gs, cs, [0] * ll, hs, es, resps, [(nw[0] + 1) % O.N] + nw[1:], lw, [(1, C)])
    assert not verifyBPM_IP(gpu, r, 3, g, [0] * nl, gs, cs, [0] * ll, hs, es, resps, nw, lw, [(1, C)])


def test_prove_loop_in_library_with_oracle_callback(gpu, oracle_lib):
    """bppp_nl_prove: proveBPM's loop (Bulletproof.hs:357-359) in C++ with the oracle injected as a callback that receives
    the whole transcript, newest first (ZKP.hs:96-101) — same responses / challenges / final opening as the oracle."""
    from bulletproofspp_amd.bulletproof import proveBPM_native
    nl, ll = 33, 6
    g, gs, hs, xs, ls, cs, q = _instance(nl, ll, 9001)
    body = O.NormLinear.make(1, q, cs, xs, gs, ls, hs)
    wit = O.PSV(body.eval_scalar(), g, body)
    rounds, _ = O.optimal_witness_size_nl(nl, ll)
    pre = O.hash_points(b"pre", 3)                       # commitments made before the argument (e.g. by a range proof)
    tr = O.Transcript(O.sha_oracle_fn())
    tr.cs = list(pre)
    fin, resps_o, es_o = O.prove_bp(rounds, wit, tr, oracle_lib)
    com = NormLinearBP(gpu, wit.sc, g, q, cs, xs, gs, ls, hs)
    resps, es, transcript = proveBPM_native(rounds, com, O.sha_oracle_fn(), transcript=pre)
    assert resps == resps_o and es == es_o and transcript == tr.cs
    nw, lw = com.getWitness()
    assert (nw, lw) == (fin.body.norm.get_witness(), fin.body.lin.get_witness())
    com.close()


@pytest.mark.parametrize("nl,ll,B", [(12, 5, 3), (33, 6, 5), (64, 7, 2), (7, 0, 2), (0, 6, 2)])
def test_lockstep_batch_prover_equals_oracle(gpu, oracle_lib, nl, ll, B):
    """bppp_nlb_*: B proofs advanced together; each proof's responses, challenges and final opening equal the oracle's
    single-proof run with that proof's own injected oracle."""
    from bulletproofspp_amd.bulletproof import NormLinearBatch, proveBPM_batch
    g, gs, hs, _, _, _, _ = _instance(nl, ll, 7000 + nl)
    rnd = random.Random(B * 100 + nl)
    xs = [[rnd.randrange(O.N) for _ in range(nl)] for _ in range(B)]
    ls = [[rnd.randrange(O.N) for _ in range(ll)] for _ in range(B)]
    cs = [[rnd.randrange(O.N) for _ in range(ll)] for _ in range(B)]
    qs = [rnd.randrange(1, O.N) for _ in range(B)]
    rounds, _ = O.optimal_witness_size_nl(nl, ll)
    rounds = max(rounds, 1)
    want, ss = [], []
    for b in range(B):
        body = O.NormLinear.make(1, qs[b], cs[b], xs[b], gs, ls[b], hs)
        wit = O.PSV(body.eval_scalar(), g, body)
        ss.append(wit.sc)
        want.append(O.prove_bp(rounds, wit, O.Transcript(O.sha_oracle_fn(b"b%d" % b)), oracle_lib))
    com = NormLinearBatch(gpu, ss, g, qs, cs, xs, gs, ls, hs)
    trs = [O.Transcript(O.sha_oracle_fn(b"b%d" % b)) for b in range(B)]
    resps, es = proveBPM_batch(rounds, com, [t.oracle for t in trs])
    nws, lws, s_fin = com.getWitness()
    com.close()
    for b in range(B):
        fin, resps_o, es_o = want[b]
        assert resps[b] == resps_o and es[b] == es_o
        assert nws[b] == fin.body.norm.get_witness() and lws[b] == fin.body.lin.get_witness() and s_fin[b] == fin.sc
