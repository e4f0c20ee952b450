"""The oracle's restatement of the round functions (NormArgument.hs / Bulletproof.hs): algebraic closure
(prove -> verify = True, tampered = False), agreement of the two EC back-ends, reference shape tables."""
import random

import pytest

import pyoracle as O


def _instance(nl, ll, seed):
    rnd = random.Random(seed)
    g, *rest = O.hash_points(b"basis%d" % seed, 1 + nl + ll)
    gs, hs = rest[:nl], rest[nl:]
    xs = [rnd.randrange(O.N) for _ in range(nl)]
    ls = [rnd.randrange(O.N) for _ in range(ll)]
    cs = [rnd.randrange(O.N) for _ in range(ll)]
    q = rnd.randrange(1, O.N)
    return g, gs, hs, xs, ls, cs, q


def _prove_verify(nl, ll, seed, ec, tamper=None):
    g, gs, hs, xs, ls, cs, q = _instance(nl, ll, seed)
    body = O.NormLinear.make(1, q, cs, xs, gs, ls, hs)
    wit = O.PSV(body.eval_scalar(), g, body)
    C = O.commit(wit.open_terms(), ec)
    rounds, (fn, fl) = O.optimal_witness_size_nl(nl, ll)
    fin, resps, es = O.prove_bp(rounds, wit, O.Transcript(O.sha_oracle_fn()), ec)
    assert (len(fin.body.norm.body), len(fin.body.lin.body)) == (fn, fl)
    assert fin.sc == fin.body.eval_scalar()        # the round invariant s = evalScalar (Bulletproof.hs:352-353)
    nw, lw = fin.body.norm.get_witness(), fin.body.lin.get_witness()
    if tamper == "witness":
        nw[0] = (nw[0] + 1) % O.N
    if tamper == "response":
        resps[0] = (resps[0][1], resps[0][0])
    basis = O.PSV(0, g, O.NormLinear.make(1, q, [0] * ll, [0] * nl, gs, [0] * ll, hs))
    pub = O.PSV(0, g, O.NormLinear.make(1, q, cs, [0] * nl, [None] * nl, [0] * ll, [None] * ll))
    witb = O.NormLinear.make(1, 1, [], nw, [], lw, [])
    return O.verify_bp([(1, C)], resps, pub, basis, witb, O.Transcript(O.sha_oracle_fn()), ec)


@pytest.mark.parametrize("nl,ll", [(5, 1), (8, 5), (7, 3), (16, 9), (33, 6), (12, 12), (64, 7)])
def test_prove_verify_closes(oracle_lib, nl, ll):
    assert _prove_verify(nl, ll, nl * 100 + ll, oracle_lib)
    assert not _prove_verify(nl, ll, nl * 100 + ll, oracle_lib, tamper="witness")
    assert not _prove_verify(nl, ll, nl * 100 + ll, oracle_lib, tamper="response")


def test_python_and_c_backends_agree(oracle_lib):
    assert _prove_verify(6, 3, 1, O.PyEC())
    g, gs, hs, xs, ls, cs, q = _instance(6, 3, 2)
    body = O.NormLinear.make(1, q, cs, xs, gs, ls, hs)
    wit = O.PSV(body.eval_scalar(), g, body)
    a = O.prove_bp(2, wit, O.Transcript(O.sha_oracle_fn()), O.PyEC())
    b = O.prove_bp(2, wit, O.Transcript(O.sha_oracle_fn()), oracle_lib)
    assert a[1] == b[1] and a[2] == b[2] and a[0].body.norm.body == b[0].body.norm.body


def test_round_counts_match_reference_shapes():
    # SURVEY.md Appendix B (computed from Bulletproof.hs:300-316, NormArgument.hs:165-178)
    assert O.optimal_witness_size_nl(512, 261) == (8, (2, 2))     # examples/64by64
    assert O.optimal_witness_size_nl(1024, 261) == (9, (2, 1))    # examples/128by64
    assert O.optimal_witness_size_nl(384, 70) == (7, (3, 1))      # examples/32by64
    assert O.optimal_witness_size_nl(768, 261) == (8, (3, 2))     # examples/96by64
    assert O.optimal_witness_size_nl(192, 2) == (6, (3, 1))       # examples/bin_test
    assert O.optimal_witness_size_nl(1152, 261) == (9, (3, 1))    # 128by64 + "typed"
    assert [O.round_reduce(n) for n in (1, 2, 3, 4, 5, 261)] == [1, 1, 2, 2, 3, 131]


def test_tensor_list_equals_vector_instance():
    """tensor' for lists (Bulletproof.hs:94-95) vs the bit-indexed Data.Vector instance (:114-122)."""
    rnd = random.Random(4)
    for nb, k in [(1, 0), (3, 1), (2, 5), (5, 3)]:
        bs = [rnd.randrange(O.N) for _ in range(nb)]
        es = [rnd.randrange(O.N) for _ in range(k)]
        qs = [rnd.randrange(O.N) for _ in range(k)]
        got = O.tensor(bs, es, lambda r: qs[r])
        xs = list(zip(range(k), reversed(es), qs))
        want = []
        for n in range(nb << k):
            v = bs[n >> k]
            for kk, e, q in xs:
                v = v * (e if (n >> kk) & 1 else q) % O.N
            want.append(v)
        assert got == want


def test_vector_instance_equals_list_instance():
    """BPCollection V.Vector (Bulletproof.hs:102-162) and BPCollection [] (:68-99) are the same functions: tensor' and contract'
    restated from both instances agree on the shapes the verifier uses (final witness 2 x 8 rounds, 3 x 9 rounds) and on ragged ones."""
    rnd = random.Random(14)
    r = lambda n: [rnd.randrange(O.N) for _ in range(n)]
    for nb, k in [(1, 0), (2, 8), (3, 9), (2, 1), (5, 3), (1, 6)]:
        bs, es, qs = r(nb), r(k), r(k)
        assert O.tensor_vector(bs, es, qs) == O.tensor(bs, es, lambda i: qs[i])
    for nx, ny in [(1, 1), (4, 261), (2, 4), (3, 10), (8, 8), (5, 4)]:
        xs, ys = r(nx), r(ny)
        assert O.contract_vector(xs, ys) == O.contract(xs, ys)


def test_rational_reduce_properties(oracle_lib):
    rnd = random.Random(8)
    for x in [0, 1, O.N - 1, 2**128, 2**200] + [rnd.randrange(O.N) for _ in range(500)]:
        a, b = O.rational_reduce_scalar(x)
        assert (a - b * x) % O.N == 0 and a * a <= 2 * O.N
        assert abs(a).bit_length() <= 129 and abs(b).bit_length() <= 129     # fits the 129 rows of Commitment.hs:286
        assert oracle_lib.rational_reduce(x) == (a, b)


def test_batch_inverse_zero_maps_to_zero():
    xs = [5, 0, 7, O.N - 1, 0, 1]
    inv = O.batch_inverse(xs, O.N)
    assert inv[1] == 0 and inv[4] == 0
    assert all(x * y % O.N == 1 for x, y in zip(xs, inv) if x)


# ----------------------------------------------------------------------------- inner-product flavour
def _ip_prove_verify(nl, ll, seed, ec, tamper=False):
    rnd = random.Random(seed)
    g, *rest = O.hash_points(b"ipo%d" % seed, 1 + nl + ll)
    gs, hs = rest[:nl], rest[nl:]
    xs = [rnd.randrange(O.N) for _ in range(nl)]
    ls = [rnd.randrange(O.N) for _ in range(ll)]
    cs = [rnd.randrange(O.N) for _ in range(ll)]
    r = rnd.randrange(1, O.N)
    body = O.NormLinearIP.make(1, r, cs, xs, gs, ls, hs, ec)
    s = body.eval_scalar()
    # the weights the range proofs rely on: qPowers' _ q = powers' (negate q^2) (InnerProductArgument.hs:230-231)
    w = O.powers1((-r * r) % O.N, nl)
    assert s == (sum(wi * x * x for wi, x in zip(w, xs)) + sum(c * l for c, l in zip(cs, ls))) % O.N
    com = O.PSV(s, g, body)
    C = O.commit(com.open_terms(), ec)
    assert C == O.commit(list(zip(xs, gs)) + list(zip(ls, hs)) + [(s, g)], ec)      # the basis change preserves the commitment
    rounds, (fn, fl) = O.optimal_witness_size_ip(nl, ll)
    tr = O.Transcript(O.sha_oracle_fn())
    resps, es = [], []
    for _ in range(rounds):
        c = com.body
        sL, a, sR, b = c.make_scalars_coms()
        ac, bc = O.commit(O.PSV(sL, g, a).open_terms(), ec), O.commit(O.PSV(sR, g, b).open_terms(), ec)
        e = tr.oracle([ac, bc])
        e0, e1 = c.make_es(e)
        com = O.PSV((com.sc + e0 * sL + e1 * sR) % O.N, g, c.collapse(e, ec))
        assert com.sc == com.body.eval_scalar()
        resps.insert(0, (ac, bc)); es.insert(0, e)
    nw, lw = O.ip_norm_get_witness(com.body.norm), com.body.lin.get_witness()
    assert (len(nw), len(lw)) == (fn, fl)
    if tamper:
        lw = [(lw[0] + 1) % O.N] + lw[1:] if lw else lw
        nw = nw if lw else [(nw[0] + 1) % O.N] + nw[1:]
    basis = O.PSV(0, g, O.NormLinearIP.make(1, r, [0] * ll, [0] * nl, gs, [0] * ll, hs, ec))
    pub = O.PSV(0, g, O.NormLinearIP.make(1, r, cs, [0] * nl, [None] * nl, [0] * ll, [None] * ll, ec))
    witb = O.NormLinearIP.make(1, 1, [], nw, [], lw, [], ec)      # decodeProof' (RangeProof.hs:81)
    return O.commit(O.verify_terms_generic(O.NormLinearIP, [(1, C)], es, resps, pub, basis, witb), ec) is None


@pytest.mark.parametrize("nl,ll", [(11, 6), (16, 6), (8, 5), (62, 24), (10, 3)])
def test_ip_flavour_closes(oracle_lib, nl, ll):
    assert _ip_prove_verify(nl, ll, nl * 7 + ll, oracle_lib)
    assert not _ip_prove_verify(nl, ll, nl * 7 + ll, oracle_lib, tamper=True)


def test_ip_round_counts_match_reference_shapes():
    # examples/32bit, 64bit, rec_test (argument = IP by default, app/Parse.hs:100)
    assert O.optimal_witness_size_ip(11, 6) == (3, (2, 1))
    assert O.optimal_witness_size_ip(16, 6) == (3, (2, 1))
    assert O.optimal_witness_size_ip(62, 24) == (5, (2, 1))


# ----------------------------------------------------------------------------- GLV path (a6)
def test_glv_decomposition_and_inner_product(oracle_lib):
    """decomposeFastPrimeEis (FastPrime.hs:186-205) recomposes to the scalar with ~128-bit halves, and the 129-row
    GLV Straus loop (Commitment.hs:374-398) yields the same group element as the plain 256-row loop."""
    rnd = random.Random(6)
    for x in [0, 1, O.N - 1, O.LAMBDA, O.N - O.LAMBDA, 2**128, (O.N + 1) // 2] + [rnd.randrange(O.N) for _ in range(300)]:
        a, b = O.decompose_eis(x)
        assert (a + b * O.LAMBDA - x) % O.N == 0
        assert abs(a).bit_length() <= 129 and abs(b).bit_length() <= 129
    # conjEis charEis recomposes to 0 mod n (why reducedChar = conjEis . charEis, Commitment.hs:296-297)
    c = O.eis_conj(O.CHAR_EIS_FR)
    assert (c[0] + c[1] * O.LAMBDA) % O.N == 0
    assert O.CHAR_EIS_FR[0] ** 2 - O.CHAR_EIS_FR[0] * O.CHAR_EIS_FR[1] + O.CHAR_EIS_FR[1] ** 2 == O.N
    pts = O.hash_points(b"glv", 12)
    sgs = [(rnd.randrange(O.N), p) for p in pts]
    sgs[2] = (0, pts[2])
    sgs[3] = (rnd.randrange(O.N), None)
    sgs[4] = (O.N - 1, pts[4])
    assert O.glv_inner_product(sgs, O.PyEC()) == oracle_lib.inner_product(sgs)
