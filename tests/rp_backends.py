"""Test-side CPU backend for bulletproofspp_amd.rangeproof: the curve operations of the range-proof protocol done by the
oracle (oracle/pyoracle.py + the C restatement), so that the GPU backend can be checked against it transcript for transcript.
TEST INFRASTRUCTURE ONLY — the package itself has no CPU path."""
import pyoracle as O
from bulletproofspp_amd.rangeproof import Backend


class _Tr:
    def __init__(self, fn):
        self.fn = fn

    def oracle(self, xs):
        return self.fn(list(xs))


class OracleBackend(Backend):
    def __init__(self, ec):
        self.ec = ec

    def commit(self, scalars, points):
        return self.ec.inner_product(list(zip([s % O.N for s in scalars], points)))

    def prove_bp(self, flavour, n_rounds, sc, g, q, cs, nrm, gs, lin, hs, oracle1):
        if flavour == "NL":
            com = O.PSV(sc % O.N, g, O.NormLinear.make(1, q, cs, nrm, gs, lin, hs))
        else:
            com = O.PSV(sc % O.N, g, O.NormLinearIP.make(1, q, cs, nrm, gs, lin, hs, self.ec))
        final, resps, _ = O.prove_bp(n_rounds, com, _Tr(oracle1), self.ec)
        if flavour == "NL":
            return resps, final.body.norm.get_witness(), final.body.lin.get_witness()
        w = final.body.get_witness()
        nl = 2 * len(final.body.norm.body)
        return resps, w[:nl], w[nl:]

    def verify_bp(self, flavour, q, sp, g, pub_nrm, gs, cs, pub_lin, hs, es, responses, wit_nrm, wit_lin, init_terms):
        nl, ll = len(gs), len(hs)
        pad = lambda xs, n: list(xs) + [0] * (n - len(xs))
        if flavour == "NL":
            basis = O.PSV(0, g, O.NormLinear.make(1, q, [0] * ll, [0] * nl, gs, [0] * ll, hs))
            pub = O.PSV(sp % O.N, g, O.NormLinear.make(1, q, pad(cs, ll), pad(pub_nrm, nl), [None] * nl, pad(pub_lin, ll), [None] * ll))
            witb = O.NormLinear.make(1, 1, [], wit_nrm, [], wit_lin, [])      # decodeProof' (RangeProof.hs:81)
            return O.commit(O.verify_terms(init_terms, es, responses, pub, basis, witb), self.ec) is None
        basis = O.PSV(0, g, O.NormLinearIP.make(1, q, [0] * ll, [0] * nl, gs, [0] * ll, hs, self.ec))
        pub = O.PSV(sp % O.N, g, O.NormLinearIP.make(1, q, pad(cs, ll), pad(pub_nrm, nl), [None] * nl, pad(pub_lin, ll), [None] * ll, self.ec))
        witb = O.NormLinearIP.make(1, 1, [], wit_nrm, [], wit_lin, [], self.ec)
        return O.commit(O.verify_terms_generic(O.NormLinearIP, init_terms, es, responses, pub, basis, witb), self.ec) is None
