"""Pins the oracle to the only known-answer material the reference holds for this path: its constants
(SURVEY.md Appendix C).  The reference has no golden vectors / runnable tests, so beyond
these the oracle is 'parity unpinned' and is cross-checked against OpenSSL and between its two
independent restatements instead."""
import pyoracle as O

# src/Data/Field/Galois/FastPrime/Internal.hs:108-116 ("3^160") and :118-126 ("(q+1238349833)/2"), 4 x Word64 LE
REF_3_160 = 3**160
REF_HALF = (O.N + 1238349833) // 2
# src/Data/Field/Galois/FastPrime/Internal.hs:48-51: 2^256 - n
REF_R = 0x14551231950B75FC4402DA1732FC9BEBF
# src/Data/Curve/Weierstrass/FastSECP256K1.hs:41, :56 (Eisenstein factorisations of p and n)
CHAR_EIS_FQ = (303414439467246543595250775667605759171, -64502973549206556628585045361533709078)


def test_moduli_and_generator():
    assert O.P == 2**256 - 2**32 - 977 and str(O.P).endswith("671663")
    assert str(O.N).endswith("494337")
    assert 2**256 - O.N == REF_R and REF_R**2 < 2 * O.P
    assert O.PyEC.on_curve((O.GX, O.GY))


def test_endomorphism_constants(oracle_lib):
    assert pow(O.BETA, 3, O.P) == 1 and O.BETA != 1
    assert pow(O.LAMBDA, 3, O.N) == 1 and O.LAMBDA != 1
    G = (O.GX, O.GY)
    want = (O.BETA * O.GX % O.P, O.GY)          # cmConj convention (src/Data/Curve/CM.hs:25-33)
    assert oracle_lib.mul(O.LAMBDA, G) == want
    assert O.PyEC().mul(O.LAMBDA, G) == want
    a, b = CHAR_EIS_FQ
    assert a * a - a * b + b * b == O.P


def test_field_test_values(oracle_lib):
    import ctypes
    U = ctypes.c_uint64
    lib = oracle_lib.lib

    def mul(a, b, which):
        out = (U * 4)()
        lib.orc_fe_mul((U * 4)(*O._to_limbs(a)), (U * 4)(*O._to_limbs(b)), out, which)
        return O._from_limbs(out)

    acc = 1
    for _ in range(160):
        acc = mul(acc, 3, 1)
    assert acc == REF_3_160 % O.N == REF_3_160      # 3^160 < 2^256 and < n
    assert mul(REF_HALF, 2, 1) == 1238349833          # 2 * (n + k)/2 = k mod n
    for m, which in ((O.P, 0), (O.N, 1)):
        for a, b in [(m - 1, m - 1), (m - 1, 2), (2**255 % m, 2**255 % m), (0, 5), (1, m - 1)]:
            assert mul(a, b, which) == a * b % m


def test_group_order(oracle_lib):
    G = (O.GX, O.GY)
    assert oracle_lib.mul(O.N - 1, G) == O.PyEC.neg(G)
    assert oracle_lib.inner_product([(O.N - 1, G), (1, G)]) is None
