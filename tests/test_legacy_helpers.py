"""The torus-domain helpers of csrc/host/mosfhet_compat_legacy.c against the REAL reference library (oracle/_ref, built from /root/reference by
oracle/ref/Makefile): same struct layouts, same inputs, bit-identical outputs.  These helpers are exact integer arithmetic on host structs (the
rows of SURVEY section 8 that sit either side of the bootstrap: a9 rotations, a11 / a24 digits, a19 LUT packing, TRGSW constructors); no GPU needed."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Poly(C.Structure):
    _fields_ = [("coeffs", C.POINTER(C.c_uint64)), ("N", C.c_int)]


class Trlwe(C.Structure):
    _fields_ = [("a", C.POINTER(C.POINTER(Poly))), ("b", C.POINTER(Poly)), ("k", C.c_int)]


class Trgsw(C.Structure):
    _fields_ = [("samples", C.POINTER(C.POINTER(Trlwe))), ("l", C.c_int), ("Bg_bit", C.c_int)]


class Lib:
    """One library exporting the reference's names (ours or the reference's own build), with typed constructors."""

    def __init__(self, cdll):
        self.l = L = cdll
        L.polynomial_new_torus_polynomial.restype = C.POINTER(Poly)
        L.polynomial_new_torus_polynomial.argtypes = [C.c_int]
        L.trlwe_alloc_new_sample.restype = C.POINTER(Trlwe)
        L.trlwe_alloc_new_sample.argtypes = [C.c_int, C.c_int]
        L.trgsw_alloc_new_sample.restype = C.POINTER(Trgsw)
        L.trgsw_alloc_new_sample.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]

    def poly(self, values):
        p = self.l.polynomial_new_torus_polynomial(len(values))
        np.ctypeslib.as_array(p.contents.coeffs, (len(values),))[:] = values
        return p

    def trlwe(self, values):             # values: [k+1][N]
        k, N = values.shape[0] - 1, values.shape[1]
        c = self.l.trlwe_alloc_new_sample(k, N)
        for p in range(k):
            np.ctypeslib.as_array(c.contents.a[p].contents.coeffs, (N,))[:] = values[p]
        np.ctypeslib.as_array(c.contents.b.contents.coeffs, (N,))[:] = values[k]
        return c

    def trgsw(self, values, Bg_bit):     # values: [(k+1) l][k+1][N]
        rows, k1, N = values.shape
        l = rows // k1
        g = self.l.trgsw_alloc_new_sample(l, Bg_bit, k1 - 1, N)
        for r in range(rows):
            s = g.contents.samples[r].contents
            for p in range(k1 - 1):
                np.ctypeslib.as_array(s.a[p].contents.coeffs, (N,))[:] = values[r, p]
            np.ctypeslib.as_array(s.b.contents.coeffs, (N,))[:] = values[r, k1 - 1]
        return g


def get_poly(p):
    return np.ctypeslib.as_array(p.contents.coeffs, (p.contents.N,)).copy()


def get_trlwe(c):
    k, N = c.contents.k, c.contents.b.contents.N
    return np.stack([np.ctypeslib.as_array(c.contents.a[p].contents.coeffs, (N,)).copy() for p in range(k)]
                    + [np.ctypeslib.as_array(c.contents.b.contents.coeffs, (N,)).copy()])


def get_trgsw(g, k=1):
    rows = (k + 1) * g.contents.l
    return np.stack([get_trlwe(g.contents.samples[r]) for r in range(rows)])


@pytest.fixture(scope="module")
def libs(native_lib):
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libmosfhet_ref_ffnt.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref not built (needs /root/reference: make -C oracle/ref)")
    from mosfhet_amd import engine
    ours = Lib(engine.lib())
    ref = Lib(C.CDLL(ref_path, mode=os.RTLD_LOCAL | os.RTLD_NOW))
    return ours, ref


def rnd(rng, *shape):
    return rng.integers(0, 2 ** 64, size=shape, dtype=np.uint64)


def both(libs, fn):
    """run fn(lib) on our library and on the reference and require identical results"""
    got, want = fn(libs[0]), fn(libs[1])
    assert got.shape == want.shape and (got == want).all()
    return got


N = 64


def test_rotations_match_the_reference(libs):
    rng = np.random.default_rng(1)
    x, y = rnd(rng, N), rnd(rng, N)
    for name in ("torus_polynomial_mul_by_xai", "torus_polynomial_mul_by_xai_addto", "torus_polynomial_mul_by_xai_minus_1"):
        for a in (0, 1, N - 1, N, N + 1, 2 * N - 1, 2 * N + 3, 37, -5):
            def run(lib):
                out, inp = lib.poly(y), lib.poly(x)
                getattr(lib.l, name)(out, inp, C.c_int(a))
                return get_poly(out)
            both(libs, run)
    c, d = rnd(rng, 2, N), rnd(rng, 2, N)
    for name in ("trlwe_mul_by_xai", "trlwe_mul_by_xai_addto", "trlwe_mul_by_xai_minus_1"):
        for a in (0, 3, N, 2 * N - 1):
            def run(lib):
                out, inp = lib.trlwe(d), lib.trlwe(c)
                getattr(lib.l, name)(out, inp, C.c_int(a))
                return get_trlwe(out)
            both(libs, run)


@pytest.mark.parametrize("Bg_bit,l", [(8, 2), (9, 4), (23, 1), (4, 8)])
def test_gadget_digits_match_the_reference(libs, Bg_bit, l):
    rng = np.random.default_rng(2)
    x = rnd(rng, N)
    x[:6] = [0, 1, 2 ** 63, 2 ** 64 - 1, (1 << (64 - Bg_bit)) - 1, 1 << (64 - l * Bg_bit - 1)]   # digit boundaries
    for i in range(l):
        def run(lib):
            out, inp = lib.poly(np.zeros(N, dtype=np.uint64)), lib.poly(x)
            lib.l.polynomial_decompose_i(out, inp, Bg_bit, l, i)
            return get_poly(out)
        both(libs, run)

    def run_all(lib):
        outs = (C.POINTER(Poly) * l)(*[lib.poly(np.zeros(N, dtype=np.uint64)) for _ in range(l)])
        lib.l.polynomial_decompose(outs, lib.poly(x), Bg_bit, l)
        return np.stack([get_poly(outs[i]) for i in range(l)])
    both(libs, run_all)
    c = rnd(rng, 2, N)

    def run_trlwe(lib):
        outs = (C.POINTER(Poly) * (2 * l))(*[lib.poly(np.zeros(N, dtype=np.uint64)) for _ in range(2 * l)])
        lib.l.trlwe_decompose(outs, lib.trlwe(c), Bg_bit, l)
        return np.stack([get_poly(outs[i]) for i in range(2 * l)])
    both(libs, run_trlwe)


def test_polynomial_arithmetic_matches_the_reference(libs):
    rng = np.random.default_rng(3)
    x, y, z = rnd(rng, N), rnd(rng, N), rnd(rng, N)
    for name, nargs in (("polynomial_add_torus_polynomials", 3), ("polynomial_sub_torus_polynomials", 3), ("polynomial_addto_torus_polynomial", 2),
                        ("polynomial_subto_torus_polynomial", 2), ("polynomial_negate_torus_polynomial", 2), ("polynomial_copy_torus_polynomial", 2),
                        ("polynomial_naive_mul_torus", 3), ("polynomial_naive_mul_addto_torus", 3)):
        def run(lib):
            out = lib.poly(z)
            args = [out, lib.poly(x)] + ([lib.poly(y)] if nargs == 3 else [])
            getattr(lib.l, name)(*args)
            return get_poly(out)
        both(libs, run)
    for log_scale in (1, 3, 11, 63):
        def run(lib):
            out = lib.poly(z)
            lib.l.polynomial_torus_scale(out, lib.poly(x), log_scale)
            return get_poly(out)
        both(libs, run)
    for scale in (0, 1, 3, 2 ** 40 + 7, 2 ** 64 - 1):
        def run(lib):
            out = lib.poly(z)
            lib.l.polynomial_torus_scale2(out, lib.poly(x), C.c_uint64(scale))
            return get_poly(out)
        both(libs, run)
        c = rnd(rng, 2, N)

        def run_t(lib):
            out = lib.trlwe(np.zeros((2, N), dtype=np.uint64))
            lib.l.trlwe_scale(out, lib.trlwe(c), C.c_uint64(scale))
            return get_trlwe(out)
        both(libs, run_t)

    def run_zero(lib):
        p = lib.poly(x)
        lib.l.polynomial_zero_torus_polynomial(p)
        return get_poly(p)
    assert not both(libs, run_zero).any()
    # the schoolbook product is the ring product: (X^a) * y = rotation of y
    mono = np.zeros(N, dtype=np.uint64)
    mono[5] = 1

    def run_mono(lib):
        out = lib.poly(z)
        lib.l.polynomial_naive_mul_torus(out, lib.poly(y), lib.poly(mono))
        rot = lib.poly(z)
        lib.l.torus_polynomial_mul_by_xai(rot, lib.poly(y), 5)
        assert (get_poly(out) == get_poly(rot)).all()
        return get_poly(out)
    both(libs, run_mono)


def test_lut_packing_matches_the_reference(libs):
    table = np.arange(1, 17, dtype=np.uint64)
    for in_prec, out_prec in ((2, 3), (4, 5), (3, 8)):
        def run(lib):
            out = lib.trlwe(np.ones((2, N), dtype=np.uint64))
            lib.l.trlwe_LUT_packing(out, table.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint64(in_prec), C.c_uint64(out_prec))
            return get_trlwe(out)
        got = both(libs, run)
        assert not got[0].any() and got[1][0] == np.uint64(1) << np.uint64(64 - out_prec)


@pytest.mark.parametrize("Bg_bit,l", [(8, 2), (9, 4)])
def test_trgsw_torus_helpers_match_the_reference(libs, Bg_bit, l):
    rng = np.random.default_rng(4)
    g1, g2, g3 = rnd(rng, 2 * l, 2, N), rnd(rng, 2 * l, 2, N), rnd(rng, 2 * l, 2, N)
    for name in ("trgsw_add", "trgsw_sub"):
        def run(lib):
            out = lib.trgsw(g3, Bg_bit)
            getattr(lib.l, name)(out, lib.trgsw(g1, Bg_bit), lib.trgsw(g2, Bg_bit))
            return get_trgsw(out)
        both(libs, run)
    for name in ("trgsw_addto", "trgsw_copy"):
        def run(lib):
            out = lib.trgsw(g3, Bg_bit)
            getattr(lib.l, name)(out, lib.trgsw(g1, Bg_bit))
            return get_trgsw(out)
        both(libs, run)
    for name in ("trgsw_mul_by_xai", "trgsw_mul_by_xai_addto", "trgsw_mul_by_xai_minus_1"):
        for a in (0, 7, N + 2, 2 * N - 1):
            def run(lib):
                out = lib.trgsw(g3, Bg_bit)
                getattr(lib.l, name)(out, lib.trgsw(g1, Bg_bit), C.c_int(a))
                return get_trgsw(out)
            both(libs, run)
    for m in (0, 1, 5, 2 ** 64 - 1):
        def run(lib):
            out = lib.trgsw(g3, Bg_bit)
            lib.l.trgsw_noiseless_trivial_sample(out, C.c_uint64(m), l, Bg_bit, 1, N)
            return get_trgsw(out)
        got = both(libs, run)
        assert got[0, 0, 0] == np.uint64((m << (64 - Bg_bit)) % 2 ** 64) and not got[0, 1].any()


def test_exact_128_bit_product_matches_the_reference(libs):
    """polynomial_full_mul_with_scale (src/polynomial.c:428-437): the reference multiplies by Karatsuba in 128-bit arithmetic, this library by the
    schoolbook product in the same ring -- identical words, for full-range 64-bit coefficients and every scale the tensor product uses.  Coefficient N - 1
    is the exception: the reference subtracts product term 2N - 1, one element PAST its (2N - 1)-entry buffer (src/fft/karatsuba.c:55,100: whatever the heap
    holds there); that term of a product of two degree-(N - 1) polynomials is 0, which is what this library subtracts and what exact arithmetic gives."""
    rng = np.random.default_rng(5)
    x, y = rnd(rng, N), rnd(rng, N)
    xi, yi = [int(v) for v in x], [int(v) for v in y]
    for scale in (0, 1, 32, 60, 63):
        def run(lib):
            out = lib.poly(np.zeros(N, dtype=np.uint64))
            lib.l.polynomial_full_mul_with_scale(out, lib.poly(x), lib.poly(y), 64, scale)
            return get_poly(out)
        got, want = run(libs[0]), run(libs[1])
        assert (got[:N - 1] == want[:N - 1]).all()
        top = sum(xi[j] * yi[N - 1 - j] for j in range(N)) % 2 ** 128          # product term N - 1; term 2N - 1 is zero
        assert int(got[N - 1]) == (top >> scale) % 2 ** 64


def test_oracle_lut_packing_keyswitch_is_the_references(libs, oracle):
    """Pins oracle.trlwe_lut_packing_keyswitch (the checker of the device kernel, tests/test_gpu_parity.py::test_lut_packing_keyswitch_bit_exact) to the
    reference's trlwe_packing_keyswitch (src/keyswitch.c:346-366): a key made by the REFERENCE's trlwe_new_packing_KS_key, its rows expanded through the
    reference's own trlwe_compressed_subto (0 - row), the same LWE inputs through both: identical TRLWE samples, bit for bit."""
    ref = libs[1]
    L = ref.l
    Nn, n, t, bb, tb = 256, 6, 3, 2, 4
    cands = (1 << bb) - 1

    class TlweKey(C.Structure):
        _fields_ = [("s", C.POINTER(C.c_uint64)), ("n", C.c_int), ("sigma", C.c_double)]

    class Tlwe(C.Structure):
        _fields_ = [("a", C.POINTER(C.c_uint64)), ("b", C.c_uint64), ("n", C.c_int)]

    class PackKey(C.Structure):
        _fields_ = [("s", C.POINTER(C.POINTER(C.POINTER(C.POINTER(C.POINTER(Trlwe)))))), ("base_bit", C.c_int), ("t", C.c_int), ("torus_base", C.c_int), ("n", C.c_int)]

    L.tlwe_new_binary_key.restype = C.POINTER(TlweKey)
    L.tlwe_new_binary_key.argtypes = [C.c_int, C.c_double]
    L.trlwe_new_binary_key.restype = C.c_void_p
    L.trlwe_new_binary_key.argtypes = [C.c_int, C.c_int, C.c_double]
    L.trlwe_new_packing_KS_key.restype = C.POINTER(PackKey)
    L.trlwe_new_packing_KS_key.argtypes = [C.c_void_p, C.POINTER(TlweKey), C.c_int, C.c_int, C.c_int]
    L.tlwe_new_sample.restype = C.POINTER(Tlwe)
    L.tlwe_new_sample.argtypes = [C.c_uint64, C.POINTER(TlweKey)]
    L.trlwe_packing_keyswitch.argtypes = [C.POINTER(Trlwe), C.POINTER(C.POINTER(Tlwe)), C.POINTER(PackKey)]
    L.trlwe_compressed_subto.argtypes = [C.POINTER(Trlwe), C.POINTER(Trlwe)]
    in_key = L.tlwe_new_binary_key(n, 1e-9)
    out_key = L.trlwe_new_binary_key(Nn, 1, 1e-12)
    key = L.trlwe_new_packing_KS_key(out_key, in_key, t, bb, tb)
    rows = np.empty((n, tb, t, cands, 2, Nn), dtype=np.uint64)
    for i in range(n):
        for e in range(tb):
            for j in range(t):
                for v in range(cands):
                    acc = ref.trlwe(np.zeros((2, Nn), dtype=np.uint64))
                    L.trlwe_compressed_subto(acc, key.contents.s[i][e][j][v])
                    rows[i, e, j, v] = (np.uint64(0) - get_trlwe(acc))
    rng = np.random.default_rng(6)
    for _ in range(3):
        msgs = rnd(rng, tb)
        cts = (C.POINTER(Tlwe) * tb)(*[L.tlwe_new_sample(C.c_uint64(int(m)), in_key) for m in msgs])
        flat = np.stack([np.concatenate([np.ctypeslib.as_array(c.contents.a, (n,)).copy(), np.array([c.contents.b], dtype=np.uint64)]) for c in cts])
        out = ref.trlwe(np.zeros((2, Nn), dtype=np.uint64))
        L.trlwe_packing_keyswitch(out, cts, key)
        assert (get_trlwe(out) == oracle.trlwe_lut_packing_keyswitch(flat, rows, bb)).all()
