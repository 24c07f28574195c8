"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the mosfhet_amd package.
All arrays are numpy uint64 / float64, C-contiguous, in the flat layouts of
mosfhet_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

U64P = C.POINTER(C.c_uint64)
F64P = C.POINTER(C.c_double)


def build(force=False):
    """Compile liboracle.so with gcc (building the checker is not using it)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("oracle_int.c", "oracle_fft.c", "oracle_tfhe.c", "oracle_ext.c", "mosfhet_oracle.h")
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_fft_plan_new.restype = C.c_void_p
        _lib.orc_fft_plan_new.argtypes = [C.c_int]
        _lib.orc_fft_plan_free.argtypes = [C.c_void_p]
        _lib.orc_torus2int.restype = C.c_uint64
        _lib.orc_torus2int.argtypes = [C.c_uint64, C.c_int]
        _lib.orc_double2torus.restype = C.c_uint64
        _lib.orc_double2torus.argtypes = [C.c_double]
        _lib.orc_tlwe_phase.restype = C.c_uint64
        _lib.orc_rng_next.restype = C.c_uint64
    return _lib


def _u(a):
    assert a.dtype == np.uint64 and a.flags.c_contiguous, (a.dtype, a.flags)
    return a.ctypes.data_as(U64P)


def _d(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(F64P)


def u64(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


class Plan:
    """Twiddle plan for ring degree N (orc_fft_plan)."""

    def __init__(self, N):
        self.N = N
        self.h = C.c_void_p(lib().orc_fft_plan_new(N))

    def __del__(self):
        try:
            lib().orc_fft_plan_free(self.h)
        except Exception:
            pass

    def twiddles(self):
        out = np.zeros(2 * (self.N // 2 - 1), dtype=np.float64)
        lib().orc_fft_make_twiddles(_d(out), C.c_int(self.N))
        return out


_plans = {}


def plan(N):
    if N not in _plans:
        _plans[N] = Plan(N)
    return _plans[N]


class Rng:
    def __init__(self, seed):
        self.s = C.c_uint64(seed)

    def ref(self):
        return C.byref(self.s)

    def next(self):
        return lib().orc_rng_next(self.ref())

    def words(self, n):
        return np.array([self.next() for _ in range(n)], dtype=np.uint64)


# ---------------- scalars ----------------
def torus2int(x, log_scale):
    return lib().orc_torus2int(C.c_uint64(int(x)), log_scale)


def double2torus(x):
    return lib().orc_double2torus(C.c_double(x))


# ---------------- integer polynomial ops ----------------
def poly_decompose_i(p, Bg_bit, l, i):
    out = np.empty_like(p)
    lib().orc_poly_decompose_i(_u(out), _u(p), C.c_int(p.size), Bg_bit, l, i)
    return out


def poly_decompose(p, Bg_bit, l):
    out = np.empty((l, p.size), dtype=np.uint64)
    lib().orc_poly_decompose(_u(out), _u(p), C.c_int(p.size), Bg_bit, l)
    return out


def poly_mul_by_xai(p, a):
    out = np.empty_like(p)
    lib().orc_poly_mul_by_xai(_u(out), _u(p), C.c_int(p.size), C.c_int(a))
    return out


def poly_mul_by_xai_addto(acc, p, a):
    out = acc.copy()
    lib().orc_poly_mul_by_xai_addto(_u(out), _u(p), C.c_int(p.size), C.c_int(a))
    return out


def poly_mul_by_xai_minus_1(p, a):
    out = np.empty_like(p)
    lib().orc_poly_mul_by_xai_minus_1(_u(out), _u(p), C.c_int(p.size), C.c_int(a))
    return out


def poly_naive_mul(a, b):
    out = np.empty_like(a)
    lib().orc_poly_naive_mul(_u(out), _u(a), _u(b), C.c_int(a.size))
    return out


def poly_permute(p, gen):
    out = np.empty_like(p)
    lib().orc_poly_permute(_u(out), _u(p), C.c_int(p.size), C.c_uint64(gen))
    return out


def trlwe_extract_tlwe(c, idx):
    k1, N = c.shape
    out = np.empty((k1 - 1) * N + 1, dtype=np.uint64)
    lib().orc_trlwe_extract_tlwe(_u(out), _u(c), k1 - 1, N, idx)
    return out


def trlwe_torus_packing(lut, k, N):
    out = np.empty((k + 1, N), dtype=np.uint64)
    lib().orc_trlwe_torus_packing(_u(out), _u(u64(lut)), k, N, len(lut))
    return out


def tlwe_keyswitch(c, ksk, n_out, t, base_bit):
    n_in = c.size - 1
    out = np.empty(n_out + 1, dtype=np.uint64)
    lib().orc_tlwe_keyswitch(_u(out), _u(c), _u(ksk), n_in, n_out, t, base_bit)
    return out


def tlwe_phase(c, s):
    return lib().orc_tlwe_phase(_u(c), _u(s), C.c_int(s.size))


def trlwe_phase(c, s):
    k1, N = c.shape
    out = np.empty(N, dtype=np.uint64)
    lib().orc_trlwe_phase(_u(out), _u(c), _u(s), k1 - 1, N)
    return out


def pbs_preprocess(c, N, kappa, theta):
    out = np.empty_like(c)
    lib().orc_pbs_preprocess(_u(out), _u(c), C.c_int(c.size - 1), N, kappa, theta)
    return out


# ---------------- transform ----------------
def torus_to_dft(p):
    out = np.empty(p.size, dtype=np.float64)
    lib().orc_torus_to_dft(plan(p.size).h, _d(out), _u(p))
    return out


def dft_to_torus(f):
    out = np.empty(f.size, dtype=np.uint64)
    lib().orc_dft_to_torus(plan(f.size).h, _u(out), _d(f))
    return out


def dft_mul(a, b):
    out = np.empty_like(a)
    lib().orc_dft_mul(_d(out), _d(a), _d(b), C.c_int(a.size))
    return out


def dft_mul_addto(acc, a, b):
    out = acc.copy()
    lib().orc_dft_mul_addto(_d(out), _d(a), _d(b), C.c_int(a.size))
    return out


def poly_mul_fft(a, b):
    out = np.empty_like(a)
    lib().orc_poly_mul_fft(plan(a.size).h, _u(out), _u(a), _u(b))
    return out


# ---------------- TRGSW / bootstrap ----------------
def trgsw_to_dft(g, k, l):
    N = g.shape[-1]
    out = np.empty(g.shape, dtype=np.float64)
    lib().orc_trgsw_to_dft(plan(N).h, _d(out), _u(g), k, l)
    return out


def bk_to_dft(bk, k, l):
    """bk: u64[n][(k+1)l][k+1][N] -> float64 same shape (oracle slot order)."""
    out = np.empty(bk.shape, dtype=np.float64)
    for i in range(bk.shape[0]):
        out[i] = trgsw_to_dft(bk[i], k, l)
    return out


class product_order:
    """with product_order("by_component"): every external product of the oracle (and so every composition) sums per input component first -- the order of a
    bootstrap split over one workgroup per accumulator component (oracle_tfhe.c: orc_set_product_order); "reference" = one chain over all rows (default)."""
    ORDERS = {"reference": 0, "by_component": 1}

    def __init__(self, order):
        self.order = self.ORDERS[order]

    def __enter__(self):
        self.old = lib().orc_get_product_order()
        lib().orc_set_product_order(self.order)
        return self

    def __exit__(self, *exc):
        lib().orc_set_product_order(self.old)
        return False


def external_product(c, g_dft, l, Bg_bit):
    k1, N = c.shape
    out = np.empty_like(c)
    lib().orc_external_product(plan(N).h, _u(out), _u(c), _d(g_dft), k1 - 1, l, Bg_bit)
    return out


def blind_rotate(acc, a, bk_dft, l, Bg_bit):
    k1, N = acc.shape
    out = acc.copy()
    lib().orc_blind_rotate(plan(N).h, _u(out), _u(a), _d(bk_dft), C.c_int(a.size), k1 - 1, l, Bg_bit)
    return out


def functional_bootstrap_wo_extract(tv, c, bk_dft, l, Bg_bit, torus_base):
    k1, N = tv.shape
    out = np.empty_like(tv)
    lib().orc_functional_bootstrap_wo_extract(plan(N).h, _u(out), _u(tv), _u(c), _d(bk_dft),
                                              C.c_int(c.size - 1), k1 - 1, l, Bg_bit, torus_base)
    return out


def functional_bootstrap(tv, c, bk_dft, l, Bg_bit, torus_base):
    k1, N = tv.shape
    out = np.empty((k1 - 1) * N + 1, dtype=np.uint64)
    lib().orc_functional_bootstrap(plan(N).h, _u(out), _u(tv), _u(c), _d(bk_dft),
                                   C.c_int(c.size - 1), k1 - 1, l, Bg_bit, torus_base)
    return out


def programmable_bootstrap(tv, c, bk_dft, l, Bg_bit, precision, kappa, theta):
    k1, N = tv.shape
    out = np.empty((k1 - 1) * N + 1, dtype=np.uint64)
    lib().orc_programmable_bootstrap(plan(N).h, _u(out), _u(tv), _u(c), _d(bk_dft),
                                     C.c_int(c.size - 1), k1 - 1, l, Bg_bit, precision, kappa, theta)
    return out


def full_domain_functional_bootstrap(tv, c, bk_dft, ksk, l, Bg_bit, t, base_bit, precision):
    k1, N = tv.shape
    out = np.empty((k1 - 1) * N + 1, dtype=np.uint64)
    lib().orc_full_domain_functional_bootstrap(plan(N).h, _u(out), _u(tv), _u(c), _d(bk_dft), _u(ksk), C.c_int(c.size - 1),
                                               k1 - 1, l, Bg_bit, t, base_bit, precision)
    return out


def multivalue_bootstrap_CLOT21(tv, c, bk_dft, l, Bg_bit, torus_base, n_luts):
    k1, N = tv.shape
    out = np.empty((n_luts, (k1 - 1) * N + 1), dtype=np.uint64)
    lib().orc_multivalue_bootstrap_CLOT21(plan(N).h, _u(out), _u(tv), _u(c), _d(bk_dft), C.c_int(c.size - 1), k1 - 1, l, Bg_bit,
                                          torus_base, n_luts)
    return out


def trlwe_torus_packing_many_LUT(lut, k, N, lut_size, n_luts):
    out = np.empty((k + 1, N), dtype=np.uint64)
    lib().orc_trlwe_torus_packing_many_LUT(_u(out), _u(u64(lut)), k, N, lut_size, n_luts)
    return out


# ---------------- GA / TRLWE key switch ----------------
def trlwe_keyswitch(c, ks_dft, t, base_bit):
    out = np.empty_like(c)
    lib().orc_trlwe_keyswitch(plan(c.shape[1]).h, _u(out), _u(c), _d(ks_dft), t, base_bit)
    return out


def trlwe_eval_automorphism(c, gen, ks_dft, t, base_bit):
    out = np.empty_like(c)
    lib().orc_trlwe_eval_automorphism(plan(c.shape[1]).h, _u(out), _u(c), C.c_uint64(gen), _d(ks_dft), t, base_bit)
    return out


def inverse_mod_2N(x, N):
    lib().orc_inverse_mod_2N.restype = C.c_uint32
    return lib().orc_inverse_mod_2N(C.c_uint32(x), N)


def ks_to_dft(ks):
    """TRLWE key-switch rows u64[...][2][N] -> float64 same shape (oracle slot order)."""
    N = ks.shape[-1]
    flat = np.ascontiguousarray(ks.reshape(-1, N))
    out = np.empty(flat.shape, dtype=np.float64)
    for i in range(flat.shape[0]):
        out[i] = torus_to_dft(flat[i])
    return out.reshape(ks.shape)


def blind_rotate_ga(acc, a, bk_dft, ak_dft, l, Bg_bit):
    """src/bootstrap_ga.c:39-60 on a caller-supplied accumulator; a: the n mask words"""
    out = acc.copy()
    lib().orc_blind_rotate_ga(plan(acc.shape[1]).h, _u(out), _u(a), _d(bk_dft), _d(ak_dft), C.c_int(a.size), l, Bg_bit)
    return out


def functional_bootstrap_ga(tv, c, bk_dft, ak_dft, l, Bg_bit, torus_base, extract=True):
    k1, N = tv.shape
    n = c.size - 1
    if extract:
        out = np.empty(N + 1, dtype=np.uint64)
        lib().orc_functional_bootstrap_ga(plan(N).h, _u(out), _u(tv), _u(c), _d(bk_dft), _d(ak_dft), n, l, Bg_bit, torus_base)
    else:
        out = np.empty_like(tv)
        lib().orc_functional_bootstrap_wo_extract_ga(plan(N).h, _u(out), _u(tv), _u(c), _d(bk_dft), _d(ak_dft), n, l, Bg_bit, torus_base)
    return out


def gen_trlwe_ks_key(rng, s_in, s_out, t, base_bit, sigma):
    N = s_in.size
    ks = np.empty((t, 2, N), dtype=np.uint64)
    lib().orc_gen_trlwe_ks_key(rng.ref(), _u(ks), _u(s_in), _u(s_out), N, t, base_bit, C.c_double(sigma))
    return ks


def gen_automorphism_keyset(rng, s, t, base_bit, sigma):
    N = s.size
    ak = np.empty((N, t, 2, N), dtype=np.uint64)
    lib().orc_gen_automorphism_keyset(rng.ref(), _u(ak), _u(s), N, t, base_bit, C.c_double(sigma))
    return ak


def gen_bootstrap_key_ga(rng, lwe_s, rlwe_s, l, Bg_bit, sigma):
    k, N = rlwe_s.shape
    n = lwe_s.size
    bk = np.empty((n, 2 * l, 2, N), dtype=np.uint64)
    lib().orc_gen_bootstrap_key_ga(rng.ref(), _u(bk), _u(lwe_s), n, _u(rlwe_s), N, l, Bg_bit, C.c_double(sigma))
    return bk


# ---------------- circuit bootstrap ----------------
def trlwe_packing1_keyswitch(c, ksk, base_bit):
    n, t, _, _, N = ksk.shape
    out = np.empty((2, N), dtype=np.uint64)
    lib().orc_trlwe_packing1_keyswitch(_u(out), _u(c), _u(ksk), n, N, t, base_bit)
    return out


def trlwe_priv_keyswitch_2(c, ks0_dft, ks1_dft, t, base_bit):
    out = np.empty_like(c)
    lib().orc_trlwe_priv_keyswitch_2(plan(c.shape[1]).h, _u(out), _u(c), _d(ks0_dft), _d(ks1_dft), t, base_bit)
    return out


def circuit_bootstrap_3(c, bk_dft, kska0_dft, kska1_dft, bba, kskb, bbb, l, Bg_bit):
    N = bk_dft.shape[-1]
    ta, tb = kska0_dft.shape[0], kskb.shape[1]
    out = np.empty((2 * l, 2, N), dtype=np.uint64)
    lib().orc_circuit_bootstrap_3(plan(N).h, _u(out), _u(c), _d(bk_dft), _d(kska0_dft), _d(kska1_dft), ta, bba, _u(kskb), tb, bbb,
                                  C.c_int(c.size - 1), l, Bg_bit)
    return out


def gen_packing1_ks_key(rng, s_in, s_out, t, base_bit, sigma):
    n, N = s_in.size, s_out.size
    ksk = np.empty((n, t, (1 << base_bit) - 1, 2, N), dtype=np.uint64)
    lib().orc_gen_packing1_ks_key(rng.ref(), _u(ksk), _u(s_in), n, _u(s_out), N, t, base_bit, C.c_double(sigma))
    return ksk


def gen_lut_packing_ks_key(rng, s_in, s_out, t, base_bit, torus_base, sigma):
    """trlwe_new_packing_KS_key (src/keyswitch.c:214-241): rows [n][torus_base][t][2^base_bit - 1][2][N]"""
    n, N = s_in.size, s_out.size
    ksk = np.empty((n, torus_base, t, (1 << base_bit) - 1, 2, N), dtype=np.uint64)
    lib().orc_gen_lut_packing_ks_key(rng.ref(), _u(ksk), _u(s_in), n, _u(s_out), N, t, base_bit, torus_base, C.c_double(sigma))
    return ksk


def trlwe_lut_packing_keyswitch(cts, ksk, base_bit):
    """trlwe_packing_keyswitch (src/keyswitch.c:346-366): cts [torus_base][n + 1] -> TRLWE [2][N]"""
    n, torus_base, t, _, _, N = ksk.shape
    out = np.empty((2, N), dtype=np.uint64)
    lib().orc_trlwe_lut_packing_keyswitch(_u(out), _u(np.ascontiguousarray(cts)), _u(ksk), n, N, t, base_bit, torus_base)
    return out


def gen_priv_ks_key(rng, s_out, s_in, t, base_bit, sigma):
    N = s_out.size
    ks0, ks1 = np.empty((t, 2, N), dtype=np.uint64), np.empty((t, 2, N), dtype=np.uint64)
    lib().orc_gen_priv_ks_key(rng.ref(), _u(ks0), _u(ks1), _u(s_out), _u(s_in), N, t, base_bit, C.c_double(sigma))
    return ks0, ks1


# ---------------- callers either side of the bootstrap (oracle_ext.c) ----------------
def public_mux(p0, p1, sel_dft, l, Bg_bit):
    N = p0.size
    out = np.empty((2, N), dtype=np.uint64)
    lib().orc_public_mux(plan(N).h, _u(out), _u(p0), _u(p1), _d(sel_dft), l, Bg_bit)
    return out


def full_domain_functional_bootstrap_KS21(tv, c, bk_dft, ksk, base_bit, l, Bg_bit, torus_base, variant=0):
    N = bk_dft.shape[-1]
    out = np.empty(N + 1, dtype=np.uint64)
    lib().orc_full_domain_functional_bootstrap_KS21(plan(N).h, _u(out), _u(tv), _u(c), _d(bk_dft), _u(ksk), C.c_int(c.size - 1), l, Bg_bit,
                                                    C.c_int(ksk.shape[1]), base_bit, torus_base, variant)
    return out


def multivalue_bootstrap_phase1(c, bk_dft, l, Bg_bit, torus_base):
    N = bk_dft.shape[-1]
    out = np.empty((torus_base + 1, 2, N), dtype=np.uint64)
    lib().orc_multivalue_bootstrap_phase1(plan(N).h, _u(out), _u(c), _d(bk_dft), C.c_int(c.size - 1), l, Bg_bit, torus_base)
    return out


def multivalue_bootstrap_phase2(lut_in, rotated, torus_base, log_torus_base):
    N = rotated.shape[-1]
    out = np.empty(N + 1, dtype=np.uint64)
    li = np.ascontiguousarray(lut_in, dtype=np.int32)
    lib().orc_multivalue_bootstrap_phase2(_u(out), li.ctypes.data_as(C.POINTER(C.c_int)), _u(rotated), N, torus_base, log_torus_base)
    return out


def trlwe_mv_extract_tlwe_scaling_addto(acc, c, scale):
    out = acc.copy()
    lib().orc_trlwe_mv_extract_tlwe_scaling_addto(_u(out), _u(c), C.c_int(c.shape[1]), scale)
    return out


def gen_priv_sk_ks_key(rng, s_in, s_out, t, base_bit, sigma):
    n, N = s_in.size, s_out.size
    ksk = np.empty((n + 1, t, (1 << base_bit) - 1, 2, N), dtype=np.uint64)
    lib().orc_gen_priv_sk_ks_key(rng.ref(), _u(ksk), _u(s_in), n, _u(s_out), N, t, base_bit, C.c_double(sigma))
    return ksk


def trlwe_priv_keyswitch(c, ksk, base_bit):
    n1, t, _, _, N = ksk.shape
    out = np.empty((2, N), dtype=np.uint64)
    lib().orc_trlwe_priv_keyswitch(_u(out), _u(c), _u(ksk), n1 - 1, N, t, base_bit)
    return out


def circuit_bootstrap(c, bk_dft, kska, bba, kskb, bbb, l, Bg_bit, variant=0):
    N = bk_dft.shape[-1]
    out = np.empty((2 * l, 2, N), dtype=np.uint64)
    lib().orc_circuit_bootstrap(plan(N).h, _u(out), _u(c), _d(bk_dft), _u(kska), C.c_int(kska.shape[1]), bba, _u(kskb), C.c_int(kskb.shape[1]), bbb,
                                C.c_int(c.size - 1), l, Bg_bit, variant)
    return out


def functional_bootstrap_trgsw_phase1(c, bk_dft, l, Bg_bit, torus_base):
    N = bk_dft.shape[-1]
    out = np.empty((2 * l, 2, N), dtype=np.float64)
    lib().orc_functional_bootstrap_trgsw_phase1(plan(N).h, _d(out), _u(c), _d(bk_dft), C.c_int(c.size - 1), l, Bg_bit, torus_base)
    return out


def functional_bootstrap_trgsw_phase2(g_dft, tv, l, Bg_bit):
    N = tv.shape[-1]
    out = np.empty(N + 1, dtype=np.uint64)
    lib().orc_functional_bootstrap_trgsw_phase2(plan(N).h, _u(out), _d(g_dft), _u(tv), l, Bg_bit)
    return out


def gen_rl_key(rng, s, t, base_bit, sigma):
    N = s.size
    ks = np.empty((t, 2, N), dtype=np.uint64)
    lib().orc_gen_rl_key(rng.ref(), _u(ks), _u(s), N, t, base_bit, C.c_double(sigma))
    return ks


def trlwe_tensor_prod_fft(c1, c2, precision, rl_dft, base_bit):
    N = c1.shape[1]
    out = np.empty_like(c1)
    lib().orc_trlwe_tensor_prod_fft(plan(N).h, _u(out), _u(c1), _u(c2), precision, _d(rl_dft), C.c_int(rl_dft.shape[0]), base_bit)
    return out


def tlwe_mul(c1, c2, precision, ksk, bbk, rl_dft, bbr):
    N = c1.size - 1
    out = np.empty(N + 1, dtype=np.uint64)
    lib().orc_tlwe_mul(plan(N).h, _u(out), _u(c1), _u(c2), precision, _u(ksk), C.c_int(ksk.shape[1]), bbk, _d(rl_dft), C.c_int(rl_dft.shape[0]), bbr)
    return out


def full_domain_functional_bootstrap_CLOT21(tv, c, bk_dft, ksk, bbk, rl_dft, bbr, l, Bg_bit, precision, variant=0):
    N = bk_dft.shape[-1]
    out = np.empty(N + 1, dtype=np.uint64)
    lib().orc_full_domain_functional_bootstrap_CLOT21(plan(N).h, _u(out), _u(tv), _u(c), _d(bk_dft), _u(ksk), C.c_int(ksk.shape[1]), bbk, _d(rl_dft),
                                                      C.c_int(rl_dft.shape[0]), bbr, C.c_int(c.size - 1), l, Bg_bit, precision, variant)
    return out


def gen_bootstrap_key_unfolded(rng, lwe_s, rlwe_s, l, Bg_bit, sigma, unfolding):
    n, N = lwe_s.size, rlwe_s.size
    su = np.empty((n * (1 << unfolding) // unfolding, 2 * l, 2, N), dtype=np.uint64)
    lib().orc_gen_bootstrap_key_unfolded(rng.ref(), _u(su), _u(lwe_s), n, _u(rlwe_s), N, l, Bg_bit, C.c_double(sigma), unfolding)
    return su


def functional_bootstrap_unfolded(tv, c, su, l, Bg_bit, torus_base, unfolding, extract=True):
    N = tv.shape[-1]
    out = np.empty(N + 1, dtype=np.uint64) if extract else np.empty((2, N), dtype=np.uint64)
    lib().orc_functional_bootstrap_unfolded(plan(N).h, _u(out), _u(tv), _u(c), _u(su), C.c_int(c.size - 1), l, Bg_bit, torus_base, unfolding, int(extract))
    return out


def monomial_table(N):
    """W[x] = exp(i pi x / N), x < 2N, as float64 [2N][2]"""
    out = np.empty((2 * N, 2), dtype=np.float64)
    lib().orc_monomial_table(_d(out), N)
    return out


def su_to_dft(su, l):
    """torus-domain unfolded key samples [cnt][2l][2][N] -> their transforms, same shape (natural slot order)"""
    return bk_to_dft(su, 1, l)


def functional_bootstrap_unfolded2_dft(tv, c, su_dft, l, Bg_bit, torus_base, extract=True):
    """functional_bootstrap(_wo_extract) with an unfolding-2 key, the per-group TRGSW assembled in the DFT domain (the GPU kernel's order)"""
    N = tv.shape[-1]
    out = np.empty(N + 1 if extract else (2, N), dtype=np.uint64)
    lib().orc_functional_bootstrap_unfolded2_dft(plan(N).h, _u(out), _u(tv), _u(c), _d(su_dft), C.c_int(c.size - 1), l, Bg_bit, torus_base, int(extract))
    return out


def multivalue_bootstrap_UBR_phase1(c, su, l, Bg_bit, unfolding):
    N = su.shape[-1]
    n = c.size - 1
    out = np.empty((n // unfolding, 2 * l, 2, N), dtype=np.float64)
    lib().orc_multivalue_bootstrap_UBR_phase1(plan(N).h, _d(out), _u(c), _u(su), n, l, Bg_bit, unfolding)
    return out


def multivalue_bootstrap_UBR_phase2(tv, c, sa_dft, l, Bg_bit, unfolding, torus_base):
    N = tv.shape[-1]
    out = np.empty(N + 1, dtype=np.uint64)
    lib().orc_multivalue_bootstrap_UBR_phase2(plan(N).h, _u(out), _u(tv), _u(c), _d(sa_dft), C.c_int(c.size - 1), l, Bg_bit, unfolding, torus_base)
    return out


def trlwe_mv_extract(c, mode, amount, acc=None):
    N = c.shape[1]
    out = np.zeros((amount if mode == 0 else 1, N + 1), dtype=np.uint64) if acc is None else acc.copy().reshape(1, N + 1)
    lib().orc_trlwe_mv_extract(_u(out), _u(c), N, mode, amount)
    return out if mode == 0 else out[0]


# ---------------- deterministic inputs ----------------
def gen_binary_key(rng, n):
    s = np.empty(n, dtype=np.uint64)
    lib().orc_gen_binary_key(rng.ref(), _u(s), n)
    return s


def tlwe_sample(rng, m, s, sigma):
    out = np.empty(s.size + 1, dtype=np.uint64)
    lib().orc_tlwe_sample(rng.ref(), _u(out), C.c_uint64(int(m) & (2**64 - 1)), _u(s), C.c_int(s.size),
                          C.c_double(sigma))
    return out


def trlwe_sample(rng, m, s, sigma):
    k, N = s.shape
    out = np.empty((k + 1, N), dtype=np.uint64)
    mp = _u(m) if m is not None else None
    lib().orc_trlwe_sample(rng.ref(), _u(out), mp, _u(s), k, N, C.c_double(sigma))
    return out


def trgsw_monomial_sample(rng, m, e, s, l, Bg_bit, sigma):
    k, N = s.shape
    out = np.empty(((k + 1) * l, k + 1, N), dtype=np.uint64)
    lib().orc_trgsw_monomial_sample(rng.ref(), _u(out), C.c_int64(m), e, _u(s), k, N, l, Bg_bit,
                                    C.c_double(sigma))
    return out


def gen_bootstrap_key(rng, lwe_s, rlwe_s, l, Bg_bit, sigma):
    k, N = rlwe_s.shape
    n = lwe_s.size
    bk = np.empty((n, (k + 1) * l, k + 1, N), dtype=np.uint64)
    lib().orc_gen_bootstrap_key(rng.ref(), _u(bk), _u(lwe_s), n, _u(rlwe_s), k, N, l, Bg_bit, C.c_double(sigma))
    return bk


def gen_tlwe_ks_key(rng, s_in, s_out, t, base_bit, sigma):
    ksk = np.empty((s_in.size, t, (1 << base_bit) - 1, s_out.size + 1), dtype=np.uint64)
    lib().orc_gen_tlwe_ks_key(rng.ref(), _u(ksk), _u(s_in), C.c_int(s_in.size), _u(s_out), C.c_int(s_out.size),
                              t, base_bit, C.c_double(sigma))
    return ksk


def torus_dist(a, b):
    """|(int64)(a - b)| -- the wrap-aware torus distance."""
    d = (np.asarray(a, dtype=np.uint64) - np.asarray(b, dtype=np.uint64)).astype(np.int64)
    return np.abs(d.astype(np.float64))
