"""ctypes binding of oracle/_ref/libmosfhet_ref_*.so -- the REAL reference library built from
/root/reference by oracle/ref/Makefile plus our flat-buffer shim (oracle/ref/ref_harness.c).

TEST INFRASTRUCTURE ONLY.  Used to pin the oracle (tests/test_oracle_vs_reference.py), to
generate tests/golden (tests/golden/make_golden.py) and as bench.py's cpu_baseline
(kind = "reference").  The .so files are git-ignored build products; they travel to the GPU box
with the snapshot, /root/reference itself does not.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")
U64P = C.POINTER(C.c_uint64)


def cpu_flags():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def avx512_ok():
    need = {"avx512f", "avx512dq", "avx512bw", "avx512vl", "avx512cd", "vaes", "aes", "rdrand", "fma", "pclmulqdq"}
    return need <= cpu_flags()


def build():
    """(Re)build oracle/_ref from /root/reference when that tree is present (this container only)."""
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "ref")])
        if os.path.exists(os.path.join(_HERE, "..", "mosfhet_amd", "libmosfhet_hip.so")):
            # the reference's vertical_packing.c, unchanged, against include/mosfhet.h + the product library (drop-in check, run under -m gpu)
            subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "ref"), "app"])
        return True
    return False


def available(backend):
    return os.path.exists(os.path.join(REF_DIR, "libmosfhet_ref_%s.so" % backend)) and (
        backend != "avx512" or avx512_ok())


def _u(a):
    assert a.dtype == np.uint64 and a.flags.c_contiguous
    return a.ctypes.data_as(U64P)


class Ref:
    """One loaded reference build ("avx512" = SPQLIOS AVX-512, "ffnt" = portable pure C)."""

    def __init__(self, backend):
        path = os.path.join(REF_DIR, "libmosfhet_ref_%s.so" % backend)
        # RTLD_LOCAL: the reference exports the mosfhet.h symbol names; keep them private
        self.l = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
        self.backend = backend
        L = self.l
        L.ref_torus2int.restype = C.c_uint64
        L.ref_torus2int.argtypes = [C.c_uint64, C.c_int]
        L.ref_double2torus.restype = C.c_uint64
        L.ref_double2torus.argtypes = [C.c_double]
        L.ref_backend.restype = C.c_char_p
        L.ref_tlwe_phase.restype = C.c_uint64
        L.ref_bk_new.restype = C.c_void_p
        L.ref_ksk_new.restype = C.c_void_p
        L.ref_bench_programmable_bootstrap.restype = C.c_double

    def init(self, N):
        self.l.ref_init(N)

    def torus2int(self, x, ls):
        return self.l.ref_torus2int(C.c_uint64(int(x)), ls)

    def double2torus(self, x):
        return self.l.ref_double2torus(C.c_double(x))

    def poly_decompose_i(self, p, Bg_bit, l, i):
        out = np.empty_like(p)
        self.l.ref_poly_decompose_i(_u(out), _u(p), C.c_int(p.size), Bg_bit, l, i)
        return out

    def poly_decompose(self, p, Bg_bit, l):
        out = np.empty((l, p.size), dtype=np.uint64)
        self.l.ref_poly_decompose(_u(out), _u(p), C.c_int(p.size), Bg_bit, l)
        return out

    def poly_mul_by_xai(self, p, a, which=0, acc=None):
        out = acc.copy() if acc is not None else np.zeros_like(p)
        self.l.ref_poly_mul_by_xai(_u(out), _u(p), C.c_int(p.size), C.c_int(a), which)
        return out

    def poly_permute(self, p, gen):
        out = np.empty_like(p)
        self.l.ref_poly_permute(_u(out), _u(p), C.c_int(p.size), C.c_uint64(gen))
        return out

    def poly_naive_mul(self, a, b):
        out = np.empty_like(a)
        self.l.ref_poly_naive_mul(_u(out), _u(a), _u(b), C.c_int(a.size))
        return out

    def poly_mul_fft(self, a, b):
        out = np.empty_like(a)
        self.l.ref_poly_mul_fft(_u(out), _u(a), _u(b), C.c_int(a.size))
        return out

    def poly_dft_roundtrip(self, a):
        out = np.empty_like(a)
        self.l.ref_poly_dft_roundtrip(_u(out), _u(a), C.c_int(a.size))
        return out

    def trlwe_extract_tlwe(self, c, idx):
        k1, N = c.shape
        out = np.empty((k1 - 1) * N + 1, dtype=np.uint64)
        self.l.ref_trlwe_extract_tlwe(_u(out), _u(c), k1 - 1, N, idx)
        return out

    def trlwe_torus_packing(self, lut, k, N):
        lut = np.ascontiguousarray(lut, dtype=np.uint64)
        out = np.empty((k + 1, N), dtype=np.uint64)
        self.l.ref_trlwe_torus_packing(_u(out), _u(lut), k, N, C.c_int(lut.size))
        return out

    def tlwe_phase(self, c, s):
        return self.l.ref_tlwe_phase(_u(c), _u(s), C.c_int(s.size))

    def external_product(self, c, g, l, Bg_bit):
        k1, N = c.shape
        out = np.empty_like(c)
        self.l.ref_external_product(_u(out), _u(c), _u(g), k1 - 1, N, l, Bg_bit)
        return out

    def bk_new(self, bk, k, l, Bg_bit):
        n, _, _, N = bk.shape
        return C.c_void_p(self.l.ref_bk_new(_u(bk), n, k, N, l, Bg_bit))

    def bk_free(self, h):
        self.l.ref_bk_free(h)

    def blind_rotate(self, acc, a, h):
        out = acc.copy()
        self.l.ref_blind_rotate(_u(out), _u(a), h)
        return out

    def functional_bootstrap_wo_extract(self, tv, c, h, torus_base):
        out = np.empty_like(tv)
        self.l.ref_functional_bootstrap_wo_extract(_u(out), _u(tv), _u(c), h, torus_base)
        return out

    def functional_bootstrap(self, tv, c, h, torus_base):
        k1, N = tv.shape
        out = np.empty((k1 - 1) * N + 1, dtype=np.uint64)
        self.l.ref_functional_bootstrap(_u(out), _u(tv), _u(c), h, torus_base)
        return out

    def programmable_bootstrap(self, tv, c, h, precision, kappa, theta):
        k1, N = tv.shape
        out = np.empty((k1 - 1) * N + 1, dtype=np.uint64)
        self.l.ref_programmable_bootstrap(_u(out), _u(tv), _u(c), h, precision, kappa, theta)
        return out

    def ksk_new(self, ksk, base_bit):
        n_in, t, _, row = ksk.shape
        return C.c_void_p(self.l.ref_ksk_new(_u(ksk), n_in, row - 1, t, base_bit))

    def ksk_free(self, h):
        self.l.ref_ksk_free(h)

    def tlwe_keyswitch(self, c, h, n_out):
        out = np.empty(n_out + 1, dtype=np.uint64)
        self.l.ref_tlwe_keyswitch(_u(out), _u(c), h, n_out)
        return out

    def full_domain_functional_bootstrap(self, tv, c, bkh, kskh, precision):
        k1, N = tv.shape
        out = np.empty((k1 - 1) * N + 1, dtype=np.uint64)
        self.l.ref_full_domain_functional_bootstrap(_u(out), _u(tv), _u(c), bkh, kskh, precision)
        return out

    def multivalue_bootstrap_CLOT21(self, tv, c, bkh, torus_base, n_luts):
        k1, N = tv.shape
        out = np.empty((n_luts, (k1 - 1) * N + 1), dtype=np.uint64)
        self.l.ref_multivalue_bootstrap_CLOT21(_u(out), _u(tv), _u(c), bkh, torus_base, n_luts)
        return out

    def trlwe_torus_packing_many_LUT(self, lut, k, N, lut_size, n_luts):
        lut = np.ascontiguousarray(lut, dtype=np.uint64)
        out = np.empty((k + 1, N), dtype=np.uint64)
        self.l.ref_trlwe_torus_packing_many_LUT(_u(out), _u(lut), k, N, lut_size, n_luts)
        return out

    def trlwe_keyswitch(self, c, ks, base_bit):
        t, _, N = ks.shape
        out = np.empty_like(c)
        self.l.ref_trlwe_keyswitch(_u(out), _u(c), _u(ks), N, t, base_bit)
        return out

    def trlwe_eval_automorphism(self, c, gen, ks, base_bit):
        t, _, N = ks.shape
        out = np.empty_like(c)
        self.l.ref_trlwe_eval_automorphism(_u(out), _u(c), C.c_uint64(gen), _u(ks), N, t, base_bit)
        return out

    def inverse_mod_2N(self, x, N):
        self.l.ref_inverse_mod_2N.restype = C.c_uint32
        return self.l.ref_inverse_mod_2N(C.c_uint32(x), N)

    def bk_ga_new(self, bk, ak, l, Bg_bit):
        n, _, _, N = bk.shape
        self.l.ref_bk_ga_new.restype = C.c_void_p
        return C.c_void_p(self.l.ref_bk_ga_new(_u(bk), _u(ak), n, N, l, Bg_bit))

    def bk_ga_free(self, h):
        self.l.ref_bk_ga_free(h)

    def functional_bootstrap_ga(self, tv, c, h, torus_base, extract=True):
        k1, N = tv.shape
        out = np.empty(N + 1, dtype=np.uint64) if extract else np.empty_like(tv)
        self.l.ref_functional_bootstrap_ga(_u(out), _u(tv), _u(c), h, torus_base, int(extract))
        return out

    def trlwe_packing1_keyswitch(self, c, ksk, base_bit):
        n, t, _, _, N = ksk.shape
        out = np.empty((2, N), dtype=np.uint64)
        self.l.ref_trlwe_packing1_keyswitch(_u(out), _u(c), _u(ksk), n, N, t, base_bit)
        return out

    def trlwe_priv_keyswitch_2(self, c, ks0, ks1, base_bit):
        t, _, N = ks0.shape
        out = np.empty_like(c)
        self.l.ref_trlwe_priv_keyswitch_2(_u(out), _u(c), _u(ks0), _u(ks1), N, t, base_bit)
        return out

    # ---- callers either side of the bootstrap; table-lookup keys are made by the reference's own key generation ----
    def generic_key_new(self, kind, s_out, s_in, t, base_bit, sigma):
        """kind 0: trlwe_new_packing1_KS_key, 1: trlwe_new_priv_SK_KS_key_N2.  Returns (handle, flat rows)."""
        self.l.ref_generic_key_new.restype = C.c_void_p
        N, n = s_out.size, s_in.size
        h = C.c_void_p(self.l.ref_generic_key_new(kind, _u(s_out), N, _u(s_in), n, t, base_bit, C.c_double(sigma)))
        flat = np.empty((n + kind, t, (1 << base_bit) - 1, 2, N), dtype=np.uint64)
        self.l.ref_generic_key_export(h, _u(flat), N)
        return h, flat

    def generic_key_free(self, h):
        self.l.ref_generic_key_free(h)

    def generic_keyswitch(self, kind, c, h, N):
        out = np.empty((2, N), dtype=np.uint64)
        self.l.ref_generic_keyswitch(kind, _u(out), _u(c), h, N)
        return out

    def public_mux(self, p0, p1, sel, Bg_bit):
        l, _, N = sel.shape
        out = np.empty((2, N), dtype=np.uint64)
        self.l.ref_public_mux(_u(out), _u(p0), _u(p1), _u(sel), N, l, Bg_bit)
        return out

    def full_domain_functional_bootstrap_KS21(self, tv, c, bkh, kskh, N, torus_base, variant=0):
        out = np.empty(N + 1, dtype=np.uint64)
        self.l.ref_full_domain_functional_bootstrap_KS21(_u(out), _u(tv), _u(c), bkh, kskh, torus_base, variant)
        return out

    def multivalue_bootstrap_phase1(self, c, bkh, N, torus_base):
        out = np.empty((torus_base + 1, 2, N), dtype=np.uint64)
        self.l.ref_multivalue_bootstrap_phase1(_u(out), _u(c), bkh, torus_base)
        return out

    def multivalue_bootstrap_phase2(self, lut_in, rotated, torus_base, log_torus_base):
        N = rotated.shape[-1]
        out = np.empty(N + 1, dtype=np.uint64)
        li = np.ascontiguousarray(lut_in, dtype=np.int32)
        self.l.ref_multivalue_bootstrap_phase2(_u(out), li.ctypes.data_as(C.POINTER(C.c_int)), _u(rotated), N, torus_base, log_torus_base)
        return out

    def circuit_bootstrap(self, c, bkh, l, N, kskah, kskbh, variant, kska_flat=None, bba=0):
        out = np.empty((2 * l, 2, N), dtype=np.uint64)
        ta = 0 if kska_flat is None else kska_flat.shape[1]
        self.l.ref_circuit_bootstrap(_u(out), _u(c), bkh, kskah, None if kska_flat is None else _u(kska_flat), ta, bba, kskbh, variant)
        return out

    def functional_bootstrap_trgsw(self, tv, c, bkh, l, torus_base):
        N = tv.shape[-1]
        acc = np.empty((2 * l, 2, N), dtype=np.uint64)
        out = np.empty(N + 1, dtype=np.uint64)
        self.l.ref_functional_bootstrap_trgsw(_u(acc), _u(out), _u(tv), _u(c), bkh, torus_base)
        return acc, out

    def trlwe_tensor_prod_FFT(self, c1, c2, precision, rl, base_bit):
        t, _, N = rl.shape
        out = np.empty_like(c1)
        self.l.ref_trlwe_tensor_prod_FFT(_u(out), _u(c1), _u(c2), precision, _u(rl), N, t, base_bit)
        return out

    def tlwe_mul(self, c1, c2, precision, kskh, rl, base_bit):
        t, _, N = rl.shape
        out = np.empty(N + 1, dtype=np.uint64)
        self.l.ref_tlwe_mul(_u(out), _u(c1), _u(c2), precision, kskh, _u(rl), N, t, base_bit)
        return out

    def full_domain_functional_bootstrap_CLOT21(self, tv, c, bkh, kskh, rl, base_bit, precision, variant=0):
        t, _, N = rl.shape
        out = np.empty(N + 1, dtype=np.uint64)
        self.l.ref_full_domain_functional_bootstrap_CLOT21(_u(out), _u(tv), _u(c), bkh, kskh, _u(rl), t, base_bit, precision, variant)
        return out

    def bk_unfolded_new(self, su, l, Bg_bit, unfolding):
        count, _, _, N = su.shape
        self.l.ref_bk_unfolded_new.restype = C.c_void_p
        return C.c_void_p(self.l.ref_bk_unfolded_new(_u(su), count * unfolding >> unfolding, N, l, Bg_bit, unfolding))

    def multivalue_bootstrap_UBR(self, tvs, c, bkh, torus_base):
        n_tv, _, N = tvs.shape
        out = np.empty((n_tv, N + 1), dtype=np.uint64)
        self.l.ref_multivalue_bootstrap_UBR(_u(out), _u(tvs), n_tv, _u(c), bkh, torus_base)
        return out

    def trlwe_mv_extract(self, c, mode, amount, acc=None):
        N = c.shape[1]
        out = np.zeros((amount if mode == 0 else 1, N + 1), dtype=np.uint64) if acc is None else acc.copy().reshape(1, N + 1)
        self.l.ref_trlwe_mv_extract(_u(out), _u(c), N, mode, amount)
        return out if mode == 0 else out[0]

    def save_host_objects(self, path, lwe_s, lwe_sigma, rlwe_s, rlwe_sigma, l, Bg_bit, tlwe_ct, trlwe_ct):
        """tlwe_save_key, trlwe_save_key, trgsw_save_key, tlwe_save_sample, trlwe_save_sample into one file, in that order."""
        k, N = rlwe_s.shape
        self.l.ref_save_host_objects.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_void_p]
        rc = self.l.ref_save_host_objects(path.encode(), _u(lwe_s), len(lwe_s), lwe_sigma, _u(rlwe_s), k, N, rlwe_sigma, l, Bg_bit, _u(tlwe_ct), _u(trlwe_ct))
        assert rc == 0

    def ksk_save(self, path, h):
        self.l.ref_ksk_save.argtypes = [C.c_char_p, C.c_void_p]
        assert self.l.ref_ksk_save(path.encode(), h) == 0

    def ksk_load(self, path):
        self.l.ref_ksk_load.argtypes = [C.c_char_p]
        self.l.ref_ksk_load.restype = C.c_void_p
        h = self.l.ref_ksk_load(path.encode())
        assert h
        return C.c_void_p(h)

    def bench_programmable_bootstrap(self, tv, c, h, precision, reps):
        """Seconds for `reps` programmable_bootstrap calls on the calling thread (GIL released)."""
        return self.l.ref_bench_programmable_bootstrap(_u(tv), _u(c), h, precision, reps)


_cache = {}


def get(backend):
    if backend not in _cache:
        _cache[backend] = Ref(backend)
    return _cache[backend]
