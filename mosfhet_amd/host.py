"""ctypes view of the MOSFHET-compatible host layer (include/mosfhet_compat.h) for tests and bench.py:
seeded key / sample generation in the flat layouts of include/mosfhet_hip.h.  Host-only C code, no GPU needed.
"""
import ctypes as C

import numpy as np

from . import engine as _e


class _TLWEKey(C.Structure):
    _fields_ = [("s", C.POINTER(C.c_uint64)), ("n", C.c_int), ("sigma", C.c_double)]


class _Poly(C.Structure):
    _fields_ = [("coeffs", C.POINTER(C.c_uint64)), ("N", C.c_int)]


class _TRLWEKey(C.Structure):
    _fields_ = [("s", C.POINTER(C.POINTER(_Poly))), ("s_dft", C.c_void_p), ("k", C.c_int), ("sigma", C.c_double)]


def _lib():
    L = _e.lib()
    if not getattr(L, "_host_ready", False):
        L.tlwe_new_binary_key.restype = C.POINTER(_TLWEKey)
        L.tlwe_new_binary_key.argtypes = [C.c_int, C.c_double]
        L.tlwe_alloc_key.restype = C.POINTER(_TLWEKey)
        L.tlwe_alloc_key.argtypes = [C.c_int, C.c_double]
        L.trlwe_new_binary_key.restype = C.POINTER(_TRLWEKey)
        L.trlwe_new_binary_key.argtypes = [C.c_int, C.c_int, C.c_double]
        L.trgsw_new_key.restype = C.c_void_p
        L.trgsw_new_key.argtypes = [C.POINTER(_TRLWEKey), C.c_int, C.c_int]
        L.mosfhet_seed.argtypes = [C.c_uint64]
        L.double2torus.restype = C.c_uint64
        L.double2torus.argtypes = [C.c_double]
        L.mosfhet_gen_bootstrap_key_flat.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_TLWEKey)]
        L.mosfhet_gen_tlwe_ks_key_flat.argtypes = [C.c_void_p, C.POINTER(_TLWEKey), C.POINTER(_TLWEKey), C.c_int, C.c_int]
        L.mosfhet_tlwe_sample_flat.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(_TLWEKey)]
        L._host_ready = True
    return L


def seed(s):
    _lib().mosfhet_seed(C.c_uint64(s))


def double2torus(x):
    return _lib().double2torus(C.c_double(x))


class LweKey:
    def __init__(self, n, sigma, ptr=None):
        self.p = ptr if ptr is not None else _lib().tlwe_new_binary_key(n, sigma)
        self.n, self.sigma = n, sigma

    @property
    def s(self):
        return np.ctypeslib.as_array(self.p.contents.s, shape=(self.n,)).copy()


class RlweKey:
    def __init__(self, N, k, sigma):
        self.p = _lib().trlwe_new_binary_key(N, k, sigma)
        self.N, self.k, self.sigma = N, k, sigma

    @property
    def s(self):
        return np.stack([np.ctypeslib.as_array(self.p.contents.s[i].contents.coeffs, shape=(self.N,)).copy()
                         for i in range(self.k)])

    def extracted_lwe_key(self):
        """The LWE key of sample-extracted ciphertexts (trlwe_extract_tlwe_key, src/trlwe.c:531-538)."""
        L = _lib()
        kp = L.tlwe_alloc_key(self.k * self.N, self.sigma)
        L.trlwe_extract_tlwe_key(kp, self.p)
        return LweKey(self.k * self.N, self.sigma, ptr=kp)


def gen_bootstrap_key(rlwe_key, lwe_key, l, Bg_bit):
    """Torus-domain bootstrap key u64[n][(k+1)l][k+1][N], BK_i = TRGSW(s_i) (src/bootstrap.c:14-18)."""
    L = _lib()
    gk = L.trgsw_new_key(rlwe_key.p, l, Bg_bit)
    out = np.empty((lwe_key.n, (rlwe_key.k + 1) * l, rlwe_key.k + 1, rlwe_key.N), dtype=np.uint64)
    L.mosfhet_gen_bootstrap_key_flat(out.ctypes.data_as(C.c_void_p), gk, lwe_key.p)
    return out


def gen_tlwe_ks_key(out_key, in_key, t, base_bit):
    """u64[n_in][t][2^bb-1][n_out+1] (src/tlwe.c:193-212)."""
    out = np.empty((in_key.n, t, (1 << base_bit) - 1, out_key.n + 1), dtype=np.uint64)
    _lib().mosfhet_gen_tlwe_ks_key_flat(out.ctypes.data_as(C.c_void_p), out_key.p, in_key.p, t, base_bit)
    return out


def tlwe_samples(messages, key):
    """Encrypt a list of torus messages -> u64[count][n+1]."""
    out = np.empty((len(messages), key.n + 1), dtype=np.uint64)
    L = _lib()
    for i, m in enumerate(messages):
        L.mosfhet_tlwe_sample_flat(out[i].ctypes.data_as(C.c_void_p), C.c_uint64(int(m) & (2 ** 64 - 1)), key.p)
    return out


def tlwe_phase(ct, s):
    """Phases b - <a, s> of u64[count][n+1] under the key words s (numpy, exact mod 2^64)."""
    ct = np.asarray(ct, dtype=np.uint64)
    s = np.asarray(s, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return ct[:, -1] - (ct[:, :-1] * s[None, :]).sum(axis=1, dtype=np.uint64)


def torus_packing(lut, k, N):
    """trlwe_torus_packing (src/trlwe.c:662-667): trivial TRLWE u64[k+1][N] with b[i] = lut[i / (N/size)]."""
    lut = np.asarray(lut, dtype=np.uint64)
    tv = np.zeros((k + 1, N), dtype=np.uint64)
    tv[k] = np.repeat(lut, N // lut.size)
    return tv


def torus_packing_many_lut(lut, k, N, lut_size, n_luts):
    """trlwe_torus_packing_many_LUT (src/trlwe.c:677-687): n_luts LUTs of lut_size slots, interleaved."""
    lut = np.asarray(lut, dtype=np.uint64)
    tv = np.zeros((k + 1, N), dtype=np.uint64)
    span = N // (lut_size * n_luts)
    for i in range(lut_size):
        for j in range(n_luts):
            tv[k, (i * n_luts + j) * span:(i * n_luts + j + 1) * span] = lut[j * lut_size + i]
    return tv


def gen_bootstrap_key_ga(rlwe_key, lwe_key, l, Bg_bit):
    """Torus-domain GA bootstrap key u64[n][2l][2][N], BK_i = TRGSW(X^{s_i}) (src/bootstrap_ga.c:17-20)."""
    L = _lib()
    gk = L.trgsw_new_key(rlwe_key.p, l, Bg_bit)
    out = np.empty((lwe_key.n, (rlwe_key.k + 1) * l, rlwe_key.k + 1, rlwe_key.N), dtype=np.uint64)
    L.mosfhet_gen_bootstrap_key_ga_flat.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_TLWEKey)]
    L.mosfhet_gen_bootstrap_key_ga_flat(out.ctypes.data_as(C.c_void_p), gk, lwe_key.p)
    return out


def gen_automorphism_keyset(rlwe_key, t, base_bit):
    """Automorphism key-switch keys u64[N][t][2][N], entry j for generator 2j+1 (src/keyswitch.c:500-511)."""
    L = _lib()
    N = rlwe_key.N
    out = np.empty((N, t, 2, N), dtype=np.uint64)
    L.mosfhet_gen_automorphism_keyset_flat.argtypes = [C.c_void_p, C.POINTER(_TRLWEKey), C.c_int, C.c_int]
    L.mosfhet_gen_automorphism_keyset_flat(out.ctypes.data_as(C.c_void_p), rlwe_key.p, t, base_bit)
    return out


def automorphism_key_sources(s):
    """The N source polynomials of trlwe_new_automorphism_KS_keyset (src/keyswitch.c:500-511, skip_even): entry j is the key polynomial s under
    X -> X^(2j+1) (polynomial_permute, src/polynomial.c:442-450) -- what Engine.generate_trlwe_ks_keys encrypts into the key set on the device."""
    s = np.asarray(s, dtype=np.uint64)
    N = s.size
    i = np.arange(N, dtype=np.uint64)
    out = np.zeros((N, N), dtype=np.uint64)
    for j in range(N):
        idx = i * np.uint64(2 * j + 1)
        neg = (idx & np.uint64(N)) != 0
        with np.errstate(over="ignore"):
            out[j, (idx & np.uint64(N - 1)).astype(np.int64)] = np.where(neg, np.uint64(0) - s, s)
    return out


def gen_priv_ks_key(out_key, in_key, t, base_bit):
    """trlwe_new_priv_KS_key (src/keyswitch.c:39-50) as u64[2][t][2][N]: entry 0 from -s_out*s_in, entry 1 from -s_out."""
    L = _lib()
    out = np.empty((2, t, 2, out_key.N), dtype=np.uint64)
    L.mosfhet_gen_priv_ks_key_flat.argtypes = [C.c_void_p, C.POINTER(_TRLWEKey), C.POINTER(_TRLWEKey), C.c_int, C.c_int]
    L.mosfhet_gen_priv_ks_key_flat(out.ctypes.data_as(C.c_void_p), out_key.p, in_key.p, t, base_bit)
    return out


def gen_packing1_ks_key(out_key, in_key, t, base_bit):
    """trlwe_new_packing1_KS_key (src/keyswitch.c:368-390), rows uncompressed: u64[n][t][2^bb-1][2][N]."""
    L = _lib()
    out = np.empty((in_key.n, t, (1 << base_bit) - 1, 2, out_key.N), dtype=np.uint64)
    L.mosfhet_gen_packing1_ks_key_flat.argtypes = [C.c_void_p, C.POINTER(_TRLWEKey), C.POINTER(_TLWEKey), C.c_int, C.c_int]
    L.mosfhet_gen_packing1_ks_key_flat(out.ctypes.data_as(C.c_void_p), out_key.p, in_key.p, t, base_bit)
    return out
