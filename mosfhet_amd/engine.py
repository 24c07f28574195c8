"""ctypes binding of the C ABI in include/mosfhet_hip.h.

Torus64 buffers are torch.int64 CUDA tensors (same bits as uint64; torch has no full uint64 support).
Every function launches on torch's current stream of the engine's device.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# parameter sets of the reference's benchmarks (test/benchmark.c:53-54 and :64-75)
PARAMS_SET1 = dict(n=585, N=1024, k=1, l=2, Bg_bit=8, t=5, base_bit=2,
                   lwe_sigma=9.141776004202573e-5, rlwe_sigma=2.989040792967434e-8)
PARAMS_LVL2 = dict(n=632, N=2048, k=1, l=4, Bg_bit=9, t=8, base_bit=4,
                   lwe_sigma=2.0 ** -15, rlwe_sigma=2.0 ** -44)
# the reference's SET_2 (its default, test/tests.c:43-45) and SET_3 (:47-49), eprint 2022/704 table 4
PARAMS_SET2 = dict(n=744, N=2048, k=1, l=1, Bg_bit=23, t=5, base_bit=3,
                   lwe_sigma=7.747831515176779e-6, rlwe_sigma=2.2148688116005568e-16)
PARAMS_SET3 = dict(n=807, N=4096, k=1, l=1, Bg_bit=22, t=5, base_bit=3,
                   lwe_sigma=1.0562341599676662e-6, rlwe_sigma=2.168404344971009e-19)


class MosfhetHipError(RuntimeError):
    pass


def lib_path():
    return os.path.join(_HERE, "libmosfhet_hip.so")


def lib():
    """Load the native library; fail loudly if it has not been built (no fallback path exists)."""
    global _LIB
    if _LIB is None:
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).  Import it
        # first so this process holds ONE runtime: loading ours first would bind torch to a second, mismatched
        # runtime/HSA pair and the device disappears ("no HIP device visible").
        import torch  # noqa: F401
        path = lib_path()
        if not os.path.exists(path):
            raise MosfhetHipError(
                "native library %s is missing: run `python -m mosfhet_amd.build` (or __graft_entry__.build())" % path)
        L = C.CDLL(path)
        L.mosfhet_hip_last_error.restype = C.c_char_p
        L.mosfhet_hip_version.restype = C.c_char_p
        L.mosfhet_hip_bsk_bytes.restype = C.c_size_t
        L.mosfhet_hip_bsk_bytes.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise MosfhetHipError("mosfhet_hip error %d: %s" % (rc, lib().mosfhet_hip_last_error().decode()))


def to_device(a, device):
    """numpy uint64 array -> torch.int64 CUDA tensor with the same bits."""
    import torch
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).to(device)


def to_numpy(t):
    """torch.int64 tensor -> numpy uint64 array with the same bits."""
    return t.detach().cpu().numpy().view(np.uint64)


def _ptr(t):
    assert t.is_cuda and t.is_contiguous(), "expected a contiguous CUDA tensor"
    return C.c_void_p(t.data_ptr())


class BootstrapKey:
    def __init__(self, engine, handle, n, k, N, l, Bg_bit):
        self.engine, self.h = engine, handle
        self.n, self.k, self.N, self.l, self.Bg_bit = n, k, N, l, Bg_bit

    @property
    def nbytes(self):
        return lib().mosfhet_hip_bsk_bytes(self.h)

    def export_dft(self):
        out = np.empty((self.n, (self.k + 1) * self.l, self.k + 1, self.N), dtype=np.float64)
        _check(lib().mosfhet_hip_bsk_export_dft(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def free(self):
        if self.h:
            lib().mosfhet_hip_bsk_destroy(self.h)
            self.h = None


class KeySwitchKey:
    def __init__(self, engine, handle, n_in, n_out, t, base_bit):
        self.engine, self.h = engine, handle
        self.n_in, self.n_out, self.t, self.base_bit = n_in, n_out, t, base_bit

    @property
    def nbytes(self):
        lib().mosfhet_hip_ksk_bytes.restype = C.c_size_t
        lib().mosfhet_hip_ksk_bytes.argtypes = [C.c_void_p]
        return lib().mosfhet_hip_ksk_bytes(self.h)

    def free(self):
        if self.h:
            lib().mosfhet_hip_ksk_destroy(self.h)
            self.h = None


class AutomorphismKeys:
    def __init__(self, engine, handle, N, t, base_bit):
        self.engine, self.h, self.N, self.t, self.base_bit = engine, handle, N, t, base_bit

    def free(self):
        if self.h:
            lib().mosfhet_hip_gak_destroy(self.h)
            self.h = None


class Engine:
    """One engine per (process, GPU): wraps mosfhet_hip_ctx_t."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise MosfhetHipError("no GPU visible: mosfhet_amd has no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.h = C.c_void_p()
        _check(lib().mosfhet_hip_ctx_create(C.byref(self.h), int(device)))

    def close(self):
        if self.h:
            lib().mosfhet_hip_ctx_destroy(self.h)
            self.h = None

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def pace_skip_credit(self):
        """paced launches of this device that will still skip the team rendezvous because an earlier launch's wait ran out (synchronises the device)"""
        v = C.c_int()
        _check(lib().mosfhet_hip_pace_skip_credit(self.h, C.byref(v)))
        return v.value

    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.int64, device=self.device)

    # ---- keys ----
    def load_bootstrap_key(self, bk_torus, k, l, Bg_bit):
        """bk_torus: numpy uint64 [n][(k+1)l][k+1][N] (torus domain) -> device DFT key."""
        bk_torus = np.ascontiguousarray(bk_torus, dtype=np.uint64)
        n, rows, k1, N = bk_torus.shape
        assert rows == (k + 1) * l and k1 == k + 1
        h = C.c_void_p()
        _check(lib().mosfhet_hip_bsk_create(self.h, C.byref(h), bk_torus.ctypes.data_as(C.c_void_p), n, k, N, l, Bg_bit))
        return BootstrapKey(self, h, n, k, N, l, Bg_bit)

    def load_bootstrap_key_unfolded(self, su, l, Bg_bit, unfolding):
        """su: numpy uint64 [n 2^u/u][2l][2][N] (torus domain) -> device key taking the unfolded blind rotation."""
        su = np.ascontiguousarray(su, dtype=np.uint64)
        cnt, rows, two, N = su.shape
        n = cnt * unfolding >> unfolding
        h = C.c_void_p()
        _check(lib().mosfhet_hip_bsk_unfolded_create(self.h, C.byref(h), su.ctypes.data_as(C.c_void_p), n, N, l, Bg_bit, unfolding))
        return BootstrapKey(self, h, n, 1, N, l, Bg_bit)

    def load_bootstrap_key_device(self, d_bk, k, l, Bg_bit):
        n, rows, k1, N = d_bk.shape
        h = C.c_void_p()
        _check(lib().mosfhet_hip_bsk_create_from_device(self.h, C.byref(h), _ptr(d_bk), n, k, N, l, Bg_bit, self._stream()))
        self.torch.cuda.current_stream(self.device).synchronize()
        return BootstrapKey(self, h, n, k, N, l, Bg_bit)

    def load_keyswitch_key(self, ksk, base_bit):
        ksk = np.ascontiguousarray(ksk, dtype=np.uint64)
        n_in, t, per_j, row = ksk.shape
        assert per_j == (1 << base_bit) - 1
        h = C.c_void_p()
        _check(lib().mosfhet_hip_ksk_create(self.h, C.byref(h), ksk.ctypes.data_as(C.c_void_p), n_in, row - 1, t, base_bit))
        return KeySwitchKey(self, h, n_in, row - 1, t, base_bit)

    def load_automorphism_keys(self, ak_torus, base_bit):
        """ak_torus: numpy uint64 [N][t][2][N] (torus domain, entry j for generator 2j+1)."""
        ak_torus = np.ascontiguousarray(ak_torus, dtype=np.uint64)
        N, t, two, N2 = ak_torus.shape
        assert two == 2 and N2 == N
        h = C.c_void_p()
        _check(lib().mosfhet_hip_gak_create(self.h, C.byref(h), ak_torus.ctypes.data_as(C.c_void_p), N, t, base_bit))
        return AutomorphismKeys(self, h, N, t, base_bit)

    def load_trlwe_ks_keys(self, rows, base_bit):
        """rows: numpy uint64 [entries][t][2][N] (torus domain) -> device FFT key-switch key set."""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        entries, t, two, N = rows.shape
        h = C.c_void_p()
        _check(lib().mosfhet_hip_trlwe_ksk_create(self.h, C.byref(h), rows.ctypes.data_as(C.c_void_p), entries, N, t, base_bit))
        return AutomorphismKeys(self, h, N, t, base_bit)

    def generate_trlwe_ks_keys(self, s_out, msgs, t, base_bit, sigma, seed):
        """On-device FFT key-switch key set: entry e switches from the polynomial msgs[e] to the binary key s_out."""
        s_out = np.ascontiguousarray(s_out, dtype=np.uint64)
        msgs = np.ascontiguousarray(msgs, dtype=np.uint64).reshape(-1, s_out.size)
        h = C.c_void_p()
        _check(lib().mosfhet_hip_trlwe_ksk_generate(self.h, C.byref(h), s_out.ctypes.data_as(C.c_void_p), s_out.size, msgs.ctypes.data_as(C.c_void_p),
                                                    msgs.shape[0], t, base_bit, C.c_double(sigma), C.c_uint64(seed)))
        return AutomorphismKeys(self, h, s_out.size, t, base_bit)

    def load_packing1_key(self, rows, base_bit):
        """rows: numpy uint64 [n][t][2^bb-1][2][N] -> device LWE -> TRLWE packing key."""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        n, t, per_j, two, N = rows.shape
        h = C.c_void_p()
        _check(lib().mosfhet_hip_packing1_ksk_create(self.h, C.byref(h), rows.ctypes.data_as(C.c_void_p), n, N, t, base_bit))
        return KeySwitchKey(self, h, n, 2 * N - 1, t, base_bit)

    def trlwe_keyswitch(self, tks, entry, ct, out=None):
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, 2, tks.N)
        _check(lib().mosfhet_hip_trlwe_keyswitch_batch(self.h, tks.h, int(entry), _ptr(out), _ptr(ct), count, self._stream()))
        return out

    def trlwe_priv_keyswitch_2(self, tks, ct, out=None):
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, 2, tks.N)
        _check(lib().mosfhet_hip_trlwe_priv_keyswitch_2_batch(self.h, tks.h, _ptr(out), _ptr(ct), count, self._stream()))
        return out

    def trlwe_packing1_keyswitch(self, ksk, ct, out=None):
        count = ct.shape[0]
        N = (ksk.n_out + 1) // 2
        if out is None:
            out = self.empty(count, 2, N)
        _check(lib().mosfhet_hip_trlwe_packing1_keyswitch_batch(self.h, ksk.h, _ptr(out), _ptr(ct), count, self._stream()))
        return out

    @staticmethod
    def _level_events(level_events, l):
        """l torch.cuda.Event objects (or None entries) -> the void*[l] the *_batch_ev entry points take; an event is recorded when its gadget level is final"""
        if level_events is None:
            return None
        if len(level_events) != l:
            raise MosfhetHipError("level_events: %d events for %d gadget levels" % (len(level_events), l))
        arr = (C.c_void_p * l)()
        for i, e in enumerate(level_events):
            if e is not None:
                e.record()   # (a torch event has no handle before its first record; the engine records it again where it belongs)
                arr[i] = e.cuda_event
        return arr

    def circuit_bootstrap_3(self, bsk, kska, kskb, ct, out=None, level_events=None):
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, 2 * bsk.l, 2, bsk.N)
        ev = self._level_events(level_events, bsk.l)
        _check(lib().mosfhet_hip_circuit_bootstrap_3_batch_ev(self.h, bsk.h, kska.h, kskb.h, _ptr(out), _ptr(ct), count, self._stream(), ev))
        return out

    def public_mux(self, p0, p1, sel, Bg_bit, out=None):
        count, l, two, N = sel.shape
        if out is None:
            out = self.empty(count, 2, N)
        _check(lib().mosfhet_hip_public_mux_batch(self.h, _ptr(out), _ptr(p0), _ptr(p1), _ptr(sel), N, l, Bg_bit, count, self._stream()))
        return out

    def full_domain_functional_bootstrap_KS21(self, bsk, pksk, tv, ct, torus_base, variant=0, out=None):
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, bsk.N + 1)
        _check(lib().mosfhet_hip_full_domain_functional_bootstrap_KS21_batch(self.h, bsk.h, pksk.h, _ptr(out), _ptr(tv), _ptr(ct), count, torus_base,
                                                                             variant, self._stream()))
        return out

    def multivalue_bootstrap_phase1(self, bsk, ct, torus_base, out=None):
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, torus_base + 1, 2, bsk.N)
        _check(lib().mosfhet_hip_multivalue_bootstrap_phase1_batch(self.h, bsk.h, _ptr(out), _ptr(ct), count, torus_base, self._stream()))
        return out

    def multivalue_bootstrap_phase2(self, lut, rotated, torus_base, log_torus_base, out=None):
        count, _, _, N = rotated.shape
        if out is None:
            out = self.empty(count, N + 1)
        h_lut = (C.c_int * torus_base)(*[int(x) for x in lut])
        _check(lib().mosfhet_hip_multivalue_bootstrap_phase2_batch(self.h, _ptr(out), h_lut, _ptr(rotated), N, torus_base, log_torus_base, count,
                                                                   self._stream()))
        return out

    def load_priv_key(self, rows, base_bit):
        """rows: numpy uint64 [n+1][t][2^bb-1][2][N] -> device table-lookup private key-switch key."""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        n1, t, per_j, two, N = rows.shape
        h = C.c_void_p()
        _check(lib().mosfhet_hip_priv_ksk_create(self.h, C.byref(h), rows.ctypes.data_as(C.c_void_p), n1 - 1, N, t, base_bit))
        return KeySwitchKey(self, h, n1, 2 * N - 1, t, base_bit)

    def trlwe_priv_keyswitch(self, ksk, ct, out=None):
        count = ct.shape[0]
        N = (ksk.n_out + 1) // 2
        if out is None:
            out = self.empty(count, 2, N)
        _check(lib().mosfhet_hip_trlwe_priv_keyswitch_batch(self.h, ksk.h, _ptr(out), _ptr(ct), count, self._stream()))
        return out

    def circuit_bootstrap(self, bsk, kska, kskb, ct, variant=0, out=None, level_events=None):
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, 2 * bsk.l, 2, bsk.N)
        ev = self._level_events(level_events, bsk.l)
        _check(lib().mosfhet_hip_circuit_bootstrap_batch_ev(self.h, bsk.h, kska.h, kskb.h, _ptr(out), _ptr(ct), count, variant, self._stream(), ev))
        return out

    def functional_bootstrap_trgsw_phase1(self, bsk, ct, torus_base, out=None):
        count = ct.shape[0]
        if out is None:
            out = self.torch.empty(count, 2 * bsk.l, 2, bsk.N, dtype=self.torch.float64, device=self.device)
        _check(lib().mosfhet_hip_functional_bootstrap_trgsw_phase1_batch(self.h, bsk.h, _ptr(out), _ptr(ct), count, torus_base, self._stream()))
        return out

    def functional_bootstrap_trgsw_phase2(self, bsk, g_dft, tv, out=None):
        count = g_dft.shape[0]
        if out is None:
            out = self.empty(count, bsk.N + 1)
        _check(lib().mosfhet_hip_functional_bootstrap_trgsw_phase2_batch(self.h, bsk.h, _ptr(out), _ptr(g_dft), _ptr(tv), tv.shape[0], count,
                                                                         self._stream()))
        return out

    def trlwe_tensor_prod_FFT(self, rlk, c1, c2, precision, out=None):
        count = c1.shape[0]
        if out is None:
            out = self.empty(count, 2, rlk.N)
        _check(lib().mosfhet_hip_trlwe_tensor_prod_FFT_batch(self.h, rlk.h, _ptr(out), _ptr(c1), _ptr(c2), precision, count, self._stream()))
        return out

    def tlwe_mul(self, pksk, rlk, c1, c2, precision, out=None):
        count = c1.shape[0]
        if out is None:
            out = self.empty(count, rlk.N + 1)
        _check(lib().mosfhet_hip_tlwe_mul_batch(self.h, pksk.h, rlk.h, _ptr(out), _ptr(c1), _ptr(c2), precision, count, self._stream()))
        return out

    def full_domain_functional_bootstrap_CLOT21(self, bsk, pksk, rlk, tv, ct, precision, variant=0, out=None):
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, bsk.N + 1)
        _check(lib().mosfhet_hip_full_domain_functional_bootstrap_CLOT21_batch(self.h, bsk.h, pksk.h, rlk.h, _ptr(out), _ptr(tv), _ptr(ct), count,
                                                                               precision, variant, self._stream()))
        return out

    def multivalue_bootstrap_UBR_phase1(self, bsk, ct, unfolding, out=None):
        count = ct.shape[0]
        if out is None:
            out = self.torch.empty(count, bsk.n // unfolding, 2 * bsk.l, 2, bsk.N, dtype=self.torch.float64, device=self.device)
        _check(lib().mosfhet_hip_multivalue_bootstrap_UBR_phase1_batch(self.h, bsk.h, _ptr(out), _ptr(ct), count, self._stream()))
        return out

    def multivalue_bootstrap_UBR_phase2(self, bsk, tvs, ct, sa, torus_base, out=None):
        count, tv_count = ct.shape[0], tvs.shape[0]
        if out is None:
            out = self.empty(count, tv_count, bsk.N + 1)
        _check(lib().mosfhet_hip_multivalue_bootstrap_UBR_phase2_batch(self.h, bsk.h, _ptr(out), _ptr(tvs), tv_count, _ptr(ct), _ptr(sa), count,
                                                                       torus_base, self._stream()))
        return out

    def trlwe_mv_extract(self, ct, mode, amount, out=None):
        count, _, N = ct.shape
        if out is None:
            out = self.empty(count, amount, N + 1) if mode == 0 else self.empty(count, N + 1)
        _check(lib().mosfhet_hip_trlwe_mv_extract_batch(self.h, _ptr(out), _ptr(ct), N, mode, amount, count, self._stream()))
        return out

    @staticmethod
    def set_keygen_secret(key32):
        """Install the 256-bit ChaCha20 key the on-device generators draw their NOISE under (process-wide; default: from the operating system) and
        restart its per-call nonce sequence: the same secret followed by the same generate calls reproduces the same keys (tests, mosfhet_seed)."""
        key32 = bytes(key32)
        assert len(key32) == 32
        _check(lib().mosfhet_hip_set_keygen_secret(key32))

    def ks_words_gave_up(self):
        """wavefronts of the word-lane key-switch kernel whose bounded counter wait ran out since the library was loaded (0 unless something is broken)"""
        n = C.c_uint()
        _check(lib().mosfhet_hip_ks_words_gave_up(self.h, C.byref(n)))
        return n.value

    def generate_table_key(self, kind, s_out, s_in, t, base_bit, sigma, seed, compressed=False):
        """On-device packing (kind 0) / private (kind 1) key-switch key; returns a KeySwitchKey.  compressed: keep only the b halves in HBM and
        regenerate the masks inside the key-switch kernels (same rows, same results)."""
        s_out = np.ascontiguousarray(s_out, dtype=np.uint64)
        s_in = np.ascontiguousarray(s_in, dtype=np.uint64)
        h = C.c_void_p()
        gen = lib().mosfhet_hip_trlwe_table_ksk_generate_compressed if compressed else lib().mosfhet_hip_trlwe_table_ksk_generate
        _check(gen(self.h, C.byref(h), kind, s_out.ctypes.data_as(C.c_void_p), s_out.size,
                                                          s_in.ctypes.data_as(C.c_void_p), s_in.size, t, base_bit, C.c_double(sigma), C.c_uint64(seed)))
        return KeySwitchKey(self, h, s_in.size + kind, 2 * s_out.size - 1, t, base_bit)

    def generate_lut_packing_key(self, s_out, s_in, t, base_bit, torus_base, sigma, seed):
        """On-device LUT-packing key (trlwe_new_packing_KS_key, src/keyswitch.c:214-241): rows (i, e, j, v), n * torus_base digit sources."""
        s_out = np.ascontiguousarray(s_out, dtype=np.uint64)
        s_in = np.ascontiguousarray(s_in, dtype=np.uint64)
        h = C.c_void_p()
        _check(lib().mosfhet_hip_trlwe_lut_packing_ksk_generate(self.h, C.byref(h), s_out.ctypes.data_as(C.c_void_p), s_out.size, s_in.ctypes.data_as(C.c_void_p),
                                                                s_in.size, t, base_bit, torus_base, C.c_double(sigma), C.c_uint64(seed)))
        return KeySwitchKey(self, h, s_in.size * torus_base, 2 * s_out.size - 1, t, base_bit)

    def trlwe_lut_packing_keyswitch(self, ksk, torus_base, cts, out=None):
        """trlwe_packing_keyswitch (src/keyswitch.c:346-366) for a batch: cts [count][torus_base][n + 1] -> [count][2][N]"""
        count = cts.shape[0]
        N = (ksk.n_out + 1) // 2
        if out is None:
            out = self.empty(count, 2, N)
        _check(lib().mosfhet_hip_trlwe_lut_packing_keyswitch_batch(self.h, ksk.h, int(torus_base), _ptr(out), _ptr(cts), count, self._stream()))
        return out

    def generate_bootstrap_key(self, s_rlwe, s_lwe, l, Bg_bit, sigma, seed, ga=False):
        """On-device bootstrap key BK_i = TRGSW(s_lwe[i]) (ga: TRGSW(X^{s_lwe[i]})) under the binary TRLWE key s_rlwe."""
        s_rlwe = np.ascontiguousarray(s_rlwe, dtype=np.uint64)
        s_lwe = np.ascontiguousarray(s_lwe, dtype=np.uint64)
        h = C.c_void_p()
        if s_rlwe.ndim == 2:   # [k][N]: any k <= 3 / any ring the engine serves (general-ring path included)
            k, N = s_rlwe.shape
            if ga:
                raise MosfhetHipError("generate_bootstrap_key: the Galois form exists for k = 1 only")
            _check(lib().mosfhet_hip_bsk_generate_k(self.h, C.byref(h), s_rlwe.ctypes.data_as(C.c_void_p), k, N, s_lwe.ctypes.data_as(C.c_void_p), s_lwe.size,
                                                    l, Bg_bit, C.c_double(sigma), C.c_uint64(seed)))
            return BootstrapKey(self, h, s_lwe.size, k, N, l, Bg_bit)
        _check(lib().mosfhet_hip_bsk_generate(self.h, C.byref(h), s_rlwe.ctypes.data_as(C.c_void_p), s_rlwe.size, s_lwe.ctypes.data_as(C.c_void_p), s_lwe.size,
                                              l, Bg_bit, C.c_double(sigma), C.c_uint64(seed), int(ga)))
        return BootstrapKey(self, h, s_lwe.size, 1, s_rlwe.size, l, Bg_bit)

    def generate_bootstrap_key_unfolded(self, s_rlwe, s_lwe, l, Bg_bit, sigma, seed, unfolding):
        """On-device key of new_bootstrap_key(.., unfolding > 1): per group of `unfolding` key bits the 2^unfolding samples TRGSW(group spells j)."""
        s_rlwe = np.ascontiguousarray(s_rlwe, dtype=np.uint64)
        s_lwe = np.ascontiguousarray(s_lwe, dtype=np.uint64)
        h = C.c_void_p()
        _check(lib().mosfhet_hip_bsk_unfolded_generate(self.h, C.byref(h), s_rlwe.ctypes.data_as(C.c_void_p), s_rlwe.size, s_lwe.ctypes.data_as(C.c_void_p), s_lwe.size,
                                                       l, Bg_bit, C.c_double(sigma), C.c_uint64(seed), int(unfolding)))
        return BootstrapKey(self, h, s_lwe.size, 1, s_rlwe.size, l, Bg_bit)

    def generate_keyswitch_key(self, s_out, s_in, t, base_bit, sigma, seed, compressed=False):
        """On-device LWE -> LWE key-switch table (tlwe_new_KS_key) from the binary keys s_in (switched from) and s_out (switched to)."""
        s_out = np.ascontiguousarray(s_out, dtype=np.uint64)
        s_in = np.ascontiguousarray(s_in, dtype=np.uint64)
        h = C.c_void_p()
        _check(lib().mosfhet_hip_tlwe_ksk_generate(self.h, C.byref(h), s_out.ctypes.data_as(C.c_void_p), s_out.size, s_in.ctypes.data_as(C.c_void_p), s_in.size,
                                                   t, base_bit, C.c_double(sigma), C.c_uint64(seed), int(compressed)))
        return KeySwitchKey(self, h, s_in.size, s_out.size, t, base_bit)

    # ---- key images (on-disk formats; include/mosfhet_hip.h "Key images") ----
    def export_bootstrap_key(self, bsk):
        """The engine's own image of a bootstrap key (DFT rows; torus-domain samples for an unfolded key) as raw bytes."""
        out = np.empty(bsk.nbytes, dtype=np.uint8)
        _check(lib().mosfhet_hip_bsk_export(bsk.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def import_bootstrap_key(self, image, n, k, N, l, Bg_bit, unfolding=1):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        # the C ABI takes a bare pointer: the length is checked here (an image is mosfhet_hip_bsk_bytes long: DFT entries, or the torus-domain samples of an unfolded key)
        want = n * (k + 1) * l * (k + 1) * N * 8 if unfolding == 1 else (n << unfolding) // unfolding * 2 * l * 2 * N * 8
        if image.nbytes != want:
            raise MosfhetHipError("bootstrap-key image of %d bytes, %d expected for n=%d k=%d N=%d l=%d unfolding=%d" % (image.nbytes, want, n, k, N, l, unfolding))
        h = C.c_void_p()
        _check(lib().mosfhet_hip_bsk_import(self.h, C.byref(h), image.ctypes.data_as(C.c_void_p), n, k, N, l, Bg_bit, unfolding))
        return BootstrapKey(self, h, n, k, N, l, Bg_bit)

    def clone_key(self, key):
        """A copy of a key handle of ANOTHER engine (or this one) on this engine's device, device to device (mosfhet_hip_*_clone: SURVEY 8(e), keys
        replicated per GPU).  Returns (handle of the same class, route: 0 same device, 1 peer to peer, 2 device to device without peer access, 3 host bounce)."""
        h = C.c_void_p()
        if isinstance(key, BootstrapKey):
            _check(lib().mosfhet_hip_bsk_clone(self.h, C.byref(h), key.h))
            out = BootstrapKey(self, h, key.n, key.k, key.N, key.l, key.Bg_bit)
        elif isinstance(key, KeySwitchKey):
            _check(lib().mosfhet_hip_ksk_clone(self.h, C.byref(h), key.h))
            out = KeySwitchKey(self, h, key.n_in, key.n_out, key.t, key.base_bit)
        else:
            _check(lib().mosfhet_hip_gak_clone(self.h, C.byref(h), key.h))
            out = AutomorphismKeys(self, h, key.N, key.t, key.base_bit)
        return out, int(lib().mosfhet_hip_last_clone_route())

    def bootstrap_key_info(self, bsk):
        v = (C.c_int * 6)()
        _check(lib().mosfhet_hip_bsk_info(bsk.h, v))
        return tuple(v)

    def alloc_keyswitch_key(self, kind, n, n_out_or_N, t, base_bit):
        """Empty table key to be filled by import_keyswitch_rows: kind 0 LWE -> LWE (n_out), 1 packing (N), 2 private (N)."""
        h = C.c_void_p()
        _check(lib().mosfhet_hip_ksk_alloc(self.h, C.byref(h), kind, n, n_out_or_N, t, base_bit))
        return KeySwitchKey(self, h, n + (kind == 2), n_out_or_N if kind == 0 else 2 * n_out_or_N - 1, t, base_bit)

    def import_keyswitch_rows(self, ksk, first_row, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        assert rows.shape[1] == ksk.n_out + 1
        _check(lib().mosfhet_hip_ksk_import_rows(ksk.h, C.c_size_t(first_row), C.c_size_t(rows.shape[0]), rows.ctypes.data_as(C.c_void_p)))

    def export_key_rows(self, ksk, first_row, count):
        out = np.empty((count, ksk.n_out + 1), dtype=np.uint64)
        _check(lib().mosfhet_hip_ksk_export_rows(ksk.h, C.c_size_t(first_row), C.c_size_t(count), out.ctypes.data_as(C.c_void_p)))
        return out

    def keyswitch_functional_bootstrap(self, ksk, bsk, tv, ct, torus_base, extract=True, out=None):
        """tlwe_keyswitch N -> n followed by functional_bootstrap[_wo_extract] in one call."""
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, bsk.N + 1) if extract else self.empty(count, 2, bsk.N)
        _check(lib().mosfhet_hip_keyswitch_functional_bootstrap_batch(self.h, ksk.h, bsk.h, _ptr(out), _ptr(tv), self._tv(tv, bsk, count), _ptr(ct), count,
                                                                      torus_base, int(extract), self._stream()))
        return out

    def cmux(self, bsk, key_index, in0, in1, out=None):
        """out[b] = in0[b] + key[key_index] (.) (in1[b] - in0[b]); out may be in0."""
        count = in0.shape[0]
        if out is None:
            out = self.empty(count, bsk.k + 1, bsk.N)
        _check(lib().mosfhet_hip_cmux_batch(self.h, bsk.h, int(key_index), _ptr(out), _ptr(in0), _ptr(in1), count, self._stream()))
        return out

    def trlwe_eval_automorphism(self, gak, ct, gen, out=None):
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, 2, gak.N)
        _check(lib().mosfhet_hip_trlwe_eval_automorphism_batch(self.h, gak.h, _ptr(out), _ptr(ct), int(gen), count, self._stream()))
        return out

    def functional_bootstrap_ga(self, bsk, gak, tv, ct, torus_base, extract=True, out=None):
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, bsk.N + 1) if extract else self.empty(count, 2, bsk.N)
        _check(lib().mosfhet_hip_functional_bootstrap_ga_batch(self.h, bsk.h, gak.h, _ptr(out), _ptr(tv), self._tv(tv, bsk, count),
                                                               _ptr(ct), count, torus_base, int(extract), self._stream()))
        return out

    # ---- bootstraps ----
    def _tv(self, tv, bsk, count):
        assert tv.dim() == 3 and tv.shape[1] == bsk.k + 1 and tv.shape[2] == bsk.N, tv.shape
        assert tv.shape[0] in (1, count)
        return tv.shape[0]

    def programmable_bootstrap(self, bsk, tv, ct, precision, kappa=0, theta=0, out=None):
        count = ct.shape[0]
        assert ct.shape[1] == bsk.n + 1
        if out is None:
            out = self.empty(count, bsk.k * bsk.N + 1)
        _check(lib().mosfhet_hip_programmable_bootstrap_batch(self.h, bsk.h, _ptr(out), _ptr(tv), self._tv(tv, bsk, count),
                                                              _ptr(ct), count, precision, kappa, theta, self._stream()))
        return out

    def functional_bootstrap(self, bsk, tv, ct, torus_base, out=None):
        count = ct.shape[0]
        assert ct.shape[1] == bsk.n + 1
        if out is None:
            out = self.empty(count, bsk.k * bsk.N + 1)
        _check(lib().mosfhet_hip_functional_bootstrap_batch(self.h, bsk.h, _ptr(out), _ptr(tv), self._tv(tv, bsk, count),
                                                            _ptr(ct), count, torus_base, self._stream()))
        return out

    def functional_bootstrap_wo_extract(self, bsk, tv, ct, torus_base, out=None):
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, bsk.k + 1, bsk.N)
        _check(lib().mosfhet_hip_functional_bootstrap_wo_extract_batch(
            self.h, bsk.h, _ptr(out), _ptr(tv), self._tv(tv, bsk, count), _ptr(ct), count, torus_base, self._stream()))
        return out

    def blind_rotate_(self, bsk, acc, ct):
        """In place: acc[b] <- blind_rotate(acc[b], ct[b].a, BK)."""
        count = ct.shape[0]
        assert acc.shape == (count, bsk.k + 1, bsk.N)
        _check(lib().mosfhet_hip_blind_rotate_batch(self.h, bsk.h, _ptr(acc), _ptr(ct), count, self._stream()))
        return acc

    def external_product(self, bsk, key_index, ct, out=None):
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, bsk.k + 1, bsk.N)
        _check(lib().mosfhet_hip_external_product_batch(self.h, bsk.h, key_index, _ptr(out), _ptr(ct), count, self._stream()))
        return out

    # ---- digit-parallel radix-integer callers (capi_vec.inc): integers are digit-major [d][M][N+1] ----
    def vector_ops(self, bsk, ksk, pksk, torus_base):
        h = C.c_void_p()
        _check(lib().mosfhet_hip_vec_create(self.h, C.byref(h), bsk.h, ksk.h, pksk.h if pksk is not None else None, int(torus_base)))
        return VectorOps(self, h, bsk, torus_base)

    # ---- DFT-level objects of the legacy API (include/mosfhet.h of the reference: TRGSW_DFT, TRLWE_DFT stay on the device in slot order) ----
    def trgsw_to_dft(self, trgsw):
        """trgsw_to_DFT (src/trgsw.c:359-366) for a batch: [count][(k+1)l][k+1][N] torus words -> the same shape in doubles (N/2 complex per polynomial)"""
        shape = tuple(trgsw.shape)
        return self.torus_to_dft(trgsw.reshape(-1, shape[-1])).reshape(shape)

    def external_product_dft(self, trgsw_dft, ct, l, Bg_bit):
        """trgsw_mul_trlwe_DFT (src/trgsw.c:385-423), result left in the DFT domain.  trgsw_dft: [2l][2][N] doubles (one TRGSW for the batch)
        or [count][2l][2][N] (one per unit); ct: [count][2][N]"""
        count, two, N = ct.shape
        per_unit = trgsw_dft.dim() == 4
        assert (trgsw_dft.shape[0] == count) if per_unit else True
        out = self.torch.empty(count, 2, N, dtype=self.torch.float64, device=self.device)
        stride = C.c_size_t(2 * l * 2 * N if per_unit else 0)
        _check(lib().mosfhet_hip_external_product_dft_batch(self.h, _ptr(trgsw_dft), stride, _ptr(out), _ptr(ct), N, l, Bg_bit, count, self._stream()))
        return out

    def bootstrap_key_view(self, trgsw_dft, k, l, Bg_bit):
        """non-owning key handle over n TRGSW_DFT entries already on the device ([n][(k+1)l][k+1][N] doubles): blind_rotate(tv, a, TRGSW_DFT *s, size)"""
        n, N = trgsw_dft.shape[0], trgsw_dft.shape[-1]
        h = C.c_void_p()
        _check(lib().mosfhet_hip_bsk_view_create(self.h, C.byref(h), _ptr(trgsw_dft), n, k, N, l, Bg_bit))
        key = BootstrapKey(self, h, n, k, N, l, Bg_bit)
        key._keep = trgsw_dft     # the view borrows the tensor's memory
        return key

    def public_mux_dft(self, p0, p1, sel_dft, Bg_bit, out=None):
        """public_mux (src/bootstrap.c:369-389) with the selector rows already in the DFT domain: sel_dft [count][l][2][N] doubles"""
        count, l, two, N = sel_dft.shape
        if out is None:
            out = self.empty(count, 2, N)
        _check(lib().mosfhet_hip_public_mux_dft_batch(self.h, _ptr(out), _ptr(p0), _ptr(p1), _ptr(sel_dft), N, l, Bg_bit, count, self._stream()))
        return out

    def blind_rotate_ga(self, bsk, gak, acc, ct):
        """blind_rotate_ga (src/bootstrap_ga.c:35-60) in place on acc [count][2][N]"""
        _check(lib().mosfhet_hip_blind_rotate_ga_batch(self.h, bsk.h, gak.h, _ptr(acc), _ptr(ct), ct.shape[0], self._stream()))
        return acc

    def trlwe_eval_automorphism_entry(self, gak, entry, ct, gen, out=None):
        """trlwe_eval_automorphism (src/trlwe.c:775-781) with the key-set entry the caller names"""
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, 2, gak.N)
        _check(lib().mosfhet_hip_trlwe_eval_automorphism_entry_batch(self.h, gak.h, int(entry), _ptr(out), _ptr(ct), int(gen), count, self._stream()))
        return out

    # ---- polynomial level ----
    def torus_to_dft(self, polys):
        count, N = polys.shape
        out = self.torch.empty(count, N, dtype=self.torch.float64, device=self.device)
        _check(lib().mosfhet_hip_torus_to_dft_batch(self.h, _ptr(out), _ptr(polys), N, count, self._stream()))
        return out

    def dft_to_torus(self, dfts):
        count, N = dfts.shape
        out = self.empty(count, N)
        _check(lib().mosfhet_hip_dft_to_torus_batch(self.h, _ptr(out), _ptr(dfts), N, count, self._stream()))
        return out

    def dft_mul(self, a, b, out=None, addto=False):
        count, N = a.shape
        if out is None:
            assert not addto
            out = self.torch.empty_like(a)
        _check(lib().mosfhet_hip_dft_mul_batch(self.h, _ptr(out), _ptr(a), _ptr(b), N, count, int(addto), self._stream()))
        return out

    # ---- key switch ----
    def tlwe_keyswitch(self, ksk, ct, out=None):
        count = ct.shape[0]
        assert ct.shape[1] == ksk.n_in + 1
        if out is None:
            out = self.empty(count, ksk.n_out + 1)
        _check(lib().mosfhet_hip_tlwe_keyswitch_batch(self.h, ksk.h, _ptr(out), _ptr(ct), count, self._stream()))
        return out

    def trlwe_extract_tlwe(self, trlwe, idx, out=None):
        count, k1, N = trlwe.shape
        if out is None:
            out = self.empty(count, (k1 - 1) * N + 1)
        if k1 == 2:
            _check(lib().mosfhet_hip_trlwe_extract_tlwe_batch(self.h, _ptr(out), _ptr(trlwe), N, idx, count, self._stream()))
        else:
            _check(lib().mosfhet_hip_trlwe_extract_tlwe_k_batch(self.h, _ptr(out), _ptr(trlwe), k1 - 1, N, idx, count, self._stream()))
        return out

    def tlwe_addto_(self, out, ct):
        count, row = ct.shape
        assert out.shape == ct.shape
        _check(lib().mosfhet_hip_tlwe_addto_batch(self.h, _ptr(out), _ptr(ct), row - 1, count, self._stream()))
        return out

    def full_domain_functional_bootstrap(self, bsk, ksk, tv, ct, precision, out=None):
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, bsk.k * bsk.N + 1)
        _check(lib().mosfhet_hip_full_domain_functional_bootstrap_batch(
            self.h, bsk.h, ksk.h, _ptr(out), _ptr(tv), self._tv(tv, bsk, count), _ptr(ct), count, precision, self._stream()))
        return out

    def multivalue_bootstrap_CLOT21(self, bsk, tv, ct, torus_base, n_luts, out=None):
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, n_luts, bsk.k * bsk.N + 1)
        _check(lib().mosfhet_hip_multivalue_bootstrap_CLOT21_batch(
            self.h, bsk.h, _ptr(out), _ptr(tv), self._tv(tv, bsk, count), _ptr(ct), count, torus_base, n_luts, self._stream()))
        return out

    # ---- measurement ----
    def time_programmable_bootstrap(self, bsk, tv, ct, precision, reps, out=None):
        """Average milliseconds per kernel launch, hipEvents on the launch stream."""
        count = ct.shape[0]
        if out is None:
            out = self.empty(count, bsk.k * bsk.N + 1)
        ms = C.c_float()
        _check(lib().mosfhet_hip_time_programmable_bootstrap(self.h, bsk.h, _ptr(out), _ptr(tv), self._tv(tv, bsk, count),
                                                             _ptr(ct), count, precision, reps, self._stream(), C.byref(ms)))
        return ms.value

    def sync(self):
        self.torch.cuda.current_stream(self.device).synchronize()


def set_unfold_split_max(max_batch):
    """chunk size of the two-phase unfolded bootstrap (-1 default, 0 = always the fused kernel)"""
    _check(lib().mosfhet_hip_set_unfold_split_max(int(max_batch)))


def set_unfold2_dft(on):
    """unfolding-2 keys: DFT-domain assembly of the per-group TRGSW (default) or the torus-domain one of u = 4 and 8"""
    _check(lib().mosfhet_hip_set_unfold2_dft(int(bool(on))))


def set_team_max_batch(max_batch):
    """Batches up to this size use the latency-oriented bootstrap kernel at N = 1024 (0 disables it)."""
    _check(lib().mosfhet_hip_set_team_max_batch(int(max_batch)))


def set_wide_team_max_batch(max_batch):
    """Batches up to this size use the latency-oriented bootstrap kernel at N = 2048 (0 disables it)."""
    _check(lib().mosfhet_hip_set_wide_team_max_batch(int(max_batch)))


def set_split_max_batch(max_batch):
    """N = 2048, l = 2, 4, 6: batches up to this size take two CUs per bootstrap (pbs_split_kernel; sums per accumulator component: FFT-level different bits). -1 = CUs / 2 (default), 0 = never."""
    _check(lib().mosfhet_hip_set_split_max_batch(int(max_batch)))


def set_ks_words(min_count):
    """Table key switches with 2 - 4 digit bits take the word-lane kernel (keyswitch_words_kernels.h) from this many ciphertexts on (default 17, 0 = never; same bits)."""
    _check(lib().mosfhet_hip_set_ks_words(int(min_count)))


def set_split_wait_limit(ticks):
    """bound (10 ns ticks) of the pairing wait of pbs_split_kernel; 0 = every bootstrap taken alone by one workgroup (same bits)"""
    _check(lib().mosfhet_hip_set_split_wait_limit(int(ticks)))


def split_last_launch():
    """(count, paired, alone) of this thread's last split launch"""
    c, p, a = C.c_int(), C.c_int(), C.c_int()
    _check(lib().mosfhet_hip_split_last_launch(C.byref(c), C.byref(p), C.byref(a)))
    return c.value, p.value, a.value


class VectorOps:
    """Handle of the digit-parallel integer callers (mosfhet_hip_vec_*): add / sub / ReLU / encrypted LUT over M independent radix-B integers"""
    def __init__(self, eng, h, bsk, torus_base):
        self.eng, self.h, self.bsk, self.torus_base = eng, h, bsk, torus_base

    def addsub(self, a, b, subtract=False, out=None):
        d, M, row = a.shape
        if out is None:
            out = self.eng.empty(d, M, row)
        _check(lib().mosfhet_hip_vec_addsub(self.h, _ptr(out), _ptr(a), _ptr(b), M, d, int(bool(subtract)), self.eng._stream()))
        return out

    def relu(self, a, out=None):
        d, M, row = a.shape
        if out is None:
            out = self.eng.empty(d, M, row)
        _check(lib().mosfhet_hip_vec_relu(self.h, _ptr(out), _ptr(a), M, d, self.eng._stream()))
        return out

    def encrypted_lut(self, table, sel):
        """table [size][M][N+1] is consumed; returns table[0] = the selected entries"""
        size, M, row = table.shape
        _check(lib().mosfhet_hip_vec_encrypted_lut(self.h, _ptr(table), _ptr(sel), size, M, self.eng._stream()))
        return table[0]

    def cmp(self, a, b, a_signed=True, b_signed=True):
        """[M][N+1]: digit 0 / 1 / 2 (over 2 B) for a < / = / > b"""
        d, M, row = a.shape
        out = self.eng.empty(M, row)
        _check(lib().mosfhet_hip_vec_cmp(self.h, _ptr(out), _ptr(a), _ptr(b), M, d, int(bool(a_signed)), int(bool(b_signed)), self.eng._stream()))
        return out

    def mux_array(self, tables, sel):
        """tables [size][d][M][N+1] (consumed), sel [log_B size][M][N+1] -> [d][M][N+1]"""
        size, d, M, row = tables.shape
        out = self.eng.empty(d, M, row)
        _check(lib().mosfhet_hip_vec_mux_array(self.h, _ptr(out), _ptr(tables), _ptr(sel), size, d, M, self.eng._stream()))
        return out

    def sl_add(self, a, g, b, h, out_digits, signed=True):
        da, M, row = a.shape
        out = self.eng.empty(out_digits, M, row)
        _check(lib().mosfhet_hip_vec_sl_add(self.h, _ptr(out), int(out_digits), _ptr(a), da, int(g), _ptr(b), b.shape[0], int(h), int(bool(signed)), M, self.eng._stream()))
        return out

    def extend(self, c, d_ini, signed=True):
        dc, M, _ = c.shape
        _check(lib().mosfhet_hip_vec_extend(self.h, _ptr(c), dc, int(d_ini), int(bool(signed)), M, self.eng._stream()))
        return c

    def mul(self, a, b, out_digits, signed=True):
        da, M, row = a.shape
        out = self.eng.empty(out_digits, M, row)
        _check(lib().mosfhet_hip_vec_mul(self.h, _ptr(out), int(out_digits), _ptr(a), da, _ptr(b), b.shape[0], int(bool(signed)), M, self.eng._stream()))
        return out

    def lut_cleartext(self, sel, lut, out_digits):
        """out[m] = lut[selector_m] for a cleartext table (numpy uint64 [size]); sel [levels][M][N+1] -> [out_digits][M][N+1]"""
        levels, M, row = sel.shape
        lut = np.ascontiguousarray(lut, dtype=np.uint64)
        out = self.eng.empty(out_digits, M, row)
        _check(lib().mosfhet_hip_vec_lut_cleartext(self.h, _ptr(out), _ptr(sel), lut.ctypes.data_as(C.c_void_p), int(lut.size), int(out_digits), M, self.eng._stream()))
        return out

    def free(self):
        if self.h:
            lib().mosfhet_hip_vec_destroy(self.h)
            self.h = None


def set_ep_plain_loop(on):
    """external products on two-wavefront teams (N = 2048, l = 4): the plain unit loop instead of the software-pipelined one (same bits)"""
    _check(lib().mosfhet_hip_set_ep_plain_loop(int(bool(on))))


def ep_kernel_info():
    """[(name, scratch bytes per lane of the pipelined build, launcher takes it)] of the multi-wavefront external-product instantiations launched so far"""
    out, i = [], 0
    name, scratch, takes = C.c_char_p(), C.c_int(), C.c_int()
    while lib().mosfhet_hip_ep_kernel_info(i, C.byref(name), C.byref(scratch), C.byref(takes)) == 0:
        out.append((name.value.decode(), scratch.value, bool(takes.value)))
        i += 1
    return out


def twiddles(N):
    out = np.empty(2 * (N // 2 - 1), dtype=np.float64)
    _check(lib().mosfhet_hip_twiddles(N, out.ctypes.data_as(C.c_void_p)))
    return out


def slot_order_to_oracle(dft, N):
    """Engine slot order (index m*T + thread, T = N/16 threads of 8 registers) -> oracle order (index thread*8 + m)."""
    assert N in (1024, 2048, 4096)
    M, T = N // 2, N // 16
    j = np.arange(M)
    dev = (j & 7) * T + (j >> 3)
    z = dft.reshape(-1, M, 2)
    return z[:, dev, :].reshape(dft.shape)
