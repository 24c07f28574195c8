#!/usr/bin/env python3
"""Generates tests/golden/fileio.npz: files written by the REAL reference's own writers (tlwe_save_key, trlwe_save_key, trgsw_save_key,
tlwe_save_sample, trlwe_save_sample, tlwe_save_KS_key; src/tlwe.c:43-99,275-287, src/trlwe.c:24-29,230-237, src/trgsw.c:38-42) for seeded inputs,
stored as byte arrays next to those inputs -- data only.  Run in the build container (needs oracle/_ref, i.e. /root/reference)."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from oracle import reflib  # noqa: E402


def main():
    O.build()
    ref = reflib.get("ffnt")
    ref.init(64)
    rng = O.Rng(0xF11E)
    n, N, k, l, Bg_bit = 12, 64, 1, 3, 7
    lwe_sigma, rlwe_sigma = 2.0 ** -15, 2.0 ** -44
    lwe_s = O.gen_binary_key(rng, n)
    rlwe_s = O.gen_binary_key(rng, N).reshape(k, N)
    ct = O.tlwe_sample(rng, O.double2torus(0.125), lwe_s, lwe_sigma)
    rct = O.trlwe_sample(rng, O.u64(rng.words(N)), rlwe_s, rlwe_sigma)
    n_in, n_out, t, bb = 6, 5, 2, 2
    s_in, s_out = O.gen_binary_key(rng, n_in), O.gen_binary_key(rng, n_out)
    table = O.gen_tlwe_ks_key(rng, s_in, s_out, t, bb, 2.0 ** -20)
    with tempfile.TemporaryDirectory() as d:
        ref.save_host_objects(os.path.join(d, "host.bin"), lwe_s, lwe_sigma, rlwe_s, rlwe_sigma, l, Bg_bit, ct, rct)
        h = ref.ksk_new(table, bb)
        ref.ksk_save(os.path.join(d, "ks.bin"), h)
        # the reference's own reader takes the file back and key-switches with it
        h2 = ref.ksk_load(os.path.join(d, "ks.bin"))
        c_in = O.tlwe_sample(rng, O.double2torus(0.25), s_in, 2.0 ** -30)
        switched = ref.tlwe_keyswitch(c_in, h2, n_out)
        assert (switched == ref.tlwe_keyswitch(c_in, h, n_out)).all()
        host_file = np.fromfile(os.path.join(d, "host.bin"), dtype=np.uint8)
        ks_file = np.fromfile(os.path.join(d, "ks.bin"), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "fileio.npz"), n=n, N=N, k=k, l=l, Bg_bit=Bg_bit, lwe_sigma=lwe_sigma, rlwe_sigma=rlwe_sigma, lwe_s=lwe_s,
                        rlwe_s=rlwe_s, tlwe_ct=ct, trlwe_ct=rct, host_file=host_file, ks_table=table, ks_t=t, ks_base_bit=bb, ks_file=ks_file,
                        ks_in=c_in, ks_switched=switched)
    print("fileio.npz: host file %d bytes, key-switch key file %d bytes" % (len(host_file), len(ks_file)))


if __name__ == "__main__":
    main()
