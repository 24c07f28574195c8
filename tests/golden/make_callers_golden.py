#!/usr/bin/env python3
"""Generates tests/golden/callers.npz: outputs of the REAL reference (both back-ends) for the callers either side of the bootstrap
(full_domain_functional_bootstrap, multivalue_bootstrap_CLOT21, functional_bootstrap_ga, the TRLWE key switches, the unfolded
bootstrap) on the seeded inputs of callers_setup.py.  Data only; keys are regenerated from the seeds by the tests.  Needs oracle/_ref."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import oracle as O  # noqa: E402
from oracle import reflib  # noqa: E402
import callers_setup as S  # noqa: E402


def main():
    if not reflib.build():
        sys.exit("/root/reference is not present")
    O.build()
    out = {}
    A, G, K, U = S.fdfb_multivalue(O), S.galois(O), S.key_switches(O), S.unfolded(O)
    # fingerprints of the replayed streams (the tests assert them before comparing anything)
    out["fp"] = np.array([A["c_mv"][0], G["cts"][3][0], K["cs"][1][0], U["cts"][3][0]], dtype=np.uint64)
    for be in ("avx512", "ffnt"):
        if not reflib.available(be):
            continue
        ref = reflib.get(be)
        ref.init(S.N)
        h, kh = ref.bk_new(A["bk"], 1, S.L, S.BG), ref.ksk_new(A["ksk"], A["bb"])
        out["fdfb_" + be] = np.stack([ref.full_domain_functional_bootstrap(A["tv"], c, h, kh, 3) for c in A["cts"]])
        out["multivalue_" + be] = ref.multivalue_bootstrap_CLOT21(A["tv16"], A["c_mv"], h, 2, 8)
        ref.bk_free(h)
        ref.ksk_free(kh)
        out["trlwe_keyswitch_" + be] = ref.trlwe_keyswitch(G["c_ks"], G["ks"], 8)
        out["automorphism_" + be] = np.stack([ref.trlwe_eval_automorphism(G["c_aut"], g, G["ak"][(g - 1) // 2], S.BG) for g in G["gens"]])
        hg = ref.bk_ga_new(G["bk"], G["ak"], S.L, S.BG)
        out["ga_" + be] = np.stack([ref.functional_bootstrap_ga(G["tv"], c, hg, 4) for c in G["cts"]])
        ref.bk_ga_free(hg)
        out["priv_keyswitch_2_" + be] = ref.trlwe_priv_keyswitch_2(K["ct"], K["ks0"], K["ks1"], 3)
        out["packing1_" + be] = np.stack([ref.trlwe_packing1_keyswitch(c, K["kskb"], 3) for c in K["cs"]])
        hu = ref.bk_unfolded_new(U["su"], S.L, S.BG, U["unfolding"])
        out["unfolded_" + be] = np.stack([ref.functional_bootstrap(U["tv"], c, hu, 4) for c in U["cts"]])
        ref.bk_free(hu)
    path = os.path.join(HERE, "callers.npz")
    np.savez_compressed(path, **out)
    print("callers.npz: %d bytes, %d arrays" % (os.path.getsize(path), len(out)))


if __name__ == "__main__":
    main()
