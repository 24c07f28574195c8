"""Makes tests/golden/ufhe_vectors.npz: what the REFERENCE's radix-integer application (applications/multi-ciphertext-arith) decrypts, on the reference's own
AVX-512 library, for a fixed list of signed 8-bit inputs -- add, sub, ReLU, encrypted 16-entry LUT, comparison (signed and unsigned), cleartext 16-entry LUT, multiplication (signed into 32 bits, unsigned full product).  Needs /root/reference (build container only); compiles
tests/golden/ufhe_vectors_ref.c (own driver) with the reference's sources where they lie (oracle/ref/Makefile: ufhe_vectors_ref), runs it (minutes: the LUT-packing
key alone is 4.8 GB of CPU-encrypted rows) and stores inputs and decrypted outputs.  tests/test_gpu_parity.py::test_vector_integer_callers holds the
digit-parallel GPU forms to these numbers."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "ref"), os.path.join(ROOT, "oracle", "_ref", "ufhe_vectors_ref")])
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ufhe_vectors_ref"), str(count)], check=True, capture_output=True, text=True).stdout
    doc = json.loads(out)
    rows = doc["rows"]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ufhe_vectors.npz"), torus_base=np.int64(doc["torus_base"]),
                        a=np.array([r["a"] for r in rows], dtype=np.int64), b=np.array([r["b"] for r in rows], dtype=np.int64),
                        sel=np.array([r["sel"] for r in rows], dtype=np.int64), table=np.array([r["table"] for r in rows], dtype=np.int64),
                        add=np.array([r["add"] for r in rows], dtype=np.int64), sub=np.array([r["sub"] for r in rows], dtype=np.int64),
                        relu=np.array([r["relu"] for r in rows], dtype=np.int64), lut=np.array([r["lut"] for r in rows], dtype=np.int64),
                        cmp_signed=np.array([r["cmp_signed"] for r in rows], dtype=np.int64), cmp_unsigned=np.array([r["cmp_unsigned"] for r in rows], dtype=np.int64),
                        lut_cleartext=np.array([r["lut_cleartext"] for r in rows], dtype=np.int64),
                        mul_signed=np.array([r["mul_signed"] for r in rows], dtype=np.int64), mul_unsigned=np.array([r["mul_unsigned"] for r in rows], dtype=np.int64),
                        **{k: np.array([r[k] for r in rows], dtype=np.int64) for k in ("sl_g", "sl_h", "sl_add_signed", "sl_add_unsigned")})
    print("wrote tests/golden/ufhe_vectors.npz: %d rows" % len(rows))


if __name__ == "__main__":
    main()
