"""Generates tests/golden/ntt_golden.npz from Oracle B (oracle/bigint_ref.py, the big-integer O(N^2)
restatement of the mathematical definitions).  The reference holds no fixtures for this path
(SURVEY.md section 4), so these vectors pin Oracle A, the host simulator and the HIP kernels to an
independent computation.  Run: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import bigint_ref as B  # noqa: E402

SEED = 20250801
CASES = [  # (N, q)  -- q = 1 mod 2N, SEAL BFVDefault primes where they apply
    (8, 0x7E00001), (16, 0xFFFFEE001), (64, 0x7FFFFFD8001), (64, 0x7FFFFFFFE90001),
    (1024, 0x7E00001), (1024, 0xFFFFFFFC001),
]


def main():
    rng = np.random.default_rng(SEED)
    out = {}
    for ci, (N, q) in enumerate(CASES):
        assert B.is_prime(q) and (q - 1) % (2 * N) == 0
        psi = B.min_psi(N, q)
        a = [int(x) for x in rng.integers(0, q, N, dtype=np.uint64)]
        b = [int(x) for x in rng.integers(0, q, N, dtype=np.uint64)]
        a[0], a[-1], b[1] = q - 1, 0, q - 1          # edge residues
        A = B.ntt_direct(a, q, psi)
        Bn = B.ntt_direct(b, q, psi)
        prod = B.negacyclic_mul(a, b, q)
        assert B.intt_direct(A, q, psi) == a
        assert [x * y % q for x, y in zip(A, Bn)] == B.ntt_direct(prod, q, psi)
        pre = f"c{ci}_"
        out[pre + "N"] = np.uint64(N)
        out[pre + "q"] = np.uint64(q)
        out[pre + "psi"] = np.uint64(psi)
        out[pre + "a"] = np.array(a, dtype=np.uint64)
        out[pre + "b"] = np.array(b, dtype=np.uint64)
        out[pre + "ntt_a"] = np.array(A, dtype=np.uint64)
        out[pre + "ntt_b"] = np.array(Bn, dtype=np.uint64)
        out[pre + "a_times_b"] = np.array(prod, dtype=np.uint64)
        out[pre + "a_plus_b"] = np.array([(x + y) % q for x, y in zip(a, b)], dtype=np.uint64)
        out[pre + "a_minus_b"] = np.array([(x - y) % q for x, y in zip(a, b)], dtype=np.uint64)
        out[pre + "neg_a"] = np.array([(-x) % q for x in a], dtype=np.uint64)
        print("case", ci, "N", N, "q bits", q.bit_length(), "psi", psi)
    out["n_cases"] = np.uint64(len(CASES))
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "ntt_golden.npz"), **out)


if __name__ == "__main__":
    main()
