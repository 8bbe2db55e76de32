"""End-to-end semantic check of the encrypted-distance path: a query vector is BFV-encrypted on the "client" (plain
Python big integers + the CPU oracle, test code), the GPU evaluates ct x pt against packed database rows exactly as
the server would (pf_ct_pt_mul with NTT-form plaintexts, then pf_poly_add to fold two ciphertexts), and decryption
yields the plaintext inner products -- i.e. the homomorphic part of the squared L2 distance
||q - x||^2 = ||q||^2 - 2 q.x + ||x||^2 the reference computes in the clear in Server::preciseSearch
(/root/reference/src/server/server_lib.cpp:140-167; the TODOs at include/client/client_lib.h:14,28-30 are this step).

Packing: q(X) = sum q_i X^i; row j is placed as -x_i X^(N - i + 128 j) (i > 0) and x_0 X^(128 j), so that coefficient
128 j of q(X) p(X) mod (X^N + 1) equals q . x_j.  N / 128 rows per ciphertext x plaintext product."""
import numpy as np
import pytest

import oracle
from oracle import bigint_ref as B

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

N, D = 4096, 128
QS = oracle.BFV_DEFAULT[4096][:2]          # 36 + 36 bits
T = 1 << 25                                 # plaintext modulus > 128 * 255^2


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    return "cuda:0"


def _rns(poly_int):
    """integer coefficients (any sign) -> [L][N] canonical residues"""
    return np.stack([np.array([int(c) % q for c in poly_int], dtype=np.uint64) for q in QS])


def test_encrypted_inner_products_decrypt_correctly():
    import prefhetch_amd as pf
    rng = np.random.default_rng(20250801)
    Q = QS[0] * QS[1]
    delta = Q // T
    o = oracle.Oracle(N, QS)
    ctx = pf.RnsContext(N, QS, _dev())
    # --- client: secret key, query, symmetric BFV encryption  ct = (-(a s + e) + delta m, a)
    s = rng.integers(-1, 2, N)
    s_ntt = o.ntt_forward(_rns(s))
    query = rng.integers(0, 256, D)
    m = np.zeros(N, dtype=object)
    m[:D] = query
    cts = []
    for _ in range(2):                                   # two encryptions of the same query (fresh randomness)
        a = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in QS])
        e = np.rint(rng.normal(0, 3.2, N)).astype(np.int64)
        a_s = o.ntt_inverse(o.dyadic_mul(o.ntt_forward(a), s_ntt))
        c0 = np.stack([np.array([(int(delta) * int(m[i]) - int(a_s[l][i]) - int(e[i])) % q for i in range(N)], dtype=np.uint64)
                       for l, q in enumerate(QS)])
        cts.append(np.stack([c0, a]))                    # [2][L][N]
    ct = np.stack(cts)                                    # [B=2][2][L][N]
    # --- server: two different blocks of N/128 database rows, one plaintext polynomial each
    rows = rng.integers(0, 256, (2, N // D, D))
    pts = []
    for blk in range(2):
        p = np.zeros(N, dtype=object)
        for j in range(N // D):
            for i in range(D):
                k = D * j - i
                if k >= 0:
                    p[k] += int(rows[blk, j, i])
                else:
                    p[k + N] -= int(rows[blk, j, i])      # X^N = -1
        pts.append(_rns(p))
    pt = pf.to_device_u64(np.stack(pts), _dev())          # [2][L][N] coefficient form
    ctx.ntt_forward_(pt)                                  # the server keeps its plaintext DB in NTT form
    prod = ctx.ct_pt_mul(pf.to_device_u64(ct, _dev()), pt)          # GPU: fused NTT -> dyadic -> INTT
    both = ctx.add(prod[0].contiguous(), prod[1].contiguous())      # Enc(q.x_blk0 + q.x_blk1) slot-wise
    # --- client: decrypt  m' = round(t/Q * (c0 + c1 s))  mod t
    def decrypt(c):
        c = np.ascontiguousarray(c)
        phase = o.addsub(c[0], o.ntt_inverse(o.dyadic_mul(o.ntt_forward(c[1]), s_ntt)), o.ADD)
        out = []
        for i in range(N):
            x, _ = B.crt([int(phase[0][i]), int(phase[1][i])], QS)
            if x > Q // 2:
                x -= Q
            out.append(((x * T + Q // 2) // Q) % T)
        return np.array(out, dtype=np.int64)

    prod_h = pf.to_host_u64(prod)
    for blk in range(2):
        dec = decrypt(prod_h[blk])
        want = rows[blk] @ query                                     # q . x_j for the block's rows
        assert (dec[::D] == want % T).all(), blk
        # the remaining distance terms are plaintext: ||q||^2 client side, ||x||^2 server side
        dist = (query @ query) - 2 * dec[::D] + (rows[blk] ** 2).sum(-1)
        assert (dist == ((rows[blk] - query) ** 2).sum(-1)).all()
    dec_sum = decrypt(pf.to_host_u64(both))
    assert (dec_sum[::D] == ((rows[0] + rows[1]) @ query) % T).all()
