"""Oracle B: independent big-integer restatement of the MATHEMATICAL definitions (test
infrastructure only).  Nothing here shares code or structure with oracle/pf_oracle.c: the NTT is
evaluated directly as a(psi^(2*brv(i)+1)), products are schoolbook negacyclic convolutions, and
multi-limb results are checked through CRT reconstruction.  O(N^2): use for N <= 1024.

Definitions follow SURVEY.md section 8c (the chosen spec for the un-vendored SEAL semantics):
forward NTT maps natural-order coefficients to bit-reversed evaluation points,
  A[i] = sum_j a[j] * psi^((2*brv(i)+1) * j)  mod q,   psi = minimal primitive 2N-th root mod q.
"""


def bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def is_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def primitive_roots_2n(N, q):
    """All primitive 2N-th roots of unity mod q (as a sorted list); brute force via one generator."""
    assert (q - 1) % (2 * N) == 0
    for x in range(2, 1 << 20):
        g = pow(x, (q - 1) // (2 * N), q)
        if pow(g, N, q) == q - 1:
            return sorted(pow(g, k, q) for k in range(1, 2 * N, 2))
    raise ValueError("no root")


def min_psi(N, q):
    return primitive_roots_2n(N, q)[0]


def ntt_direct(a, q, psi):
    """Direct O(N^2) evaluation; natural in, bit-reversed out."""
    N = len(a)
    logn = N.bit_length() - 1
    pw = [1] * (2 * N)
    for i in range(1, 2 * N):
        pw[i] = pw[i - 1] * psi % q
    out = []
    for i in range(N):
        e = 2 * bitrev(i, logn) + 1
        acc = 0
        idx = 0
        for j in range(N):
            acc += a[j] * pw[idx]
            idx += e
            if idx >= 2 * N:
                idx -= 2 * N
        out.append(acc % q)
    return out


def intt_direct(A, q, psi):
    """Inverse of ntt_direct: a[j] = N^-1 * sum_i A[i] * psi^(-(2*brv(i)+1)*j)."""
    N = len(A)
    logn = N.bit_length() - 1
    ipsi = pow(psi, q - 2, q)
    pw = [1] * (2 * N)
    for i in range(1, 2 * N):
        pw[i] = pw[i - 1] * ipsi % q
    ninv = pow(N, q - 2, q)
    es = [2 * bitrev(i, logn) + 1 for i in range(N)]
    out = []
    for j in range(N):
        acc = 0
        for i in range(N):
            acc += A[i] * pw[(es[i] * j) % (2 * N)]
        out.append(acc % q * ninv % q)
    return out


def negacyclic_mul(a, b, q):
    """Schoolbook product in Z_q[X]/(X^N+1)."""
    N = len(a)
    res = [0] * N
    for i, ai in enumerate(a):
        if ai == 0:
            continue
        for j, bj in enumerate(b):
            k = i + j
            if k < N:
                res[k] += ai * bj
            else:
                res[k - N] -= ai * bj
    return [r % q for r in res]


def crt(residues, moduli):
    """Reconstruct x mod prod(moduli) from its residues."""
    M = 1
    for q in moduli:
        M *= q
    x = 0
    for r, q in zip(residues, moduli):
        Mi = M // q
        x += r * Mi * pow(Mi, -1, q)
    return x % M, M


def key_switch(target, ksk_ntt, ct, moduli, psis):
    """Independent restatement of RNS key switching with one special prime (last modulus), everything in the
    coefficient domain with big integers (no NTT-domain products): used to pin oracle A's pfo_key_switch.
    target [L][N] coefficient form; ksk_ntt [L][2][K][N] NTT form; ct [2][L][N]; returns new ct."""
    K = len(moduli)
    L = K - 1
    N = len(target[0])
    P = moduli[-1]
    half = P // 2
    # keys to coefficient form, per modulus
    key = [[[intt_direct(ksk_ntt[i][c][j], moduli[j], psis[j]) for j in range(K)] for c in range(2)] for i in range(L)]
    out = [[list(ct[c][j]) for j in range(L)] for c in range(2)]
    for c in range(2):
        S = []
        for j in range(K):
            m = moduli[j]
            acc = [0] * N
            for i in range(L):
                d = [x % m for x in target[i]]
                prod = negacyclic_mul(d, key[i][c][j], m)
                acc = [(a + p) % m for a, p in zip(acc, prod)]
            S.append(acc)
        t = [(x + half) % P for x in S[K - 1]]
        for j in range(L):
            q = moduli[j]
            pinv = pow(P, -1, q)
            for n in range(N):
                r = (S[j][n] - t[n] + half) * pinv % q
                out[c][j][n] = (out[c][j][n] + r) % q
    return out
