// ct x pt for BOTH polynomials of one (ciphertext, limb) pair, coefficient form in and out, no accumulation (k_ctpt_pair: launches that
// do not fill the device twice).  Such a launch is one round of workgroups marching in lockstep -- everybody loads, then everybody computes,
// then everybody stores -- so memory and VALU never overlap (N = 4096 x 2 limbs x 256: 10 us of loads + 15 us of arithmetic + 7 us of stores
// = the 31 us measured).  Here a workgroup requests its second polynomial right behind its first and transforms the first while the second
// is still travelling; the first's stores drain under the second's arithmetic, and the plaintext limb is read once for both.
template <class G, class A, class Sync>
PF_HD void body_ctpt_pair(const A &ar, const typename A::Tw *__restrict__ tw, const typename A::Tw *__restrict__ itw,
                          const uint64_t *ct0, const uint64_t *ct1, const uint64_t *pt, uint64_t *out0, uint64_t *out1,
                          typename A::V *lds, int tid, Sync &&sync) {
    using V = typename A::V;
    V r[G::R], pv[G::R];
    uint64_t nxt[G::R];
    load_l0<G, A>(r, ct0, tid);
#pragma unroll
    for (int k = 0; k < G::R; ++k) nxt[k] = (ct1 + G::koff(0, k))[tid];          // raw: converting would wait for it
#pragma unroll 1
    for (int c = 0; c < 2; ++c) {
        PassTw<G, A, G::LAST> tl;
        if (c == 0) {
            fwd_all<G, A>(r, ar, tw, lds, tid, sync);
            load_last<G, A>(pv, pt, tid);
        } else {
            fwd_all<G, A>(r, ar, tw, lds, tid, sync);
        }
        dyadic_all<G, A, true>(r, pv, ar);
        tl.load(itw, tid);
        PF_LAUNDER(tid);
        inv_all<G, A>(r, ar, tl, itw, lds, tid, sync);
        uint64_t *out = c ? out1 : out0;
#pragma unroll
        for (int k = 0; k < G::R; ++k) (out + G::koff(0, k))[tid] = A::to_u64(ar.canon_small(r[k]));
        if (c == 0) {
#pragma unroll
            for (int k = 0; k < G::R; ++k) r[k] = A::from_u64(nxt[k]);
        }
    }
}

