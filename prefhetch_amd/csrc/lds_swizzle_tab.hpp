// lds_swizzle_tab.hpp -- LDS slot swizzles of the NTT exchanges, found by tools/lds_swizzle_search.py.
// slot(i) = i ^ XOR_t (((i >> s_t) & (2^w_t - 1)) << d_t): GF(2)-linear and unitriangular (a bijection); each
// entry makes ds_read_b64 and ds_write_b64 conflict-free from both layouts the exchange joins
// (bank model: MI355X_MICROARCH.md, LDS table).  PF_SWZ(LOGN, LOGR, exchange index, three s, w, d terms; w == 0 = unused).
#pragma once
namespace pf {
struct SwzTerm { int s, w, d; };
struct SwzEntry { int logn, logr, pair; SwzTerm t[3]; };
#define PF_SWZ(LN, LR, P, s0, w0, d0, s1, w1, d1, s2, w2, d2) \
    SwzEntry{LN, LR, P, {SwzTerm{s0, w0, d0}, SwzTerm{s1, w1, d1}, SwzTerm{s2, w2, d2}}},
constexpr SwzEntry SWZ_TABLE[] = {
    PF_SWZ(10, 4, 0, 6, 3, 2,  0, 0, 0,  0, 0, 0)
    PF_SWZ(10, 4, 1, 4, 1, 1,  2, 2, 0,  6, 3, 2)
    PF_SWZ(10, 5, 0, 5, 5, 0,  0, 0, 0,  0, 0, 0)
    PF_SWZ(11, 4, 0, 7, 2, 3,  0, 0, 0,  0, 0, 0)
    PF_SWZ(11, 4, 1, 7, 2, 3,  3, 3, 0,  0, 0, 0)
    PF_SWZ(11, 5, 0, 6, 4, 1,  0, 0, 0,  0, 0, 0)
    PF_SWZ(11, 5, 1, 1, 1, 0,  4, 1, 0,  6, 4, 1)
    PF_SWZ(12, 4, 0, 8, 2, 3,  0, 0, 0,  0, 0, 0)
    PF_SWZ(12, 4, 1, 4, 1, 2,  5, 5, 0,  0, 0, 0)
    PF_SWZ(12, 5, 0, 7, 3, 2,  0, 0, 0,  0, 0, 0)
    PF_SWZ(12, 5, 1, 2, 1, 0,  4, 2, 0,  7, 3, 2)
    PF_SWZ(13, 4, 0, 0, 0, 0,  0, 0, 0,  0, 0, 0)
    PF_SWZ(13, 4, 1, 5, 4, 1,  0, 0, 0,  0, 0, 0)
    PF_SWZ(13, 4, 2, 1, 1, 0,  6, 1, 4,  5, 4, 1)
    PF_SWZ(13, 5, 0, 8, 2, 3,  0, 0, 0,  0, 0, 0)
    PF_SWZ(13, 5, 1, 3, 1, 0,  8, 2, 3,  4, 3, 0)
    PF_SWZ(14, 4, 0, 0, 0, 0,  0, 0, 0,  0, 0, 0)
    PF_SWZ(14, 4, 1, 6, 3, 2,  0, 0, 0,  0, 0, 0)
    PF_SWZ(14, 4, 2, 4, 1, 1,  2, 2, 0,  6, 3, 2)
    PF_SWZ(14, 5, 0, 9, 1, 4,  0, 0, 0,  0, 0, 0)
    PF_SWZ(14, 5, 1, 4, 1, 3,  5, 5, 0,  0, 0, 0)
    PF_SWZ(15, 5, 0, 10, 1, 4,  0, 0, 0,  0, 0, 0)
    PF_SWZ(15, 6, 0, 9, 2, 3,  0, 0, 0,  0, 0, 0)
    PF_SWZ(15, 6, 1, 4, 2, 1,  6, 5, 0,  0, 0, 0)
    PF_SWZ(15, 5, 1, 5, 1, 3,  6, 5, 0,  0, 0, 0)
};
#undef PF_SWZ
constexpr SwzEntry swz_lookup(int logn, int logr, int pair) {
    for (const SwzEntry &e : SWZ_TABLE)
        if (e.logn == logn && e.logr == logr && e.pair == pair) return e;
    return SwzEntry{logn, logr, pair, {SwzTerm{0, 0, 0}, SwzTerm{0, 0, 0}, SwzTerm{0, 0, 0}}};
}
}  // namespace pf
