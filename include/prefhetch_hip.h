/*
 * prefhetch_hip.h -- C ABI of libprefhetch_hip.so: the MI355X (gfx950) implementation of the
 * PreFHEtch server-side encrypted-query hot path.
 *
 * Drop-in boundary.  The reference's server (class Server, /root/reference/include/server/server_lib.h:12-50)
 * reaches its numeric engines -- Microsoft SEAL and the PreFHEtch-faiss fork, both linked C++ libraries,
 * /root/reference/CMakeLists.txt:22-38,62-68 -- by direct C++ calls.  This header is the FFI a maintainer
 * binds instead (plain pointers and sizes, no C++/torch/HIP types); INTEGRATION.md shows the binding.
 * Each entry point names the reference interface it stands in for.
 *
 * Conventions
 *   - Every data pointer is a DEVICE pointer (hipMalloc / pf_malloc / torch tensor data_ptr) on the
 *     device the handle was created on, unless the parameter name ends in _host or says "host or device".
 *   - `stream` is a hipStream_t passed as void* (NULL = the device's null stream).  Calls enqueue work
 *     and return; nothing synchronises unless documented.  The polynomial calls (NTT, dyadic, add, ct x pt)
 *     and pf_flat_search after pf_flat_reserve never allocate, free or synchronise, so they can be
 *     captured into a hipGraph; pf_key_switch (see pf_key_switch_reserve) and pf_ivfpq_search_lists say what they do.
 *   - RNS polynomial buffers: `n_limb_polys` polynomials of N uint64 coefficients, contiguous;
 *     polynomial p belongs to RNS limb (p % L).  A SEAL Ciphertext's data() -- size x L x N,
 *     poly-major, then limb, then coefficient -- and a batch of them therefore pass through unchanged.
 *     Coefficients are canonical residues in [0, q_limb) on input and on output.
 *   - Return value: PF_OK (0) or a negative pf_status; functions never throw and never abort.
 *     pf_last_error() gives a thread-local human-readable detail string.
 *   - Handles are not thread-safe individually; distinct handles may be used from distinct threads.
 */
#ifndef PREFHETCH_HIP_H
#define PREFHETCH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t pf_status;
enum {
    PF_OK = 0,
    PF_ERR_INVALID_ARG = -1,
    PF_ERR_UNSUPPORTED = -2,   /* parameter set outside what the kernels are built for */
    PF_ERR_HIP = -3,           /* a HIP runtime call failed; see pf_last_error() */
    PF_ERR_NO_DEVICE = -4,
    PF_ERR_OOM = -5
};

typedef struct pf_ctx pf_ctx;     /* one RNS ring: N, L moduli, twiddle tables in HBM */
typedef struct pf_flat pf_flat;   /* one brute-force L2 index: fp32 base matrix in HBM */
typedef void *pf_stream;          /* hipStream_t */

const char *pf_status_str(pf_status s);
const char *pf_last_error(void);
/* "" for the product build; otherwise the extra compiler flags the library was built with, prefixed "experiment:" when one of them
 * is a timing-only ablation or debug switch (such a library returns wrong results and must never be shipped) */
const char *pf_build_flags(void);
pf_status pf_device_count(int *count);

/* ---- device memory helpers, so that a host written without HIP headers can own buffers ---------- */
pf_status pf_malloc(int device, void **dptr, size_t bytes);
pf_status pf_free(int device, void *dptr);
pf_status pf_memcpy_h2d(int device, void *dst, const void *src_host, size_t bytes, pf_stream stream);
pf_status pf_memcpy_d2h(int device, void *dst_host, const void *src, size_t bytes, pf_stream stream);
pf_status pf_memcpy_d2d(int device, void *dst, const void *src, size_t bytes, pf_stream stream);
pf_status pf_stream_synchronize(int device, pf_stream stream);

/* ---- RNS ring context ------------------------------------------------------------------------- */
/* Builds, per modulus: the minimal primitive 2N-th root psi, forward / inverse twiddle tables (with
 * Shoup quotients, and FP64 images when q < 2^45), Barrett ratio floor(2^128/q); uploads them.
 * Stands in for seal::SEALContext / util::NTTTables construction.  N in {1024,...,32768} (power of 2),
 * every modulus prime, < 2^61, = 1 mod 2N.  Blocking (synchronises the upload). */
pf_status pf_ctx_create(pf_ctx **ctx, int device, uint32_t N, uint32_t L, const uint64_t *moduli_host);
pf_status pf_ctx_destroy(pf_ctx *ctx);
/* arith_path_out[l]: 0 = exact-FP64 butterflies (every q < 2^45), 2 = 64-bit Shoup butterflies with the range
 * corrections hoisted out (every q < 2^56), 1 = 64-bit Shoup/Harvey butterflies (any q < 2^61).  One family per
 * context.  Any out pointer may be NULL. */
pf_status pf_ctx_info(const pf_ctx *ctx, uint32_t *N, uint32_t *L, uint64_t *moduli_out_host,
                      uint64_t *psi_out_host, int32_t *arith_path_out_host);
/* Testing hook: 0 = automatic choice, 1 = 64-bit integer butterflies (family 2 where the moduli allow, else 1),
 * 2 = the general Harvey butterflies (family 1). */
pf_status pf_ctx_force_u64(pf_ctx *ctx, int on);

/* ---- polynomial arithmetic -------------------------------------------------------------------- */
/* util::ntt_negacyclic_harvey: in-place forward negacyclic NTT, natural order in, bit-reversed out. */
pf_status pf_ntt_forward(pf_ctx *ctx, uint64_t *polys, size_t n_limb_polys, pf_stream stream);
/* util::inverse_ntt_negacyclic_harvey: in-place inverse, bit-reversed in, natural out, scaled by N^-1. */
pf_status pf_ntt_inverse(pf_ctx *ctx, uint64_t *polys, size_t n_limb_polys, pf_stream stream);
/* The same transforms from `src` into `dst` (Evaluator::transform_to_ntt / transform_from_ntt with a destination):
 * src is left untouched; dst may equal src. */
pf_status pf_ntt_forward_to(pf_ctx *ctx, const uint64_t *src, uint64_t *dst, size_t n_limb_polys, pf_stream stream);
pf_status pf_ntt_inverse_to(pf_ctx *ctx, const uint64_t *src, uint64_t *dst, size_t n_limb_polys, pf_stream stream);
/* util::dyadic_product_coeffmod: out[i] = a[i]*b[i] mod q.  out may alias a or b. */
pf_status pf_dyadic_mul(pf_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n_limb_polys, pf_stream stream);
/* util::add_poly_coeffmod / sub_poly_coeffmod / negate_poly_coeffmod (Evaluator::add_inplace etc.). */
pf_status pf_poly_add(pf_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n_limb_polys, pf_stream stream);
pf_status pf_poly_sub(pf_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n_limb_polys, pf_stream stream);
pf_status pf_poly_negate(pf_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n_limb_polys, pf_stream stream);

/* Evaluator::multiply_plain (+ transform_to_ntt / transform_from_ntt / add_inplace), fused:
 *   ct      [B][2][L][N]   ciphertexts, coefficient form (or NTT form with PF_CTPT_IN_NTT)
 *   pt_ntt  [pt_count][L][N] plaintexts already in NTT form; pt_count == B, or 1 to broadcast
 *   out     [B][2][L][N]   product in coefficient form (or NTT form with PF_CTPT_OUT_NTT);
 *                          with PF_CTPT_ACCUMULATE out += product (out must hold valid residues in
 *                          the output domain).  out may alias ct.
 * One launch; per (ciphertext, poly, limb): NTT -> dyadic -> inverse NTT without leaving the CU. */
enum { PF_CTPT_ACCUMULATE = 1, PF_CTPT_IN_NTT = 2, PF_CTPT_OUT_NTT = 4 };
pf_status pf_ct_pt_mul(pf_ctx *ctx, const uint64_t *ct, const uint64_t *pt_ntt, size_t pt_count,
                       uint64_t *out, size_t B, int flags, pf_stream stream);

/* util::GaloisTool::apply_galois (coefficient form), the permutation half of Evaluator::apply_galois_inplace /
 * rotate_rows / rotate_columns: out(X) = in(X^galois_elt) mod (X^N + 1), i.e. coefficient i moves to position
 * i * galois_elt mod 2N, negated when that position is >= N.  galois_elt odd, in [1, 2N).  out must not alias in.
 * Follow with pf_key_switch (target = the permuted c1, key = the Galois key of galois_elt) to return to the original
 * secret key.  HBM-bound, 16 bytes per coefficient (contiguous writes, permuted reads). */
pf_status pf_apply_galois(pf_ctx *ctx, const uint64_t *in, uint64_t *out, size_t n_limb_polys, uint32_t galois_elt, pf_stream stream);
/* util::negacyclic_shift_poly_coeffmod: out = in * X^exponent mod (X^N + 1) on coefficient-form limb-polynomials,
 * exponent in [0, 2N) (X^-s = X^(2N - s)); coefficient i moves to i + exponent mod 2N, negated when that is >= N.
 * out must not alias in.  HBM-bound, 16 bytes per coefficient. */
pf_status pf_poly_mul_monomial(pf_ctx *ctx, const uint64_t *in, uint64_t *out, size_t n_limb_polys, uint32_t exponent, pf_stream stream);
/* sum = a + b and diff = (a - b) * X^exponent in one pass (the butterfly of SealPIR's query expansion: add_inplace,
 * sub_inplace and multiply_power_of_X).  sum may alias a or b; diff must alias nothing.  32 bytes per coefficient. */
pf_status pf_poly_addsub_monomial(pf_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *sum, uint64_t *diff, size_t n_limb_polys,
                                  uint32_t exponent, pf_stream stream);
/* The same permutation over B ciphertexts [B][2][L][N] in ONE launch, laid out for the key switch that follows
 * (the prologue of Evaluator::apply_galois_inplace): ct_out[b][0] = tau(ct_in[b][0]), ct_out[b][1] = 0,
 * target[b] = tau(ct_in[b][1]) ([B][L][N]); then pf_key_switch(key ring, target, galois key, ct_out, B). */
pf_status pf_apply_galois_ct(pf_ctx *ctx, const uint64_t *ct_in, uint64_t *ct_out, uint64_t *target, size_t B, uint32_t galois_elt,
                             pf_stream stream);

/* Evaluator::switch_key_inplace (what relinearize_inplace / rotate_rows / apply_galois run), BFV form: the
 * polynomial `target` (coefficient form) is re-encrypted under the secret key through the key-switching key and
 * ADDED into both components of `ct`.  `ctx` must hold the KEY moduli: the data primes q_0..q_{D-1} followed by
 * the special prime P (SEAL's last coeff modulus); the ciphertexts live at the data level (D limbs).
 *   target [B][D][N]        limb I canonical mod q_I
 *   ksk    [D][2][D+1][N]   digit I's key as SEAL stores it (PublicKey data: 2 polys x K limbs), NTT form
 *   ct     [B][2][D][N]     coefficient form, updated in place
 * RNS digit decomposition -> D*(D+1) forward NTTs -> 128-bit lazy multiply-accumulate with the key ->
 * inverse NTTs -> division by P with rounding.  Uses an internal workspace (4096-8192 digit transforms per round: 32 ciphertexts at config 5 = 2.2 GB, 192 at N = 8192 = 0.4 GB).
 * After pf_key_switch_reserve(ctx, B') with B' >= B the call neither allocates nor synchronises nor reads the environment (it can be
 * captured into a hipGraph); without it the workspace is grown on first use.
 * At N = 32768 with every modulus below 2^56 (SEAL's defaults: config 5) the digit transforms run as two passes with the key products
 * accumulated in registers inside the second (prefhetch_amd/csrc/ks_split.hpp). */
pf_status pf_key_switch(pf_ctx *ctx, const uint64_t *target, const uint64_t *ksk, uint64_t *ct, size_t B, pf_stream stream);

/* Allocates the key-switching workspace for calls of up to B polynomials (may free and re-allocate: not inside graph capture). */
pf_status pf_key_switch_reserve(pf_ctx *ctx, size_t B);

/* Same, with every input ciphertext multiplied by `fanout` consecutive plaintexts: out[b] = ct[b / fanout] x pt[b],
 * b < B; ct holds ceil(B / fanout) ciphertexts, pt_ntt B plaintexts, out B ciphertexts (must not alias ct when
 * fanout > 1).  This is the shape of the encrypted precise search: one query ciphertext against
 * ceil(COARSE_PROBE * d / N) packed blocks of candidate rows. */
pf_status pf_ct_pt_mul_fanout(pf_ctx *ctx, const uint64_t *ct, const uint64_t *pt_ntt, uint64_t *out, size_t B, uint32_t fanout,
                              int flags, pf_stream stream);

/* Server side of the encrypted precise search (the step the reference's TODOs at include/client/client_lib.h:14,28-30
 * and the "preciseQuery" fields of src/server/controllers/Query.cc:37,80 leave in the clear): packs base rows into
 * plaintext polynomials such that, for a query polynomial q(X) = sum_i q_i X^i (i < d), coefficient d*j of
 * q(X) * p(X) mod (X^N + 1) is the inner product <q, row_j>:
 *     p(X) = sum_{j < rows_per_poly} sum_{i < d} x[ids[p][j]][i] * X^(d*j - i),     X^(-i) = -X^(N-i).
 * Row values are rounded to the nearest integer (|v| < 2^24); out[p][l][c] is the coefficient's canonical residue
 * mod q_l, coefficient form -- follow with pf_ntt_forward to obtain the NTT-form plaintexts pf_ct_pt_mul takes.
 * ids [n_polys][rows_per_poly] (device); an id < 0 or >= nb contributes a zero row.  rows_per_poly <= N / d.
 * No allocation, no synchronisation. */
pf_status pf_pack_rows(pf_ctx *ctx, const pf_flat *idx, const int64_t *ids, size_t n_polys, uint32_t rows_per_poly,
                       uint64_t *out, pf_stream stream);

/* pf_pack_rows followed by pf_ntt_forward in ONE kernel: the block is packed from the base rows straight into the
 * registers of the forward transform, so the coefficient-form plaintext never exists in memory (1 write per
 * coefficient instead of 2 writes + 1 read).  out [n_polys][L][N] in NTT form, bit for bit what the two calls give. */
pf_status pf_pack_rows_ntt(pf_ctx *ctx, const pf_flat *idx, const int64_t *ids, size_t n_polys, uint32_t rows_per_poly,
                           uint64_t *out, pf_stream stream);

/* Sums of ciphertext x plaintext products, operands and results in NTT form: with groups of `chunk` consecutive plaintexts,
 *     out[g] = sum_{p in group g} ct_ntt[p mod n_ct] . pt_ntt[p],     g < ceil(n_pt / chunk),   chunk divides n_ct.
 * ct_ntt [n_ct][2][L][N], pt_ntt [n_pt][L][N], out [ceil(n_pt / chunk)][2][L][N] (must not alias the operands).  One pass over
 * the operands with 128-bit lazy sums (PF_ERR_UNSUPPORTED when chunk * q^2 could reach 2^128): Evaluator::multiply_plain +
 * add_many of SealPIR's answer step without the products ever existing in memory.  No allocation, no synchronisation. */
pf_status pf_ct_pt_dot(pf_ctx *ctx, const uint64_t *ct_ntt, size_t n_ct, const uint64_t *pt_ntt, size_t n_pt, size_t chunk, uint64_t *out,
                       pf_stream stream);

/* The server side of the encrypted precise search in ONE kernel: out[b] = ct[b / fanout] x pack(ids[b]), b < B, with the
 * plaintext of product b packed from the base rows (pf_pack_rows), transformed and multiplied into both components of
 * its ciphertext inside one workgroup per (product, limb) -- the plaintexts never exist in memory, in either form.
 * ct_ntt [ceil(B / fanout)][2][L][N] in NTT form (pf_ntt_forward_to of the query ciphertexts), ids [B][rows_per_poly]
 * (device), out [B][2][L][N] coefficient form, must not alias ct_ntt.  Bit for bit pf_pack_rows_ntt followed by
 * pf_ct_pt_mul_fanout(..., PF_CTPT_IN_NTT).  N <= 16384.  No allocation, no synchronisation. */
pf_status pf_ct_rows_mul(pf_ctx *ctx, const uint64_t *ct_ntt, const pf_flat *idx, const int64_t *ids, size_t B, uint32_t rows_per_poly,
                         uint32_t fanout, uint64_t *out, pf_stream stream);

/* ---- plaintext distance stages ---------------------------------------------------------------- */
/* faiss::IndexFlatL2(d) + add(nb, xb): copies the base matrix [nb][d] fp32 (host or device pointer)
 * into HBM and precomputes row norms.  Blocking. */
pf_status pf_flat_create(pf_flat **idx, int device, const float *xb_host_or_device, size_t nb, uint32_t d);
pf_status pf_flat_destroy(pf_flat *idx);
pf_status pf_flat_info(const pf_flat *idx, size_t *nb, uint32_t *d);
/* faiss::IndexFlatL2::search(nq, xq, k, D, I): squared L2, ascending; ties -> smaller id;
 * entries beyond nb are (+inf, -1).  xq [nq][d], D [nq][k] fp32, I [nq][k] int64.  k <= 1024. */
pf_status pf_flat_search(pf_flat *idx, const float *xq, size_t nq, uint32_t k, float *D, int64_t *I, pf_stream stream);
/* Server::preciseSearch (/root/reference/src/server/server_lib.cpp:140-167): D[i][j] = squared L2
 * between xq[i] and base row ids[i][j], accumulated exactly as the reference does
 * (float += pow(float diff, 2), i.e. through double per step).  ids [nq][c] int64, D [nq][c]. */
pf_status pf_l2_gathered(pf_flat *idx, const float *xq, const int64_t *ids, size_t nq, uint32_t c, float *D, pf_stream stream);
/* Server::preciseVectorPIR (server_lib.cpp:169-196) and Server::retrieve_centroids (:101-109):
 * out[i] = base row ids[i].  ids [n_ids] int64, out [n_ids][d]. */
pf_status pf_gather_rows(pf_flat *idx, const int64_t *ids, size_t n_ids, float *out, pf_stream stream);

/* ---- IVF-PQ coarse stage (Server::coarseSearch, /root/reference/src/server/server_lib.cpp:111-138) ------------- */
/* faiss::IndexIVFPQ(quantizer, d, nlist, M, 8) with trained tables: coarse centroids [nlist][d] and PQ codebooks
 * [M][256][d/M] (host pointers, copied).  by_residual semantics: codes quantise x - centroid. */
typedef struct pf_ivfpq pf_ivfpq;
pf_status pf_ivfpq_create(pf_ivfpq **idx, int device, uint32_t d, uint32_t nlist, uint32_t M, const float *centroids_host,
                          const float *codebooks_host);
pf_status pf_ivfpq_destroy(pf_ivfpq *idx);
/* IndexIVFPQ::add after assignment/encoding: vector i goes to list list_ids_host[i] with code codes_host[i][M], label ids_host[i]. */
pf_status pf_ivfpq_add_encoded(pf_ivfpq *idx, size_t n, const int64_t *list_ids_host, const uint8_t *codes_host, const int64_t *ids_host);
pf_status pf_ivfpq_info(const pf_ivfpq *idx, uint32_t *d, uint32_t *nlist, uint32_t *M, size_t *ntotal, uint64_t *list_sizes_host);
pf_status pf_ivfpq_get_list(const pf_ivfpq *idx, uint32_t list, uint8_t *codes_host, int64_t *ids_host);
/* IndexIVFPQ::search_encrypted (PreFHEtch-faiss fork; semantics from the call site and its consumer,
 * src/client/client_lib.cpp:122-156): for query q and each of its nprobe GIVEN lists, in the given order, the asymmetric
 * PQ distance and label of EVERY stored vector, unsorted; a query's results are contiguous, queries back to back;
 * list_sizes_host[q] = number of results of query q.  xq [nq][d] device, probe_host [nq][nprobe] host (-1 = skip),
 * D / I device buffers of `capacity` entries.  The probe ids and output offsets travel through pinned staging buffers owned by
 * the index (an event, not a stream synchronisation, guards their reuse by the next call); probe_host may be reused on return. */
pf_status pf_ivfpq_search_lists(pf_ivfpq *idx, const float *xq, const int64_t *probe_host, size_t nq, uint32_t nprobe, float *D,
                                int64_t *I, size_t capacity, uint64_t *list_sizes_host, pf_stream stream);

/* 16-bit operands for the pre-filter (any row length, any number of queries; rows that are not whole k-steps of the matrix instruction are
 * padded with zeros in the image).  pf_flat_create keeps a bf16 image of the
 * base (nearest-even) and checks on the device, value by value, whether it IS the base: every value an integer of magnitude
 * <= 256 (SIFT, the reference's dataset -- 8-bit values, /root/reference/include/common/client_server_utils.h:10-20 --
 * qualifies); query tiles are checked the same way at every search.
 *   exact operands    every product and every partial sum is an integer below 2^24: the bf16 matrix instruction with fp32
 *                     accumulation evaluates the distance test itself, and a survivor's distance is recomputed from the
 *                     16-bit rows -- bit for bit the fp32 chain's number.
 *   inexact operands  the bf16 tiles run as a CONSERVATIVE FILTER (thresholds lowered by the bound on the operands' rounding,
 *                     2^-8 (|x|^2 + max |y|^2)); every survivor's distance is the k-ordered fp32 chain over the fp32 rows.
 *   rows above 256    batches of more than 64 queries: bf16 tiles with both operands staged through LDS in k-slabs, always as a
 *                     conservative filter (margin 2.1 x 2^-8 of |x|^2 + |y|^2, or the accumulations' rounding alone when every value
 *                     on both sides is exactly representable); every candidate's distance is the fp32 chain's.
 * Either way (D, I) are what the fp32-operand loop returns, bit for bit; the caller never has to know.
 * mode 1 = on (default), 0 = fp32 operands always, -1 = query only.
 * *active_out (may be NULL): 2 = on with an exactly representable base (rows up to 256), 1 = on as a filter (an inexact base, or rows
 * above 256), 0 = off. */
pf_status pf_flat_exact16(pf_flat *idx, int mode, int *active_out);

/* 8-bit integer operands for the pre-filter.  When EVERY value of the base is an integer in [0, 255] (SIFT: the reference's
 * dataset, /root/reference/include/common/client_server_utils.h:10-20) and d is at most 128, pf_flat_create also
 * keeps int8 images (value - 128; row-major, and in the matrix instruction's fragment order) and the filtered chunks of a query tile that is
 * 8-bit as well (checked on the device at every search) run v_mfma_i32_16x16x64_i8 straight from memory: exact integer accumulators,
 * integer thresholds, no LDS staging.  Survivors are evaluated exactly (v_dot4_u32_u8): (D, I) are bit-identical to the fp32-operand loop's.
 * mode 1 on (default), 0 off (the bf16 tiles then run on the same data), -1 query; active_out: 1 when an image exists and is on. */
pf_status pf_flat_operands8(pf_flat *idx, int mode, int *active_out);

/* pf_flat_search that also (or only: D and I may then be NULL) writes the exchange record of the multi-GPU gather,
 * packed[q][i] = { uint32 id low word, uint32 id high word, uint32 distance bits }, 12 bytes per result, straight from
 * the selection kernel -- the block a rank contributes to the ONE all-gather of SURVEY.md section 8(e) needs no packing
 * pass.  The reference is single-device (/root/reference/src/server/server_lib.cpp:48-53) and has no counterpart. */
pf_status pf_flat_search_packed(pf_flat *idx, const float *xq, size_t nq, uint32_t k, float *D, int64_t *I, uint32_t *packed,
                                pf_stream stream);

/* Bytes of scratch pf_flat_search needs for (nq, k); the library grows an internal workspace on
 * first use (outside graph capture) -- call pf_flat_reserve up front to keep searches allocation-free. */
pf_status pf_flat_reserve(pf_flat *idx, size_t nq_max, uint32_t k_max);

/* ---- device group: one process, G devices ------------------------------------------------ */
/* The reference server is one process on one device (/root/reference/src/server/server_lib.cpp:48-53: one Drogon
 * listener, one index).  A group extends that process to the G GPUs of a node: queries / ciphertexts are independent
 * units, so the batch is split contiguously over the members (member r takes units [r*n_local, (r+1)*n_local)), the
 * RNS tables and the fp32 base matrix are REPLICATED on every member, and the only exchange step is ONE all-gather of the
 * packed per-member top-k blocks (RCCL over xGMI).  Every member has its own host thread and HIP stream; a group call
 * hands the same job to all member threads, which enqueue on their streams and return -- nothing synchronises unless
 * documented.  `exchange`: PF_MULTI_AUTO picks RCCL (ncclCommInitAll + ncclAllGather, librccl loaded on first use) when
 * the devices are distinct and direct device-to-device copies when a device is listed more than once (RCCL refuses
 * that; it is how a one-GPU machine rehearses the control flow).  PF_MULTI_PEER_COPY forces the copies
 * (hipMemcpyPeerAsync pushes ordered by events), PF_MULTI_RCCL insists on RCCL. */
typedef struct pf_multi pf_multi;
enum { PF_MULTI_AUTO = 0, PF_MULTI_RCCL = 1, PF_MULTI_PEER_COPY = 2 };
pf_status pf_multi_create(pf_multi **grp, const int *devices_host, int n_devices, int exchange);
pf_status pf_multi_destroy(pf_multi *grp);
/* exchange_out: PF_MULTI_RCCL or PF_MULTI_PEER_COPY, whichever the group uses.  Any out pointer may be NULL. */
pf_status pf_multi_info(const pf_multi *grp, int *n_devices, int *devices_out_host, int *exchange_out);
/* pf_ctx_create / pf_flat_create on every member (tables and base matrix replicated; xb_host is a HOST pointer).
 * Blocking.  A second call replaces the first. */
pf_status pf_multi_ring(pf_multi *grp, uint32_t N, uint32_t L, const uint64_t *moduli_host);
pf_status pf_multi_flat(pf_multi *grp, const float *xb_host, size_t nb, uint32_t d);
pf_status pf_multi_reserve(pf_multi *grp, size_t nq_local_max, uint32_t k_max);
/* Member r's handles, for callers that own per-device buffers (any out pointer may be NULL; ctx / flat are NULL until
 * pf_multi_ring / pf_multi_flat ran).  The handles belong to the group. */
pf_status pf_multi_member(pf_multi *grp, int rank, int *device, pf_stream *stream, pf_ctx **ctx, pf_flat **flat);
/* Device-resident search: xq_dev[r] = member r's shard [nq_local][d] on its device, gathered_dev[r] = member r's copy of
 * the result [G * nq_local][k][3] uint32 (record layout of pf_flat_search_packed) on its device.  Member r's selection
 * kernel writes its block in place at gathered_dev[r] + r * nq_local * k * 3, then ONE all-gather completes every copy.
 * Enqueues and returns (pf_multi_synchronize waits). */
pf_status pf_multi_flat_search(pf_multi *grp, const float *const *xq_dev, size_t nq_local, uint32_t k, uint32_t *const *gathered_dev);
/* pf_ct_pt_mul on every member's shard: ct_dev[r] [B_local][2][L][N], pt_dev[r] [pt_count][L][N] (pt_count = B_local,
 * or 1 to broadcast), out_dev[r] [B_local][2][L][N].  No exchange: the result ciphertexts go back to their clients. */
pf_status pf_multi_ct_pt_mul(pf_multi *grp, const uint64_t *const *ct_dev, const uint64_t *const *pt_dev, size_t pt_count,
                             uint64_t *const *out_dev, size_t B_local, int flags);
pf_status pf_multi_synchronize(pf_multi *grp);
/* What a host server calls (the shape of Server::preciseSearch's caller, /root/reference/src/server/controllers/Query.cc:65-98,
 * at batch size): xq_host [nq][d] in host memory is sharded over the members (earlier members take the remainder),
 * uploaded, searched, gathered, and member 0's copy comes back as D_host [nq][k] / I_host [nq][k].  Blocking. */
pf_status pf_multi_flat_search_host(pf_multi *grp, const float *xq_host, size_t nq, uint32_t k, float *D_host, int64_t *I_host);

#ifdef __cplusplus
}
#endif
#endif /* PREFHETCH_HIP_H */
