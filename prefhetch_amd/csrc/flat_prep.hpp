// flat_prep.hpp -- row norms, the 16-bit / 8-bit operand images and their on-device eligibility checks (k_row_norms, k_rows_prep, k_aux_margin)
// (part of the pre-filter translation unit pf_flat.hip: included there, in order; gfx950 only)
#pragma once
#include "flat_common.hpp"

namespace pf {

// row norms, fp32 fma chain in index order
__global__ void __launch_bounds__(256) k_row_norms(const float *__restrict__ x, size_t n, uint32_t d, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *r = x + i * d;
    float acc = 0.f;
    for (uint32_t k = 0; k < d; ++k) acc = fmaf(r[k], r[k], acc);
    out[i] = acc;
}

// row norms (fp32 fma chain in index order) + 16-bit image + eligibility.  A workgroup of 64 threads takes ROWS rows: the rows
// are read coalesced (and converted / checked) by all lanes into LDS, then lane r chains row r's norm out of LDS (row pitch
// d + 1 floats: conflict-free).  d <= MAXD <= PREP_MAX_D; wider rows take the one-thread-per-row kernel below.  ROWS = 64 (32) for the
// base (millions of rows), 4 for a batch of queries (1024 rows in 64 rows per workgroup were 16 workgroups and 37 us).
// The image has `pitch16` 16-bit words per row.  With `aux` (the base), words d .. d+7 of a row hold the column's half of the
// threshold term the bf16 tiles feed to the matrix pipe as a ninth k-step: (-b0, -b1, -b2, 1, 1, 1, 0, 0), b0 + b1 + b2 =
// |y|^2 / 2 exactly (bf16_split3); the query's half is built by the tile kernel (k_l2_tile16).
constexpr uint32_t PREP_MAX_D = 256;                            // rows up to 128 values: 64 per workgroup; up to 256: 32 (the staging tile stays at 33 KiB)
template <uint32_t ROWS, uint32_t MAXD>
__global__ void __launch_bounds__(64) k_rows_prep(const float *__restrict__ x, size_t n, uint32_t d, float *__restrict__ norms,
                                                  uint16_t *__restrict__ x16, uint32_t pitch16, bool aux, uint32_t *__restrict__ inexact,
                                                  uint32_t rows_per_flag, int8_t *__restrict__ x8 = nullptr, uint32_t pitch8 = 0,
                                                  int8_t *__restrict__ x8f = nullptr, int *__restrict__ c0f = nullptr, int *__restrict__ sx8 = nullptr,
                                                  uint32_t dp = 0) {
    // dp (0: d): the row length of the IMAGES -- d padded with zeros (value 0: the byte -128 in the 8-bit images) to whole k-steps of the matrix
    // instructions, so that every row length takes the tile path.  A zero contributes nothing to a dot product or a norm, and the integer algebra of
    // the 8-bit filter holds for the padded vectors as it does for any others (its constants are taken at dp).
    dp = dp ? dp : d;
    __shared__ float tile[ROWS * (MAXD + 1)];
    const size_t r0 = (size_t)blockIdx.x * ROWS;
    const uint32_t rows = (uint32_t)(n - r0 < ROWS ? n - r0 : ROWS), total = rows * d, lane = threadIdx.x;
    const float *src = x + r0 * d;
    uint32_t bad = 0, bad8 = 0;                                   // bit (row / rows_per_flag within this block's span) ... kept per lane
    for (uint32_t e = lane; e < total; e += 64) {
        const float v = src[e];
        const uint32_t r = e / d, k = e - r * d;
        tile[r * (d + 1) + k] = v;
        if (x16) x16[(r0 + r) * pitch16 + k] = bf16_rne(v);                              // exact when the value passes; nearest otherwise
        const uint32_t fbit = 1u << (rows_per_flag ? ((r0 + r) / rows_per_flag - r0 / rows_per_flag) : 0);
        if (inexact && !bf16_exact(v)) bad |= fbit;
        if (x8) {                                                                        // 8-bit data: value - 128 as int8 (meaningless, and flagged, otherwise)
            const bool ok8 = v == rintf(v) && v >= 0.f && v <= 255.f;
            const int8_t b8 = (int8_t)(ok8 ? (int)v - 128 : 0);
            x8[(r0 + r) * (size_t)pitch8 + k] = b8;
            if (x8f) x8f[frag8_offset(r0 + r, k, frag8_ksteps(dp))] = b8;                     // the same byte in the streamed walk's operand order (flat_common.hpp)
            if (!ok8) bad8 |= fbit;
        }
    }
    if (dp > d) {                                                 // the padding of this block's rows (the queries' images live in a reused workspace: written every time)
        const uint32_t pad = dp - d;
        for (uint32_t e = lane; e < rows * pad; e += 64) {
            const uint32_t r = e / pad, k = d + (e - r * pad);
            if (x16) x16[(r0 + r) * pitch16 + k] = 0;
            if (x8) {
                x8[(r0 + r) * (size_t)pitch8 + k] = (int8_t)-128;
                if (x8f) x8f[frag8_offset(r0 + r, k, frag8_ksteps(dp))] = (int8_t)-128;
            }
        }
    }
    if (bad | bad8) {                                             // a block of <= 64 rows touches at most two flags (rows_per_flag >= 64) or one
        const uint32_t w0 = (bad & 1u) | ((bad8 & 1u) << 2), w1 = ((bad >> 1) & 1u) | (((bad8 >> 1) & 1u) << 2);
        if (w0) atomicOr(&inexact[rows_per_flag ? r0 / rows_per_flag : 0], w0);
        if (w1) atomicOr(&inexact[r0 / rows_per_flag + 1], w1);
    }
    __syncthreads();
    if (lane < rows) {
        const float *row = tile + lane * (d + 1);
        float acc = 0.f;
        int sum8 = -128 * (int)(dp - d);                             // sum (value - 128) over the image row, its zero padding included (8-bit data: read below)
#pragma unroll 8                                                       // (the LDS reads of eight steps in flight ahead of the serial chain)
        for (uint32_t k = 0; k < d; ++k) { acc = fmaf(row[k], row[k], acc); sum8 += (int)row[k] - 128; }
        norms[r0 + lane] = acc;
        if (x16 && aux) {
            uint32_t b[3];
            bf16_split3(0.5f * acc, b);
            u32x4 w;
            w[0] = (b[0] ^ BF16_SIGN) | ((b[1] ^ BF16_SIGN) << 16);
            w[1] = (b[2] ^ BF16_SIGN) | (BF16_ONE << 16);
            w[2] = BF16_ONE | (BF16_ONE << 16);
            w[3] = 0;
            *reinterpret_cast<u32x4 *>(x16 + (r0 + lane) * pitch16 + dp) = w;       // 16-byte aligned: dp and pitch16 are multiples of 8
        }
        if (x8 && aux) {                                         // the column's half of the integer threshold (tile16_walk): c0 = -floor(C / 2), C = |y|^2 - 256 sum (y - 128)
            const int sy = sum8;
            const int Cc = (int)acc - 256 * sy;
            u32x4 w;
            w[0] = (uint32_t)(-(Cc >> 1)); w[1] = (uint32_t)sy; w[2] = 0; w[3] = 0;      // (sum y' for the unfiltered launch, which forms distances)
            *reinterpret_cast<u32x4 *>(x8 + (r0 + lane) * (size_t)pitch8 + dp) = w;  // 16-byte aligned: dp and pitch8 are multiples of 16
            if (c0f) c0f[frag8_c0_index(r0 + lane)] = Cc;                               // C itself: the walk halves it, the flush forms distances with it
        }
        if (x8 && !aux && sx8) {                                 // queries: sum (x - 128), the row half of the integer threshold needs it (tile8_walk)
            sx8[r0 + lane] = sum8;
        }
    }
}

// Inexact base: the column's share of the filter margin goes into its threshold words, b0 + b1 + b2 = |y|^2 (1/2 - 1.05 x 2^-8)
// (k_l2_tile16: the bound on the operands' rounding is (2^-8 + 2^-17)(|x|^2 + |y|^2), priced per column -- a base whose rows
// differ widely in length would otherwise pay the longest row's margin in every column).
constexpr float BF16_MARGIN = 1.05f * 0x1p-8f;
__global__ void __launch_bounds__(256) k_aux_margin(uint16_t *__restrict__ x16, const float *__restrict__ norms, size_t n, uint32_t d /* image row length */, uint32_t pitch16) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    uint32_t b[3];
    bf16_split3(norms[r] * (0.5f - BF16_MARGIN), b);
    u32x4 w;
    w[0] = (b[0] ^ BF16_SIGN) | ((b[1] ^ BF16_SIGN) << 16);
    w[1] = (b[2] ^ BF16_SIGN) | (BF16_ONE << 16);
    w[2] = BF16_ONE | (BF16_ONE << 16);
    w[3] = 0;
    *reinterpret_cast<u32x4 *>(x16 + r * pitch16 + d) = w;
}

}  // namespace pf
