// Instruction-rate microbenchmark for gfx950: decides which modular-multiply formulation the NTT
// butterflies use ("micro-benchmark first").  Each kernel runs ITERS dependent-free
// (8 independent chains per lane) instances of one op; reports wave-instruction issue cycles per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_rates.hip -o tools/ubench_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int CH = 8;

template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint64_t *out, uint64_t seed) {
    uint64_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[CH]; double d[CH];
    double q = (double)(0xFFFFFFFC001ull), w = (double)(0x123456789ABull + seed), wq = w / q;
    uint64_t qi = 0xFFFFFFFC001ull, wi = 0x123456789ABull + seed, wqi = (uint64_t)(((unsigned __int128)wi << 64) / qi);
#pragma unroll
    for (int c = 0; c < CH; c++) { a[c] = tid * 0x9E3779B97F4A7C15ull + c * 77 + seed; d[c] = (double)(a[c] >> 20); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if constexpr (OP == 0) d[c] = __builtin_fma(d[c], wq, w);                       // v_fma_f64
            else if constexpr (OP == 1) d[c] = d[c] * wq;                                    // v_mul_f64
            else if constexpr (OP == 2) d[c] = d[c] + w;                                     // v_add_f64
            else if constexpr (OP == 3) d[c] = __builtin_rint(d[c]) + 0.5;                   // v_rndne_f64 (+add)
            else if constexpr (OP == 4) a[c] = (uint64_t)(uint32_t)a[c] * (uint32_t)wi + a[c];   // v_mad_u64_u32
            else if constexpr (OP == 5) a[c] = (uint32_t)a[c] * (uint32_t)wi;                // v_mul_lo_u32
            else if constexpr (OP == 6) a[c] = __umulhi((uint32_t)a[c], (uint32_t)wi) + 1u;  // v_mul_hi_u32
            else if constexpr (OP == 7) a[c] = a[c] + wi;                                    // 64-bit add
            else if constexpr (OP == 8) a[c] = __umul64hi(a[c], wqi) | 1;                    // full mulhi64
            else if constexpr (OP == 9) a[c] = a[c] * wi + 1;                                // mullo64
            else if constexpr (OP == 10) {                                                   // fp64 lazy butterfly half: T = y*w mod q (6 ops)
                double y = d[c];
                double h = y * w, l = __builtin_fma(y, w, -h);
                double cq = __builtin_rint(y * wq);
                double t = __builtin_fma(-cq, q, h) + l;
                d[c] = t;
            } else if constexpr (OP == 11) {                                                 // integer Shoup lazy mul: [0,2q)
                uint64_t y = a[c];
                uint64_t hi = __umul64hi(y, wqi);
                a[c] = y * wi - hi * qi;
            } else if constexpr (OP == 12) {                                                 // fp64 magic-round variant
                double y = d[c];
                double h = y * w, l = __builtin_fma(y, w, -h);
                double cq = __builtin_fma(y, wq, 6755399441055744.0) - 6755399441055744.0;
                double t = __builtin_fma(-cq, q, h) + l;
                d[c] = t;
            } else if constexpr (OP == 13) d[c] = __builtin_floor(d[c]) + 0.5;               // v_floor_f64 (+add)
            else if constexpr (OP == 14) a[c] = __umul24((uint32_t)a[c] & 0xFFFFFF, (uint32_t)wi & 0xFFFFFF) + 1u;  // v_mul_u32_u24/mad
            else if constexpr (OP == 15) { float f = __uint_as_float((uint32_t)a[c]); f = __builtin_fmaf(f, 1.0001f, 0.5f); a[c] = __float_as_uint(f); } // v_fma_f32
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) s += a[c] + (uint64_t)d[c];
    if (s == 0x1234567) out[tid] = s;   // practically never; keeps the chains alive
}

template <int OP>
int run(const char *name, int n_ops_per_inst, uint64_t *dout) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 8;   // 8 blocks of 256 thr per CU = 32 waves/CU = 8 waves/SIMD
    k_rate<OP><<<blocks, 256>>>(dout, 1); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; r++) k_rate<OP><<<blocks, 256>>>(dout, r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    // wave-instructions per SIMD: waves/SIMD = blocks*4 waves /(256 CU*4 SIMD) ; each wave ITERS*CH insts
    double waves_per_simd = blocks * 4.0 / (256 * 4);
    double insts = waves_per_simd * ITERS * CH;
    double ns_per_inst = ms * 1e6 / insts;
    printf("%-28s %8.3f ms  %7.2f ns/wave-inst/SIMD  = %6.1f cyc@2.4GHz  (%d machine ops -> %5.1f cyc/op)\n", name, ms, ns_per_inst,
           ns_per_inst * 2.4, n_ops_per_inst, ns_per_inst * 2.4 / n_ops_per_inst);
    return 0;
}

int main() {
    uint64_t *dout; CK(hipMalloc(&dout, 256 * 8 * 256 * 8));
    run<15>("v_fma_f32", 1, dout);
    run<0>("v_fma_f64", 1, dout);
    run<1>("v_mul_f64", 1, dout);
    run<2>("v_add_f64", 1, dout);
    run<3>("v_rndne_f64 + add", 2, dout);
    run<13>("v_floor_f64 + add", 2, dout);
    run<4>("v_mad_u64_u32", 1, dout);
    run<5>("v_mul_lo_u32", 1, dout);
    run<6>("v_mul_hi_u32 + add", 2, dout);
    run<14>("v_mul_u32_u24 + add", 2, dout);
    run<7>("u64 add (2 ops)", 2, dout);
    run<8>("umul64hi", 1, dout);
    run<9>("mullo64 + 1", 1, dout);
    run<10>("fp64 mulmod (rndne) 6op", 6, dout);
    run<12>("fp64 mulmod (magic) 6op", 6, dout);
    run<11>("int Shoup lazy mulmod", 1, dout);
    return 0;
}
