// Does a 16-byte access at a 64-byte (or 128-byte) lane stride cost more than the same bytes lane-contiguous?
// Every workgroup moves CHUNK bytes per iteration; pattern 0: lane l, access c -> byte (l*16 + c*1024)  [contiguous per instruction]
//                                                   pattern 1: lane l, access c -> byte (l*64 + c*16)    [runs of 64 B per lane]
//                                                   pattern 2: lane l, access c -> byte (l*128 + c*16)   [runs of 128 B per lane, 8 accesses]
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_stride.hip -o tools/ubench_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int PAT, bool STORE>
__global__ void __launch_bounds__(512, 2) k(uint4 *buf, size_t chunks_per_block) {
    // a wave covers 4 KiB (pattern 0, 1: 4 accesses) or 8 KiB (pattern 2: 8 accesses) per step
    constexpr int ACC = PAT == 2 ? 8 : 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 acc = make_uint4(lane, wave, 0, 0);
    for (size_t it = 0; it < chunks_per_block; ++it) {
        char *base = reinterpret_cast<char *>(buf) + ((blockIdx.x * chunks_per_block + it) * 8 + wave) * (size_t)(ACC * 1024);
#pragma unroll
        for (int c = 0; c < ACC; ++c) {
            const size_t off = PAT == 0 ? (size_t)lane * 16 + c * 1024 : PAT == 1 ? (size_t)lane * 64 + c * 16 : (size_t)lane * 128 + c * 16;
            uint4 *p = reinterpret_cast<uint4 *>(base + off);
            if (STORE) *p = acc; else { uint4 v = *p; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
        }
    }
    if (!STORE && acc.x == 0x12345 && acc.w == 77) buf[0] = acc;
}

template <int PAT, bool STORE>
int run(uint4 *buf, size_t bytes, const char *name) {
    constexpr int ACC = PAT == 2 ? 8 : 4;
    const size_t per_block_step = 8 * (size_t)ACC * 1024;
    const unsigned blocks = 256 * 2 * 4;
    const size_t steps = bytes / per_block_step / blocks;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<PAT, STORE><<<blocks, 512>>>(buf, steps); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) k<PAT, STORE><<<blocks, 512>>>(buf, steps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, (double)steps * blocks * per_block_step / ms / 1e6);
    return 0;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    uint4 *buf; CK(hipMalloc((void **)&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    run<0, false>(buf, bytes, "load  16 B/lane, lane-contiguous");
    run<1, false>(buf, bytes, "load  16 B/lane, runs of 64 B per lane");
    run<2, false>(buf, bytes, "load  16 B/lane, runs of 128 B per lane");
    run<0, true>(buf, bytes, "store 16 B/lane, lane-contiguous");
    run<1, true>(buf, bytes, "store 16 B/lane, runs of 64 B per lane");
    run<2, true>(buf, bytes, "store 16 B/lane, runs of 128 B per lane");
    return 0;
}
