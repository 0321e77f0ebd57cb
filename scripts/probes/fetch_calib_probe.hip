// Probe: calibration of rocprofv3's FETCH_SIZE for the access SHAPES of this repo's kernels.
// MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read and
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute".
// Every kernel below reads a buffer of KNOWN size exactly once (1 GiB >> the 256 MiB Infinity Cache, so every byte comes from HBM) in
// one of the shapes the conv kernels use; run it under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (its own pass) and divide:
//     factor(shape) = bytes read / (FETCH_SIZE [KiB] x 1024)
// scripts/pmc_traffic.py applies the factor of a kernel family's dominant read shape instead of a blanket x 2.
//   k_stream16   16 B per lane, fully coalesced, grid-stride                       (the guide's reference shape: factor 2)
//   k_quarter4   NHWC records of 256 B (64 channels fp32); a workgroup walks a tile of 256 pixels chunk by chunk (16 channels =
//                64 B per pixel): FOUR lanes per pixel, 16 B each - a wave instruction covers 16 pixels (round-4 staging of
//                conv3x3_f16x3_qp / conv3x3s2_v2)
//   k_oct2       the same walk with TWO lanes per pixel, each issuing two 16-byte loads (32 pixels per instruction: the round-3
//                staging, still used by conv3x3_upq / conv3x3_upc / conv3x3_f16x3_one)
//   k_record     whole 128-byte records (32 channels): 8 lanes x 16 B per pixel as 4 lanes x 2 loads (conv3x3_res32 / conv3x3_up0)
//   k_dword      4 B per lane, coalesced rows of an NCHW plane (conv3x3_first and the fused first block)
//   k_ldsdma     global_load_lds, 1 KiB per wave instruction, linear (the weight blocks)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void sink(u32x4 v, u32x4* out) { if (v[0] == 0x12345678u && v[3] == 0x9abcdef0u) out[0] = v; }

__global__ void k_stream16(const u32x4* __restrict__ a, u32x4* out, size_t n) {
    u32x4 s = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const u32x4 v = a[i]; s[0] ^= v[0]; s[3] += v[3]; }
    sink(s, out);
}
// records of 256 B; tile = 256 consecutive records; 256 threads; 4 chunks of 64 B per record
__global__ void k_quarter4(const unsigned char* __restrict__ a, u32x4* out, size_t ntiles) {
    u32x4 s = {0, 0, 0, 0};
    const int px = threadIdx.x >> 2, q = threadIdx.x & 3;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x)
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(a + ((t * 256 + px + 64 * it) * 256 + c * 64 + q * 16));
                s[0] ^= v[0]; s[3] += v[3];
            }
    sink(s, out);
}
__global__ void k_oct2(const unsigned char* __restrict__ a, u32x4* out, size_t ntiles) {
    u32x4 s = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int px = 32 * w + (lane & 7) + 8 * (lane >> 4), oct = (lane >> 3) & 1;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x)
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int l = 0; l < 2; ++l) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(a + ((t * 256 + px + 128 * it) * 256 + c * 64 + oct * 32 + l * 16));
                    s[0] ^= v[0]; s[3] += v[3];
                }
    sink(s, out);
}
// records of 128 B; tile = 256 records; lane: pixel (lane & 7) + 8 (lane >> 5), group (lane >> 3) & 3 of 32 B, two 16-byte loads
__global__ void k_record(const unsigned char* __restrict__ a, u32x4* out, size_t ntiles) {
    u32x4 s = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int px = 16 * w + (lane & 7) + 8 * (lane >> 5), sg = (lane >> 3) & 3;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(a + ((t * 256 + px + 64 * it) * 128 + sg * 32 + l * 16));
                s[0] ^= v[0]; s[3] += v[3];
            }
    sink(s, out);
}
__global__ void k_dword(const unsigned* __restrict__ a, u32x4* out, size_t n) {
    unsigned s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s ^= a[i];
    if (s == 0x12345678u) out[0] = u32x4{s, s, s, s};
}
__global__ void k_ldsdma(const unsigned char* __restrict__ a, u32x4* out, size_t npieces) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[4 * 8 * 1024];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned s = 0;
    for (size_t p = (size_t)blockIdx.x * 32; p < npieces; p += (size_t)gridDim.x * 32) {       // 32 pieces of 1 KiB per workgroup round: 8 per wave
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds(a + (p + w * 8 + j) * 1024 + lane * 16, (lds_ptr)(buf + (w * 8 + j) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s ^= *reinterpret_cast<const unsigned*>(buf + (w * 8) * 1024 + lane * 16);
    }
    if (s == 0x12345678u) out[0] = u32x4{s, s, s, s};
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    unsigned char* a; u32x4* out;
    hipMalloc(&a, bytes); hipMalloc(&out, 64);
    {   // non-trivial contents (zero pages could be treated differently somewhere along the path)
        unsigned* h = (unsigned*)malloc(bytes);
        unsigned x = 12345u;
        for (size_t i = 0; i < bytes / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = x; }
        hipMemcpy(a, h, bytes, hipMemcpyHostToDevice);
        free(h);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 4096;
    for (int k = 0; k < 6; ++k) {
        hipEventRecord(e0);
        switch (k) {
            case 0: k_stream16<<<grid, 256>>>((const u32x4*)a, out, bytes / 16); break;
            case 1: k_quarter4<<<grid, 256>>>(a, out, bytes / (256 * 256)); break;
            case 2: k_oct2<<<grid, 256>>>(a, out, bytes / (256 * 256)); break;
            case 3: k_record<<<grid, 256>>>(a, out, bytes / (256 * 128)); break;
            case 4: k_dword<<<grid, 256>>>((const unsigned*)a, out, bytes / 4); break;
            default: k_ldsdma<<<grid, 256>>>(a, out, bytes / 1024); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static const char* nm[6] = {"k_stream16", "k_quarter4", "k_oct2", "k_record", "k_dword", "k_ldsdma"};
        printf("%-10s reads %zu bytes once  %.3f ms  %.2f TB/s\n", nm[k], bytes, ms, bytes / ms / 1e9);
    }
    return 0;
}
