// Probe: practical HBM rates on this box for the access mixes of the level-0 layers: read-only, write-only, copy (1:1) and a
// 4:3 read:write mix (conv 32->32 with its 1.33x halo re-reads).  16 B per lane, grid-stride, buffers of 1 GiB (>> 256 MiB MALL).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read(const float4* __restrict__ a, float4* __restrict__ out, size_t n) {
    float4 s = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (s.x == 123.456f) out[0] = s;
}
__global__ void k_write(float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = float4{1.f, 2.f, 3.f, (float)i};
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// reads 4 units, writes 3 (every 4th element of the index space is read but not written)
__global__ void k_mix43(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i];
        if ((i >> 6) & 3) b[i] = v; else if (v.x == 123.456f) b[i] = v;
    }
}
int main() {
    const size_t n = (size_t)1 << 26;     // 64 Mi float4 = 1 GiB
    float4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
    hipMemset(a, 1, n * 16); hipMemset(b, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 4096, 16384}) {
        for (int k = 0; k < 4; ++k) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (k == 0) k_read<<<grid, 256>>>(a, b, n); else if (k == 1) k_write<<<grid, 256>>>(b, n);
                else if (k == 2) k_copy<<<grid, 256>>>(a, b, n); else k_mix43<<<grid, 256>>>(a, b, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double bytes = (k == 0 || k == 1) ? n * 16.0 : (k == 2 ? n * 32.0 : n * 16.0 * 1.75);
            printf("grid %5d %-6s %.3f ms  %.2f TB/s\n", grid, k == 0 ? "read" : k == 1 ? "write" : k == 2 ? "copy" : "mix4:3", best, bytes / best / 1e9);
        }
    }
    return 0;
}
