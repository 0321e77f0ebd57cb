// Probe: cycles per "chunk step" of the split-fp16 consumer loop (9 taps x [8 ds_read_b128 + 12 v_mfma_f32_32x32x16_f16])
// with no staging at all.  Variant 0: reads then MFMAs per tap (what conv3x3_f16x3 does); variant 1: fragments of tap t+1
// are read while tap t's MFMAs run (double-buffered registers).  Build: hipcc --offload-arch=gfx950 -O3 -o probe mfma_phase_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kRec = 80, P = 340, BN = 64, PW = 34;

template <int VAR>
__global__ __launch_bounds__(256, 2) void probe(float* out, unsigned long long* cyc, int steps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < (P * kRec + 9 * BN * kRec) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u ^ (i * 2654435761u & 0x03ff03ffu);
    __syncthreads();
    unsigned char* sA = smem; unsigned char* sB = smem + P * kRec;
    int abase[2];
    for (int mt = 0; mt < 2; ++mt) { const int m = 64 * w + 32 * mt + r; abase[mt] = ((m >> 5) * PW + (m & 31)) * kRec + 16 * h; }
    const int bbase = r * kRec + 16 * h;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        if (VAR == 0) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = ((tap / 3) * PW + (tap % 3)) * kRec;
                half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) { ah[mt] = *(const half8*)(sA + abase[mt] + toff); al[mt] = *(const half8*)(sA + abase[mt] + toff + 32); }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) { bh[nt] = *(const half8*)(sB + (tap * BN + nt * 32) * kRec + bbase); bl[nt] = *(const half8*)(sB + (tap * BN + nt * 32) * kRec + bbase + 32); }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
            }
        } else {
            half8 fa[2][2][2], fb[2][2][2];
            auto ld = [&](int buf, int tap) {
                const int toff = ((tap / 3) * PW + (tap % 3)) * kRec;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) { fa[buf][mt][0] = *(const half8*)(sA + abase[mt] + toff); fa[buf][mt][1] = *(const half8*)(sA + abase[mt] + toff + 32); }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) { fb[buf][nt][0] = *(const half8*)(sB + (tap * BN + nt * 32) * kRec + bbase); fb[buf][nt][1] = *(const half8*)(sB + (tap * BN + nt * 32) * kRec + bbase + 32); }
            };
            ld(0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int c = tap & 1;
                if (tap + 1 < 9) ld(c ^ 1, tap + 1);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[c][mt][1], fb[c][nt][0], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[c][mt][0], fb[c][nt][1], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[c][mt][0], fb[c][nt][0], acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) s += acc[a][b][i];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int VAR> void run(int grid, size_t pad, const char* name) {
    const int steps = 400; float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)grid * 256 * 4); hipMalloc(&cyc, (size_t)grid * 8);
    const size_t smem = P * kRec + 9 * BN * kRec + pad;
    hipFuncSetAttribute((const void*)probe<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<VAR><<<grid, 256, smem>>>(out, cyc, steps); hipDeviceSynchronize();
    hipEventRecord(e0); probe<VAR><<<grid, 256, smem>>>(out, cyc, steps); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid); hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    const double flops = (double)grid * 4 * steps * 108 * 32768.0;
    printf("%-34s grid %4d: %.0f cycles/step (floor 3456), %.2f ms, %.0f TFLOP/s fp16-MFMA, implied clock %.2f GHz\n", name, grid, avg / steps, ms,
           flops / (ms * 1e-3) / 1e12, avg / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>(256, 40000, "reads-then-MFMA, 1 WG/CU");
    run<1>(256, 40000, "pipelined frags,  1 WG/CU");
    run<0>(512, 0, "reads-then-MFMA, 2 WG/CU");
    run<1>(512, 0, "pipelined frags,  2 WG/CU");
    return 0;
}
