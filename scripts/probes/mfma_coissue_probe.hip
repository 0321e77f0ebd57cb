// Probe: does anything ride in the shadow of v_mfma_f32_32x32x16_f16 on gfx950?  One loop body = 4 independent MFMAs (4 accumulators),
// each followed by K filler instructions of one kind (independent v_fma_f32 / a dependent v_fma_f32 chain / v_cvt_pk_f16_f32 /
// ds_read_b128).  Reports wall-clock ns and cycles (from the event time and the measured clock) per MFMA for K = 0..8, at one and
// two waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_coissue_probe mfma_coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FILL_INDEP  asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(c0), "v"(c1));
template <int KIND, int K>
__device__ __forceinline__ void fillers(float& f0, float& f1, float& f2, float& f3, float& f4, float& f5, float& f6, float& f7,
                                        float c0, float c1, unsigned& p0, half8& l0, const unsigned char* lds) {
    float* f[8] = {&f0, &f1, &f2, &f3, &f4, &f5, &f6, &f7};
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*f[k & 7]) : "v"(c0), "v"(c1));          // independent
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(c0), "v"(c1));                  // dependent chain
        if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p0) : "v"(*f[k & 7]), "v"(c1));
        if (KIND == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l0) : "v"((unsigned)(size_t)lds), "i"(0));
    }
    if (KIND == 3 && K > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int KIND, int K>
__global__ __launch_bounds__(512, 1) void probe(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u;
    __syncthreads();
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc0, acc1, acc2, acc3;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }
    float f0 = lane, f1 = 1.f, f2 = 2.f, f3 = 3.f, f4 = 4.f, f5 = 5.f, f6 = 6.f, f7 = 7.f;
    const float c0 = 0.999f, c1 = 0.001f;
    unsigned p0 = 0; half8 l0 = a;
    const unsigned char* lp = lds + (lane * 80) % 60000;
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        fillers<KIND, K>(f0, f1, f2, f3, f4, f5, f6, f7, c0, c1, p0, l0, lp);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        fillers<KIND, K>(f0, f1, f2, f3, f4, f5, f6, f7, c0, c1, p0, l0, lp);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
        fillers<KIND, K>(f0, f1, f2, f3, f4, f5, f6, f7, c0, c1, p0, l0, lp);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc3, 0, 0, 0);
        fillers<KIND, K>(f0, f1, f2, f3, f4, f5, f6, f7, c0, c1, p0, l0, lp);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (float)p0 + (float)l0[0];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int KIND, int K>
void run(const char* name, int threads, float* d_out, double ghz) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<KIND, K>), dim3(blocks), dim3(threads), 0, 0, d_out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND, K>), dim3(blocks), dim3(threads), 0, 0, d_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 256.0;
    const double ns_per_mfma = ms * 1e6 / (iters * 4.0 * waves_per_simd);      // per MFMA of the SIMD's pipe
    printf("%-12s K=%d threads=%d: %.2f ns per MFMA = %.1f cycles at %.2f GHz\n", name, K, threads, ns_per_mfma, ns_per_mfma * ghz, ghz);
}

template <int KIND> void sweep(const char* name, float* d_out, double ghz) {
    for (int threads : {256, 512}) {
        run<KIND, 0>(name, threads, d_out, ghz); run<KIND, 2>(name, threads, d_out, ghz); run<KIND, 4>(name, threads, d_out, ghz);
        run<KIND, 5>(name, threads, d_out, ghz); run<KIND, 6>(name, threads, d_out, ghz); run<KIND, 8>(name, threads, d_out, ghz);
    }
}

int main() {
    float* d_out; hipMalloc(&d_out, 256 * 512 * 4);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz / 1e6;
    printf("nominal clock %.2f GHz (cycles below assume it; under MFMA load the chip clocks lower)\n", ghz);
    sweep<0>("indep fma", d_out, ghz);
    sweep<1>("dep fma", d_out, ghz);
    sweep<2>("cvt_pk", d_out, ghz);
    sweep<3>("ds_read_b128", d_out, ghz);
    return 0;
}
