// Probe: does the MFMA SHAPE change the delivered FLOP/s on this chip under load?  (MI355X_MICROARCH.md, DVFS give-back item 7:
// v_mfma_f32_16x16x32_bf16 held ~1.12-1.15x the FLOP/s of 32x32x16 on LDS-fed loops.)  Same output tile per wave (64 x 64 fp32
// accumulators = 64 registers), same operand bytes per FLOP, random fp16 operands:
//   VAR 0: 32x32x16, operands in registers          VAR 1: 16x16x32, operands in registers
//   VAR 2: 32x32x16, every operand re-read from LDS (8 ds_read_b128 per 12 MFMAs)
//   VAR 3: 16x16x32, every operand re-read from LDS (16 ds_read_b128 per 48 MFMAs)
// LDS reads are lane-linear (conflict-free by construction): the probe prices the shape, not a layout.
// FLOP accounting (round 3 - the round-2 version priced every variant as 108 x 32768 FLOP per step and so UNDER-counted the
// 16x16x32 arms by exactly 2x: their "tap" is a 32-channel k-step): a step = 9 groups; VAR 0/2: 12 MFMAs of 32x32x16 (32768 FLOP)
// per group = 3.54 MFLOP per wave and step; VAR 1/3: 48 MFMAs of 16x16x32 (16384 FLOP) per group = 7.08 MFLOP.  Printed per
// variant: shader cycles per MFMA and per MFLOP (s_memtime), the in-kernel clock (s_memtime / s_memrealtime x 100 MHz, guide
// "DVFS give-back" item 6) and TFLOP/s by wall time.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_probe mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(256, 2) void probe(float* out, unsigned long long* cyc, unsigned long long* rt, int steps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // 64 KB of pseudo-random fp16 values in [1, 2) with random sign
    for (int i = tid; i < 16384; i += 256) {
        unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u | (x & 0x83ff83ffu);
    }
    __syncthreads();
    const unsigned char* base = smem + lane * 16;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if constexpr (VAR == 0 || VAR == 2) {
        f32x16 acc[2][2];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
        half8 fa[2][2], fb[2][2];
        for (int a = 0; a < 2; ++a) for (int p = 0; p < 2; ++p) { fa[a][p] = *(const half8*)(base + (a * 2 + p) * 1024); fb[a][p] = *(const half8*)(base + (4 + a * 2 + p) * 1024); }
        for (int st = 0; st < steps; ++st) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if constexpr (VAR == 2) {
                    const int o = ((st * 9 + tap) & 7) * 8192;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int p = 0; p < 2; ++p) { fa[a][p] = *(const half8*)(base + o + (a * 2 + p) * 1024); fb[a][p] = *(const half8*)(base + o + (4 + a * 2 + p) * 1024); }
                }
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a][p == 0], fb[b][p == 1], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) s += acc[a][b][i];
    } else {
        f32x4 acc[4][4];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) acc[a][b][i] = 0.f;
        half8 fa[4][2], fb[4][2];
        for (int a = 0; a < 4; ++a) for (int p = 0; p < 2; ++p) { fa[a][p] = *(const half8*)(base + (a * 2 + p) * 1024); fb[a][p] = *(const half8*)(base + (8 + a * 2 + p) * 1024); }
        for (int st = 0; st < steps; ++st) {
            // one "tap" here = 32 channels: 48 MFMAs of 16x16x32; 5 of them = 4.5 taps x 2 ... keep the FLOPs equal: 108 x 32x32x16
            // = 432 x 16x16x32 per step -> 9 groups of 48
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if constexpr (VAR == 3) {
                    const int o = ((st * 9 + tap) & 3) * 16384;
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int p = 0; p < 2; ++p) { fa[a][p] = *(const half8*)(base + o + (a * 2 + p) * 1024); fb[a][p] = *(const half8*)(base + o + (8 + a * 2 + p) * 1024); }
                }
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a][p == 0], fb[b][p == 1], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) s += acc[a][b][i];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int VAR> double run(int grid, size_t pad, const char* name, int reps) {
    // FLOPs per step and wave: VAR 0/2: 108 x 32x32x16 (32768 FLOP each); VAR 1/3: 432 x 16x16x32 (16384 FLOP each) = TWICE as many
    const int steps = 300; float* out; unsigned long long* cyc; unsigned long long* rt;
    const int mfma_per_step = (VAR & 1) ? 432 : 108; const double flop_per_mfma = (VAR & 1) ? 16384.0 : 32768.0;
    hipMalloc(&out, (size_t)grid * 256 * 4); hipMalloc(&cyc, (size_t)grid * 8); hipMalloc(&rt, (size_t)grid * 8);
    const size_t smem = 65536 + pad;
    hipFuncSetAttribute((const void*)probe<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) probe<VAR><<<grid, 256, smem>>>(out, cyc, rt, steps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) probe<VAR><<<grid, 256, smem>>>(out, cyc, rt, steps);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<unsigned long long> h(grid), hr(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), rt, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0, avr = 0; for (auto v : h) avg += v; for (auto v : hr) avr += v; avg /= grid; avr /= grid;
    const double flops = (double)grid * 4 * steps * mfma_per_step * flop_per_mfma;
    printf("%-28s grid %4d: %6.2f cyc/MFMA %7.1f cyc/MFLOP/wave  clock %.2f GHz  %.3f ms  %6.0f TFLOP/s fp16-MFMA\n", name, grid,
           avg / steps / mfma_per_step, avg / steps / (mfma_per_step * flop_per_mfma / 1e6), avg / avr * 0.1, ms, flops / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc); hipFree(rt);
    return ms;
}
int main() {
    for (int round = 0; round < 3; ++round) {       // interleaved rounds in one process
        run<0>(512, 0, "32x32x16 regs, 2 WG/CU", 20);
        run<1>(512, 0, "16x16x32 regs, 2 WG/CU", 20);
        run<2>(512, 0, "32x32x16 LDS-fed, 2 WG/CU", 20);
        run<3>(512, 0, "16x16x32 LDS-fed, 2 WG/CU", 20);
        run<0>(256, 40000, "32x32x16 regs, 1 WG/CU", 20);
        run<1>(256, 40000, "16x16x32 regs, 1 WG/CU", 20);
        run<2>(256, 40000, "32x32x16 LDS-fed, 1 WG/CU", 20);
        run<3>(256, 40000, "16x16x32 LDS-fed, 1 WG/CU", 20);
    }
    return 0;
}
