// Coronal maximum / mean projection of a CT volume (the step right before the hot path: reference ts2d/tool.py:152-160 ->
// ts2d/core/util/image.py:46-101, sitk Max/MeanProjectionImageFilter along axis 1).  One thread per output pixel (z, x);
// the volume is addressed through signed element strides, so the axis permutation / flips of `reorient_image` (DICOMOrient
// 'RAI') cost nothing.  Pure HBM streaming.  Mean of integer volumes follows ITK: exact sum, truncated back to the
// integer type [UPSTREAM-RECALL], then cast to float (reference tool.py:182-185).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace ts2d {

template <typename T>
__global__ void project_coronal(const T* __restrict__ vol, int nz, int ny, int nx, long long sz, long long sy, long long sx,
                                long long base, float* __restrict__ out_max, float* __restrict__ out_mean) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nz * nx) return;
    const int x = (int)(i % nx), z = (int)(i / nx);
    const T* p = vol + base + z * sz + x * sx;
    if constexpr (std::is_floating_point<T>::value) {
        float mx = (float)p[0]; double s = 0.0;
        for (int y = 0; y < ny; ++y) { const float v = (float)p[y * sy]; mx = v > mx ? v : mx; s += (double)v; }
        out_max[i] = mx; out_mean[i] = (float)(s / ny);
    } else {                                                       // integer volume: exact sum, real-valued mean
        long long mx = (long long)p[0], s = 0;
        for (int y = 0; y < ny; ++y) { const long long v = (long long)p[y * sy]; mx = v > mx ? v : mx; s += v; }
        out_max[i] = (float)mx; out_mean[i] = (float)((double)s / (double)ny);
    }
}

// Per-channel z-score of the two projections (nnU-Net ZScoreNormalization without mask, the step between TS2D._project and the
// network: ts2d/tool.py:152-160,182-185 -> prediction_worker.py:194-199): mean and population standard deviation in float64,
// two passes, (x - mean) / max(std, 1e-8) rounded once to float.  Deterministic: kZBlocks fixed-size partial sums per channel,
// added in index order by ONE thread (no float atomics).  zs_partial<0>: sum(x); zs_partial<1>: sum((x - mean)^2).
constexpr int kZBlocks = 64;

template <int PASS>
__global__ __launch_bounds__(256) void zs_partial(const float* __restrict__ x, long long n, const double* __restrict__ stats, double* __restrict__ part) {
    // grid = (kZBlocks, channels); x = [channels][n]; stats = [channels][2] (mean, std); part = [channels][kZBlocks]
    const int c = blockIdx.y;
    const float* p = x + (long long)c * n;
    const double mean = PASS ? stats[2 * c] : 0.0;
    const long long per = (n + kZBlocks - 1) / kZBlocks, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    double acc = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) { const double d = (double)p[i] - mean; acc += PASS ? d * d : d; }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) part[c * kZBlocks + blockIdx.x] = red[0];
}

template <int PASS>
__global__ void zs_combine(const double* __restrict__ part, long long n, double* __restrict__ stats) {
    const int c = threadIdx.x;                     // one thread per channel
    double s = 0.0;
    for (int b = 0; b < kZBlocks; ++b) s += part[c * kZBlocks + b];
    if (PASS == 0) stats[2 * c] = s / (double)n; else stats[2 * c + 1] = sqrt(s / (double)n);
}

// normalise + the non-zero bounding box of the UN-normalised data over all channels (nnU-Net crops to it before normalising: the
// caller uses the device result only when the box is the whole image).  box = {min row, max row, min col, max col} via integer atomics.
__global__ void zs_apply(const float* __restrict__ x, long long n, int nx, int channels, const double* __restrict__ stats,
                         float* __restrict__ out, int* __restrict__ box) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool nz = false;
    for (int c = 0; c < channels; ++c) {
        const float v = x[(long long)c * n + i];
        nz |= v != 0.f;
        const double sd = stats[2 * c + 1] > 1e-8 ? stats[2 * c + 1] : 1e-8;
        out[(long long)c * n + i] = (float)(((double)v - stats[2 * c]) / sd);
    }
    if (nz) {
        const int row = (int)(i / nx), col = (int)(i % nx);
        atomicMin(box + 0, row); atomicMax(box + 1, row); atomicMin(box + 2, col); atomicMax(box + 3, col);
    }
}

// Synthetic slice stream (BASELINE config 4: "synthetic 10k-slice stream ... generated on device per rank from (seed, slice_index)"):
// the portable counter-based generator of totalsegmentator2d_amd/prng.py restated for the device - splitmix64 of the element
// counter, eight 16-bit uniforms summed (Irwin-Hall) and standardised in double, rounded once to float.  Element e of the stream
// depends on (key, e) only, so any rank produces the bits of any slice; integer arithmetic + one IEEE double multiply: the values
// are bit-identical to prng.normal_f32(seed, stream, ..., offset) on the host (tests/test_gpu_stream.py).
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void synth_normal(float* __restrict__ out, unsigned long long key, unsigned long long first, unsigned long long n) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long idx = first + i;
    const unsigned long long h1 = splitmix64((idx * 2ull) * 0xD1342543DE82EF95ull + key);
    const unsigned long long h2 = splitmix64((idx * 2ull + 1ull) * 0xD1342543DE82EF95ull + key);
    long long s = 0;
#pragma unroll
    for (int sh = 0; sh < 64; sh += 16) s += (long long)((h1 >> sh) & 0xFFFFull) + (long long)((h2 >> sh) & 0xFFFFull);
    // 1 / (65536 * sqrt(8 / 12)): the same double constant as prng._IH_SCALE (passed as its bit pattern: no libm on either side)
    const double scale = __longlong_as_double(0x3EF3988E1409212Ell);
    out[i] = (float)((double)(s - 4 * 65535) * scale);
}

}  // namespace ts2d
