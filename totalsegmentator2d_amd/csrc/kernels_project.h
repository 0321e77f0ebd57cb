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
    } else {                                                       // integer volume: exact sum, truncating division
        long long mx = (long long)p[0], s = 0;
        for (int y = 0; y < ny; ++y) { const long long v = (long long)p[y * sy]; mx = v > mx ? v : mx; s += v; }
        out_max[i] = (float)mx; out_mean[i] = (float)(T)(s / ny);
    }
}

}  // namespace ts2d
