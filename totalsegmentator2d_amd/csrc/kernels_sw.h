// Device side of nnU-Net's sliding-window inference: tile gather with mirroring, and the
// Gaussian-weighted aggregation into upstream's float16 buffers (predicted_logits / n_predictions / gaussian are torch.half;
// a half operation = fp32 operation + round-to-nearest-even to half, which is what ATen's CPU half kernels and numpy do).
// Two orders (ts2d_engine_set_tile_dtype; pinned by plain-ATen statements in tests/test_oracle.py):
//   tile_half == 0 (default, the reference's CPU path): the mirror-averaged tile stays fp32, `p *= g` is fp32 x float(g) in
//     fp32, `logits[sl] += p` is a float add with ONE cast to half;
//   tile_half != 0 (the CUDA autocast path): the tile is cast to half first, the product and the sum each round to half.
// Tiles are accumulated in upstream order per output pixel, so the result is bit-identical to the host implementation in
// predictor.py (tests/test_gpu_predictor.py).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace ts2d {

// batch row (t * V + v) = tile t, mirror variant v; variant flips: bit0 = flip H (tensor dim 2), bit1 = flip W (dim 3)
__global__ void sw_gather(const float* __restrict__ img, int C, int Hp, int Wp, int ph, int pw, int V,
                          const int* __restrict__ tile_y, const int* __restrict__ tile_x, const int* __restrict__ vflip,
                          float* __restrict__ batch, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % pw); long long r = i / pw;
    const int y = (int)(r % ph); r /= ph;
    const int c = (int)(r % C); const int row = (int)(r / C);
    const int t = row / V, f = vflip[row % V];
    const int sy = (f & 1) ? ph - 1 - y : y, sx = (f & 2) ? pw - 1 - x : x;
    batch[i] = img[((size_t)c * Hp + tile_y[t] + sy) * Wp + tile_x[t] + sx];
}

__device__ __forceinline__ __half h_mul(__half a, __half b) { return __float2half_rn(__half2float(a) * __half2float(b)); }
__device__ __forceinline__ __half h_add(__half a, __half b) { return __float2half_rn(__half2float(a) + __half2float(b)); }
__device__ __forceinline__ __half h_div(__half a, __half b) { return __float2half_rn(__half2float(a) / __half2float(b)); }

// one thread per output element (k, Y, X) of the padded image
__global__ void sw_aggregate(const float* __restrict__ logits, int K, int Hp, int Wp, int ph, int pw, int T, int V,
                             const int* __restrict__ tile_y, const int* __restrict__ tile_x, const int* __restrict__ vflip,
                             const __half* __restrict__ gauss, __half* __restrict__ out16, uint8_t* __restrict__ seg,
                             float thr, long long total, int* __restrict__ inf_flag, int tile_half) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int X = (int)(i % Wp); long long r = i / Wp;
    const int Y = (int)(r % Hp); const int k = (int)(r / Hp);
    __half acc = __float2half_rn(0.f), n = __float2half_rn(0.f);
    for (int t = 0; t < T; ++t) {
        const int yy = Y - tile_y[t], xx = X - tile_x[t];
        if (yy < 0 || yy >= ph || xx < 0 || xx >= pw) continue;
        const size_t plane = (size_t)ph * pw;
        float y = logits[((size_t)(t * V) * K + k) * plane + (size_t)yy * pw + xx];
        for (int v = 1; v < V; ++v) {                      // y += flip(net(flip(x)), axes): read the un-flipped position
            const int f = vflip[v];
            const int sy = (f & 1) ? ph - 1 - yy : yy, sx = (f & 2) ? pw - 1 - xx : xx;
            y += logits[((size_t)(t * V + v) * K + k) * plane + (size_t)sy * pw + sx];
        }
        if (V > 1) y /= (float)V;
        const __half g = gauss ? gauss[(size_t)yy * pw + xx] : __float2half_rn(1.f);
        if (tile_half) {
            __half p = __float2half_rn(y);
            if (gauss) p = h_mul(p, g);
            acc = h_add(acc, p);
        } else {      // (explicit _rn intrinsics: the product must round to fp32 before the add - no FMA contraction)
            const float pf = gauss ? __fmul_rn(y, __half2float(g)) : y;
            acc = __float2half_rn(__fadd_rn(__half2float(acc), pf));
        }
        n = h_add(n, g);
    }
    const __half res = h_div(acc, n);
    if ((__half_as_ushort(res) & 0x7FFFu) == 0x7C00u) *inf_flag = 1;      // upstream's "Encountered inf in predicted array" check
    if (out16) out16[i] = res;
    if (seg) seg[i] = __half2float(res) > thr ? 1 : 0;
}

}  // namespace ts2d
