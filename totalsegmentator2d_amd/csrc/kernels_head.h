// Segmentation head (SURVEY K7 + A7): 1x1 conv C=32 -> K <= 32 logits + sigmoid > 0.5 masks, on the matrix cores.
// One wave owns blocks of 32 consecutive pixels: the A operand of v_mfma_f32_32x32x16_f16 (row = pixel, 8 channels per lane)
// is read STRAIGHT from the NHWC activation (16/32 contiguous bytes per lane, no LDS), normalised + LeakyReLU'd in registers and
// split into fp16 hi/lo like the convolutions (3 products, fp32-equivalent; 1 product for fp16 storage); the B operand
// (pre-split, pre-scaled weights, K padded to 32 columns) stays in registers for the whole kernel.  The C/D layout puts one
// output channel per lane and 4 consecutive pixels per register quad: NCHW logits leave as 16-byte stores, the mask word of a
// 32-pixel block is assembled from the two lane halves.  The next block's loads are issued before the current block's arithmetic.
// The exact mode keeps head_1x1 (kernels.h), whose result is the plain fp32 FMA chain.
#pragma once
#include "kernels_h32.h"

namespace ts2d {

template <typename ST, int NP>
__global__ __launch_bounds__(256) void head_mfma32(const HeadArgs a, const int blocks_per_wave) {
    constexpr int C = 32;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
    const long long nblk = a.total >> 5;                                    // blocks of 32 pixels (HW % 32 == 0)
    const long long b0 = ((long long)blockIdx.x * 4 + wv) * blocks_per_wave;
    const long long b1 = b0 + blocks_per_wave < nblk ? b0 + blocks_per_wave : nblk;
    if (b0 >= b1) return;

    // B operand: column r (output channel), channels 16 ks + 8 h .. + 7: [ks][column][16 hi | 16 lo] halves
    half8 bh[2], bl[2];
    const _Float16* wp = reinterpret_cast<const _Float16*>(a.wph);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bh[ks] = *reinterpret_cast<const half8*>(wp + ((ks * 32 + r) * 32 + 8 * h));
        if (NP == 3) bl[ks] = *reinterpret_cast<const half8*>(wp + ((ks * 32 + r) * 32 + 16 + 8 * h));
    }
    const float oscale = *a.oscale;
    const float bv = r < a.K ? a.bias[r] : 0.f;
    const _Float16 slope_h = (_Float16)a.slope;
    const unsigned slope2 = (unsigned)__builtin_bit_cast(unsigned short, slope_h) * 0x10001u;

    long long n = (b0 << 5) / a.HW;                                         // image of the first block
    int o = (int)((b0 << 5) - n * a.HW);                                    // pixel offset inside the image
    f32x4 s[4], t[4];                                                       // scale / shift of this lane's 16 channels
    auto load_st = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const size_t q = (size_t)n * C + ks * 16 + 8 * h;
            s[2 * ks] = *reinterpret_cast<const f32x4*>(a.sc + q); s[2 * ks + 1] = *reinterpret_cast<const f32x4*>(a.sc + q + 4);
            t[2 * ks] = *reinterpret_cast<const f32x4*>(a.sh + q); t[2 * ks + 1] = *reinterpret_cast<const f32x4*>(a.sh + q + 4);
        }
    };
    load_st();

    constexpr int NL = sizeof(ST) == 4 ? 4 : 2;                             // 16-byte loads per lane and block
    uint4 raw[NL];
    const ST* src = reinterpret_cast<const ST*>(a.src);
    auto prefetch = [&](long long b) {
        const ST* p = src + ((size_t)(b << 5) + r) * C + 8 * h;
#pragma unroll
        for (int l = 0; l < NL; ++l)                                        // fp32: (ks, half) = (l >> 1, l & 1); fp16: ks = l
            raw[l] = *reinterpret_cast<const uint4*>(p + (sizeof(ST) == 4 ? (l >> 1) * 16 + (l & 1) * 4 : l * 16));
    };
    prefetch(b0);
    bool bad = false;
    // Mask words leave FOUR blocks at a time (round 5): a word per block and channel is a 4-byte write into a 128-byte line that 31 later
    // writes complete - measured 0.09 ms of the kernel's 0.9 for 38 MB of masks.  A wave's blocks are consecutive, so lane r keeps the words
    // of blocks 4 g .. 4 g + 3 of its channel and stores them as 16 bytes (groups never straddle an image: both counts are multiples of 4).
    const bool quad = (blocks_per_wave & 3) == 0 && ((a.HW >> 5) & 3) == 0;
    typedef unsigned u32x4m __attribute__((ext_vector_type(4)));
    u32x4m mw = {0u, 0u, 0u, 0u};
    for (long long b = b0; b < b1; ++b) {
        uint4 x[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) x[l] = raw[l];
        if (b + 1 < b1) prefetch(b + 1);
        half8 ah[2], al[2];
        if constexpr (sizeof(ST) == 4) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f32x4 va = __builtin_bit_cast(f32x4, x[2 * ks]), vb = __builtin_bit_cast(f32x4, x[2 * ks + 1]);
                va = va * s[2 * ks] + t[2 * ks]; vb = vb * s[2 * ks + 1] + t[2 * ks + 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    va[e] = fmaxf(va[e], va[e] * a.slope);
                    vb[e] = fmaxf(vb[e], vb[e] * a.slope);
                    const _Float16 ha = (_Float16)va[e], hb = (_Float16)vb[e];
                    ah[ks][e] = ha; ah[ks][e + 4] = hb;
                    if (NP == 3) { al[ks][e] = (_Float16)(va[e] - (float)ha); al[ks][e + 4] = (_Float16)(vb[e] - (float)hb); }
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                ah[ks] = __builtin_bit_cast(half8, norm_lrelu_8(x[ks], s[2 * ks], s[2 * ks + 1], t[2 * ks], t[2 * ks + 1], slope2));
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (NP == 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[ks], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[ks], acc, 0, 0, 0);
        }
        // lane = output channel r; register i = pixel (i & 3) + 8 (i >> 2) + 4 h of the block
        unsigned bits = 0;
        float* lp = a.logits != nullptr ? a.logits + ((size_t)n * a.K + r) * a.HW + o + 4 * h : nullptr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[4 * q + e] * oscale + bv;
                bad |= not_finite(v[e]);
                bits |= (v[e] > kSigmoidHalfThreshold ? 1u : 0u) << (e + 8 * q + 4 * h);
            }
            if (lp != nullptr && r < a.K) *reinterpret_cast<f32x4*>(lp + 8 * q) = v;
        }
        if (a.mask != nullptr) {
            bits |= __shfl_xor(bits, 32);
            if (quad) {
                const int u = (int)(b - b0) & 3;
                mw[0] = u == 0 ? bits : mw[0]; mw[1] = u == 1 ? bits : mw[1]; mw[2] = u == 2 ? bits : mw[2]; mw[3] = u == 3 ? bits : mw[3];
                if (u == 3 && h == 0 && r < a.K)
                    *reinterpret_cast<u32x4m*>(a.mask + ((((size_t)n * a.K + r) * a.HW + o) >> 5) - 3) = mw;
            } else if (h == 0 && r < a.K) a.mask[(((size_t)n * a.K + r) * a.HW + o) >> 5] = bits;
        }
        o += 32;
        if (o == a.HW) { o = 0; ++n; if (b + 1 < b1) load_st(); }
    }
    if (bad && r < a.K && a.nonfinite != nullptr) atomicOr(a.nonfinite, 1);
}

}  // namespace ts2d
