// Split-fp16 ("f16x3") 3x3 convolution: fp32 activations and weights are split on the fly into fp16 hi + lo parts
// (x = hi + lo to 22 significant bits) and the contraction is issued as THREE v_mfma_f32_32x32x16_f16 products
// (hi*hi + hi*lo + lo*hi) accumulated in fp32.  Measured on the canonical net: the logits are as
// close to the fp64-accumulating truth as ATen's native fp32 path is (4.6e-5 vs 4.8e-5 max-abs), while the MFMA
// cost per MAC drops 5.3x vs v_mfma_f32_32x32x2_f32 (3 x 32 cycles per 32x32x16 block instead of 8 x 64).
// Weights are pre-split at load time, pre-scaled by a per-layer power of two so that hi AND lo stay in fp16's normal
// range; the epilogue multiplies by the exact inverse.
#pragma once
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace ts2d {

// LDS record (one patch pixel or one weight column, 16 channels): [16 hi halves | 16 lo halves | 16 B pad] = 80 B.
// 80 B = 5 x 16-B slots: consecutive records start on slots 0,5,10,15,4,... -> a ds_read_b128 lane group (16 lanes on
// 16 consecutive records) touches 16 distinct slots: conflict-free.
constexpr int kRec = 80;

// Activation STORAGE type ST: float (the fp32-parity modes) or _Float16 ("mixed fp16": fp16 storage, one fp16 MFMA product,
// fp32 accumulation and statistics - BASELINE configs 3/5).  NP = MFMA products per MAC: 3 (hi/lo split) or 1.
template <typename ST> struct Raw8;
template <> struct Raw8<float> {
    f32x4 a, b;
    __device__ __forceinline__ void zero() { a = f32x4{0.f, 0.f, 0.f, 0.f}; b = a; }
    __device__ __forceinline__ void load(const void* base, size_t off) {
        const float* p = reinterpret_cast<const float*>(base) + off;
        a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4);
    }
    __device__ __forceinline__ void get(f32x4& va, f32x4& vb) const { va = a; vb = b; }
};
template <> struct Raw8<_Float16> {
    half8 h;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (_Float16)0.f;
    }
    __device__ __forceinline__ void load(const void* base, size_t off) {
        h = *reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(base) + off);
    }
    __device__ __forceinline__ void get(f32x4& va, f32x4& vb) const {
#pragma unroll
        for (int e = 0; e < 4; ++e) { va[e] = (float)h[e]; vb[e] = (float)h[e + 4]; }
    }
};
// (round_act / store_act: kernels.h)


// fp32 x 8 -> fp16 hi[8] + lo[8] with x = hi + lo to 22 bits: hi = RNE(x) by v_cvt_pk_f16_f32, lo = RNE(x - hi) by ONE mixed-precision
// FMA per element (v_fma_mixlo/hi_f16 computes -1 * hi + x from the fp16 hi and the fp32 x exactly and rounds once) - the same
// values as (half)(x - (float)hi), 1.5 instead of ~2.75 VALU instructions per element.
__device__ __forceinline__ void split_hi_lo_8(const f32x4& va, const f32x4& vb, uint4& hi, uint4& lo) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    asm volatile(
        "v_cvt_pk_f16_f32 %0, %8, %9\n\t"
        "v_cvt_pk_f16_f32 %1, %10, %11\n\t"
        "v_cvt_pk_f16_f32 %2, %12, %13\n\t"
        "v_cvt_pk_f16_f32 %3, %14, %15\n\t"
        "v_fma_mixlo_f16 %4, %0, -1.0, %8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %5, %1, -1.0, %10 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %6, %2, -1.0, %12 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %7, %3, -1.0, %14 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %4, %0, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %5, %1, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %6, %2, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %7, %3, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(va[0]), "v"(va[1]), "v"(va[2]), "v"(va[3]), "v"(vb[0]), "v"(vb[1]), "v"(vb[2]), "v"(vb[3]));
    hi = uint4{h0, h1, h2, h3}; lo = uint4{l0, l1, l2, l3};
}

// fp32 x 4 -> fp16 hi[4] + lo[4] (x = hi + lo to 22 bits): the 4-value form of split_hi_lo_8 above
__device__ __forceinline__ void split_hi_lo_4(const f32x4& v, uint2& hi, uint2& lo) {
    unsigned h0, h1, l0, l1;
    asm volatile(
        "v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
        "v_fma_mixlo_f16 %2, %0, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %1, -1.0, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %1, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    hi = uint2{h0, h1}; lo = uint2{l0, l1};
}

// buffer store of one activation in the storage type (a per-lane byte offset + a wave-uniform one)
template <typename ST> __device__ __forceinline__ void buffer_store_act(float v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ void buffer_store_act<float>(float v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 0);
}
template <> __device__ __forceinline__ void buffer_store_act<_Float16>(float v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), rs, voff, soff, 0);
}

// Shared epilogue: undo the weight pre-scale, add bias, store the raw NHWC output, per-tile InstanceNorm partials.
// C/D map of the 32x32 MFMA: column = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5).

template <int BN, typename ST = float>
__device__ __forceinline__ void split_epilogue(const ConvArgs& a, f32x16 (&acc_t)[2][BN / 32], unsigned char* smem8,
                                               int n0col, int nimg0, int ty0, int tx0, int tpi, int tin) {
    constexpr int NT = BN / 32;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int NIMG = a.NIMG;
    const float oscale = *a.oscale;
    if (a.part == nullptr && NIMG > 1 && a.TH == a.Ht && a.TW == a.Wt && NIMG * a.TH * a.TW == 256) {
        // tile = NIMG WHOLE images (the <= 8 x 8 levels) and no statistics from here (split-K slice, or stats_direct behind the op): GEMM row m is pixel
        // nimg0 * HW + m of the flattened [B * H * W][Cout] output - one lane offset per block, a scalar offset per row, images beyond the batch dropped by
        // the buffer's range check (round 6; the loop below spends ~ 30 VALU per row on divisions, bounds tests and 64-bit addresses)
        const size_t slice = a.ksplit > 1 ? (size_t)blockIdx.y * a.kslice_stride : 0;
        const size_t all_b = (size_t)a.B * a.Ht * a.Wt * a.Cout * (a.ksplit > 1 ? sizeof(float) : sizeof(ST));
        const auto rp = __builtin_amdgcn_make_buffer_rsrc(a.ksplit > 1 ? reinterpret_cast<unsigned char*>(reinterpret_cast<float*>(a.dst) + slice)
                                                                       : reinterpret_cast<unsigned char*>(a.dst), 0, (int)all_b, 0x00020000);
        if (all_b < ((size_t)1 << 31)) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = n0col + nt * 32 + r;
                const float bv = a.ksplit > 1 ? 0.f : a.bias[co];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const unsigned px0 = (unsigned)(nimg0 * a.Ht * a.Wt + 64 * w + 32 * mt + 4 * h);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int rowoff = (i & 3) + 8 * (i >> 2);
                        const float v = __builtin_fmaf(acc_t[mt][nt][i], oscale, bv);
                        if (a.ksplit > 1) buffer_store_act<float>(v, rp, (px0 * a.Cout + co) * 4u, (unsigned)(rowoff * a.Cout * 4));
                        else buffer_store_act<ST>(v, rp, (px0 * a.Cout + co) * (unsigned)sizeof(ST), (unsigned)(rowoff * a.Cout * (int)sizeof(ST)));
                    }
                }
            }
            return;
        }
    }
    float st_s[NT], st_q[NT], st_k[NT], st_n[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { st_s[nt] = 0.f; st_q[nt] = 0.f; st_n[nt] = 0.f; }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0col + nt * 32 + r;
        const float bv = a.ksplit > 1 ? 0.f : a.bias[co];      // split-K partials: bias is added by splitk_reduce_stats
        const float kv = stat_pivot(round_act<ST>(__builtin_fmaf(acc_t[0][nt][0], oscale, bv)));      // shifted statistics (kernels.h)
        st_k[nt] = kv;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                const int m = 64 * w + 32 * mt + row;
                int il, ty, tx;
                tile_row(a, m, il, ty, tx);
                const int n = nimg0 + il, oy = ty0 + ty, ox = tx0 + tx;
                if (il < NIMG && n < a.B && oy < a.Ht && ox < a.Wt) {
                    float v = __builtin_fmaf(acc_t[mt][nt][i], oscale, bv);      // explicit FMAs: the same rounding in every instantiation
                    const size_t o = ((size_t)(n * a.Ht + oy) * a.Wt + ox) * a.Cout + co;
                    if (a.ksplit > 1) a.dst[(size_t)blockIdx.y * a.kslice_stride + o] = v;          // fp32 partial
                    else { store_act<ST>(a.dst, o, v); v = round_act<ST>(v); }                      // statistics of what is stored
                    const float d = v - kv;
                    st_s[nt] += d; st_q[nt] = __builtin_fmaf(d, d, st_q[nt]); st_n[nt] += 1.f;
                }
            }
        }
    }
    if (a.part != nullptr) {
        lds_barrier();
        float* red = reinterpret_cast<float*>(smem8);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float s = st_s[nt], q = st_q[nt], n = st_n[nt];
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32); n += __shfl_xor(n, 32);
            if (h == 0) stat_wave_put(red, w * BN + nt * 32 + r, s, q, st_k[nt], n);
        }
        lds_barrier();
        if (tid < BN) stat_tile_store(red, 4, BN, tid, a.part + ((size_t)(nimg0 * tpi + tin) * a.Cout + n0col + tid) * 4);
    }
}

// Epilogue of the one-image-tile kernels (lgNIMG == 0, ksplit == 1).  When the 256-pixel tile lies completely inside the output
// image, the 16 rows a lane holds of a 32x32 block differ from the lane's first row by a WAVE-UNIFORM element offset
// ((rowoff & (TW-1)) + (rowoff >> lgTW) * Wt pixels, rowoff = (i&3) + 8*(i>>2); the lane part 4*(lane>>5) never carries across
// a tile row since TW >= 16): buffer stores with one per-lane base offset per 32x32 block and a scalar offset per row - no
// per-element address arithmetic, bounds tests or exec-mask regions (the generic epilogue spends ~30 VALU + 14 SALU per row).
// Same values, same statistics order as split_epilogue.

template <int BN, typename ST = float>
__device__ __forceinline__ void split_epilogue_one(const ConvArgs& a, f32x16 (&acc_t)[2][BN / 32], unsigned char* smem8,
                                                   int n0col, int nimg0, int ty0, int tx0, int tpi, int tin) {
    constexpr int NT = BN / 32;
    const int TH = a.TH, TW = a.TW;
    const bool whole = a.lgTW >= 4 && a.lgTH + a.lgTW == 8 && ty0 + TH <= a.Ht && tx0 + TW <= a.Wt && nimg0 < a.B;     // wave-uniform
    const bool full = a.ksplit == 1 && whole;
    // (power-of-two tiles of 256 pixels only: lgTW = -1 for the tile shapes of ragged levels, which take the general epilogue)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const float oscale = *a.oscale;
    if (a.ksplit > 1 && whole) {
        // split-K slice (round 6: the small-batch dispatch runs every <= 32 x 32 level this way): un-biased fp32 partials, no statistics - the same
        // one-offset-per-block addressing as below instead of the general epilogue's ~ 30 VALU per row; values as split_epilogue writes them
        const size_t img_f = (size_t)a.Ht * a.Wt * a.Cout;
        const auto rp = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(a.dst) + (size_t)blockIdx.y * a.kslice_stride + (size_t)nimg0 * img_f, 0,
                                                          (int)(img_f * sizeof(float)), 0x00020000);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int m0 = 64 * w + 32 * mt + 4 * h;
                const int oy = ty0 + (m0 >> a.lgTW), ox = tx0 + (m0 & (TW - 1));
                const unsigned voff = (unsigned)(((oy * a.Wt + ox) * a.Cout + n0col + nt * 32 + r) * 4);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rowoff = (i & 3) + 8 * (i >> 2);
                    const unsigned soff = (unsigned)((((rowoff & (TW - 1)) + (rowoff >> a.lgTW) * a.Wt) * a.Cout) * 4);       // scalar
                    buffer_store_act<float>(__builtin_fmaf(acc_t[mt][nt][i], oscale, 0.f), rp, voff, soff);
                }
            }
        return;
    }
    if (!full) { split_epilogue<BN, ST>(a, acc_t, smem8, n0col, nimg0, ty0, tx0, tpi, tin); return; }
    const size_t img_el = (size_t)a.Ht * a.Wt * a.Cout;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<ST*>(a.dst) + (size_t)nimg0 * img_el, 0, (int)(img_el * sizeof(ST)), 0x00020000);
    float st_s[NT], st_q[NT], st_k[NT];
    float bvs[NT];       // every bias value before the first store: a load issued between stores waits (in-order vmcnt) for the stores ahead of it
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bvs[nt] = a.bias[n0col + nt * 32 + r];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0col + nt * 32 + r;
        const float bv = bvs[nt];
        const float kv = stat_pivot(round_act<ST>(__builtin_fmaf(acc_t[0][nt][0], oscale, bv)));      // shifted statistics (kernels.h)
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m0 = 64 * w + 32 * mt + 4 * h;                 // the lane's first row of this block
            const int oy = ty0 + (m0 >> a.lgTW), ox = tx0 + (m0 & (TW - 1));
            const unsigned voff = (unsigned)(((oy * a.Wt + ox) * a.Cout + co) * (int)sizeof(ST));
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rowoff = (i & 3) + 8 * (i >> 2);
                const unsigned soff = (unsigned)((((rowoff & (TW - 1)) + (rowoff >> a.lgTW) * a.Wt) * a.Cout) * (int)sizeof(ST));   // scalar
                float v = __builtin_fmaf(acc_t[mt][nt][i], oscale, bv);          // explicit FMAs: the same rounding as split_epilogue
                buffer_store_act<ST>(v, rs, voff, soff);
                const float d = round_act<ST>(v) - kv;                               // statistics of what is stored
                s += d; q = __builtin_fmaf(d, d, q);
            }
        }
        st_s[nt] = s; st_q[nt] = q; st_k[nt] = kv;
    }
    if (a.part != nullptr) {
        lds_barrier();
        float* red = reinterpret_cast<float*>(smem8);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float s = st_s[nt], q = st_q[nt];
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
            if (h == 0) stat_wave_put(red, w * BN + nt * 32 + r, s, q, st_k[nt], 64.f);
        }
        lds_barrier();
        if (tid < BN) stat_tile_store(red, 4, BN, tid, a.part + ((size_t)(nimg0 * tpi + tin) * a.Cout + n0col + tid) * 4);
    }
}

// PF = chunks of raw patch data kept in flight in registers (2 for the HBM-bound small-Cin layers: more bytes in flight).
template <int BN, int MAXU, int PF = 1, typename ST = float, int NP = 3>
__global__ __launch_bounds__(kBlock, 2) void conv3x3_f16x3(const ConvArgs a) {
    constexpr int NT = BN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];

    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = bid >> 3;
    const int mtile = (q8 / a.n_ctiles) * 8 + xcd;
    const int ctile = q8 % a.n_ctiles;
    if (mtile >= a.n_mtiles) return;
    const int n0col = ctile * BN;

    const int NIMG = a.NIMG;
    const int tpi = a.tiles_x * a.tiles_y;
    const int grp = mtile / tpi, tin = mtile - grp * tpi;
    const int tyi = tin / a.tiles_x, txi = tin - tyi * a.tiles_x;
    const int nimg0 = grp * NIMG;
    const int ty0 = tyi * a.TH, tx0 = txi * a.TW;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int PHW = a.PH * a.PW;
    const int P = PHW * NIMG;
    unsigned char* sA = smem8;
    unsigned char* sB = smem8 + P * kRec;

    // ---- staging plan: unit u = (patch pixel u >> 1, channel octet u & 1); divisions by float reciprocal (exact here)
    int goff[MAXU];
    unsigned long long imgbits = 0;
    const int total = P * 2;
    const float inv_phw = 1.0f / (float)PHW, inv_pw = 1.0f / (float)a.PW;
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int u = tid + it * kBlock;
        int g = -1;
        if (u < total) {
            const int pp = u >> 1;
            const int il = (int)(((float)pp + 0.5f) * inv_phw), rem = pp - il * PHW;
            const int py = (int)(((float)rem + 0.5f) * inv_pw), px = rem - py * a.PW;
            const int n = nimg0 + il, iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            if (n < a.B && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) g = (n * a.Hin + iy) * a.Win + ix;
            imgbits |= (unsigned long long)il << (4 * it);
        }
        goff[it] = g;
    }
    const int oct = (tid & 1) * 8;     // this thread's channel octet inside a 16-channel chunk

    int abase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = 64 * w + 32 * mt + r;
        int il, ty, tx;
        tile_row(a, m, il, ty, tx);
        abase[mt] = (il < NIMG ? (il * PHW + ty * a.PW + tx) * kRec : 0) + 16 * h;
    }
    const int bbase = r * kRec + 16 * h;

    f32x16 acc_t[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_t[mt][nt][i] = 0.f;

    const int nchunks = (a.C0 + a.C1) / 16;
    Raw8<ST> pv[PF][MAXU];             // prefetched raw patch values of the next PF chunks (in flight during the MFMAs)

    auto chunk_src = [&](int ch, const float*& src, const float*& sc, const float*& sh, int& C, int& cb) {
        cb = ch * 16;
        if (cb < a.C0) { src = a.src0; sc = a.sc0; sh = a.sh0; C = a.C0; }
        else { cb -= a.C0; src = a.src1; sc = a.sc1; sh = a.sh1; C = a.C1; }
    };
    auto prefetch = [&](int ch, Raw8<ST> (&pq)[MAXU]) {
        const float* src; const float* sc; const float* sh; int C, cb;
        chunk_src(ch, src, sc, sh, C, cb);
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            pq[it].zero();
            if (goff[it] >= 0) pq[it].load(src, (size_t)goff[it] * C + cb + oct);
        }
    };

    // split-K (tiny layers): this workgroup handles chunks [kbeg, kend) and writes un-biased partials
    const int kper = (nchunks + a.ksplit - 1) / a.ksplit;
    const int kbeg = (int)blockIdx.y * kper, kend = (kbeg + kper < nchunks) ? kbeg + kper : nchunks;
#pragma unroll
    for (int s = 0; s < PF; ++s) if (kbeg + s < kend) prefetch(kbeg + s, pv[s]);
    for (int ch0 = kbeg; ch0 < kend; ch0 += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        const int ch = ch0 + s;
        if (PF > 1 && ch >= kend) break;
        const float* src; const float* sc; const float* sh; int C, cb;
        chunk_src(ch, src, sc, sh, C, cb);
        __syncthreads();   // the previous chunk's MFMA reads of LDS are done
        // ---- weights of this chunk (L2-resident): issue the loads FIRST, their latency hides behind the patch conversion below.
        //      Named registers (an array was demoted to scratch by the compiler).
        constexpr int WU = 9 * BN * 4, WIT = (WU + kBlock - 1) / kBlock;
        // the weight image holds one contiguous [tap][column][16 hi | 16 lo] block per (chunk, column tile): no index arithmetic
        const uint4* wsrc = reinterpret_cast<const uint4*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * (9 * BN * 4);
        uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8;
#define TS2D_WLOAD(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU) && (NP == 3 || (idx & 2) == 0)) R = wsrc[idx]; }
        TS2D_WLOAD(0, w0) TS2D_WLOAD(1, w1) TS2D_WLOAD(2, w2) TS2D_WLOAD(3, w3) TS2D_WLOAD(4, w4)
        TS2D_WLOAD(5, w5) TS2D_WLOAD(6, w6) TS2D_WLOAD(7, w7) TS2D_WLOAD(8, w8)
#undef TS2D_WLOAD
        // ---- patch: normalise + LeakyReLU, split into fp16 hi/lo, write the LDS records
        {
            f32x4 s1a = f32x4{1.f, 1.f, 1.f, 1.f}, s1b = s1a, s2a = f32x4{0.f, 0.f, 0.f, 0.f}, s2b = s2a;
            if (sc != nullptr && NIMG == 1 && nimg0 < a.B) {
                const size_t o = (size_t)nimg0 * C + cb + oct;
                s1a = *reinterpret_cast<const f32x4*>(sc + o); s1b = *reinterpret_cast<const f32x4*>(sc + o + 4);
                s2a = *reinterpret_cast<const f32x4*>(sh + o); s2b = *reinterpret_cast<const f32x4*>(sh + o + 4);
            }
#pragma unroll
            for (int it = 0; it < MAXU; ++it) {
                const int u = tid + it * kBlock;
                if (u < total) {
                    f32x4 va, vb;
                    pv[s][it].get(va, vb);
                    if (sc != nullptr && goff[it] >= 0) {
                        if (NIMG != 1) {
                            const size_t o = (size_t)(nimg0 + (int)((imgbits >> (4 * it)) & 15)) * C + cb + oct;
                            s1a = *reinterpret_cast<const f32x4*>(sc + o); s1b = *reinterpret_cast<const f32x4*>(sc + o + 4);
                            s2a = *reinterpret_cast<const f32x4*>(sh + o); s2b = *reinterpret_cast<const f32x4*>(sh + o + 4);
                        }
                        va = va * s1a + s2a; vb = vb * s1b + s2b;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            va[e] = fmaxf(va[e], va[e] * a.slope);     // LeakyReLU (0 < slope < 1)
                            vb[e] = fmaxf(vb[e], vb[e] * a.slope);
                        }
                    }
                    half8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const _Float16 ha = (_Float16)va[e], hb = (_Float16)vb[e];
                        hi[e] = ha; hi[e + 4] = hb;
                        if (NP == 3) { lo[e] = (_Float16)(va[e] - (float)ha); lo[e + 4] = (_Float16)(vb[e] - (float)hb); }
                    }
                    unsigned char* d = sA + (u >> 1) * kRec + (u & 1) * 16;
                    *reinterpret_cast<half8*>(d) = hi;
                    if (NP == 3) *reinterpret_cast<half8*>(d + 32) = lo;
                }
            }
        }
        // ---- weight registers -> LDS records [tap][col][hi16 | lo16 | pad]
#define TS2D_WSTORE(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU) && (NP == 3 || (idx & 2) == 0)) \
            *reinterpret_cast<uint4*>(sB + (idx >> 2) * kRec + (idx & 3) * 16) = R; }
        TS2D_WSTORE(0, w0) TS2D_WSTORE(1, w1) TS2D_WSTORE(2, w2) TS2D_WSTORE(3, w3) TS2D_WSTORE(4, w4)
        TS2D_WSTORE(5, w5) TS2D_WSTORE(6, w6) TS2D_WSTORE(7, w7) TS2D_WSTORE(8, w8)
#undef TS2D_WSTORE
        __syncthreads();
        if (ch + PF < kend) prefetch(ch + PF, pv[s]);      // HBM latency hides behind the MFMA phases

        f32x16 acc_c[2][NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc_c[mt][nt][i] = 0.f;
        __builtin_amdgcn_s_setprio(1);     // the MFMA phase outranks the co-resident workgroup's staging phase
        constexpr bool PIPE = (BN == 64 && MAXU == 3);     // register budget: 256 VGPRs at 2 workgroups/CU; BN = 32 must stay <= 168 for 3
        if constexpr (PIPE) {
        // Software-pipelined over the 9 taps: the fragments of tap t+1 are read (double-buffered registers) while the MFMAs of
        // tap t run; the sched_barrier keeps the read-ahead at exactly one tap.  Probe (scripts/probes/mfma_phase_probe.hip):
        // 5.6k instead of 17.9k cycles per chunk for the bare loop at 2 workgroups/CU.
        half8 fa[2][2][2], fb[2][NT][2];            // [buffer][tile][hi, lo]
        auto load_frags = [&](int buf, int tap) {
            const int toff = ((tap / 3) * a.PW + (tap % 3)) * kRec;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                fa[buf][mt][0] = *reinterpret_cast<const half8*>(sA + abase[mt] + toff);
                if (NP == 3) fa[buf][mt][1] = *reinterpret_cast<const half8*>(sA + abase[mt] + toff + 32);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                fb[buf][nt][0] = *reinterpret_cast<const half8*>(sB + (tap * BN + nt * 32) * kRec + bbase);
                if (NP == 3) fb[buf][nt][1] = *reinterpret_cast<const half8*>(sB + (tap * BN + nt * 32) * kRec + bbase + 32);
            }
        };
        load_frags(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = tap & 1;
            if (tap + 1 < 9) load_frags(cur ^ 1, tap + 1);
            if (NP == 3) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][1], fb[cur][nt][0], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][1], acc_c[mt][nt], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][0], acc_c[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        } else {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = ((tap / 3) * a.PW + (tap % 3)) * kRec;
            // fragment reads in first-use order (lo*hi products first): the MFMAs can start after two reads have landed
            half8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) if (NP == 3) al[mt] = *reinterpret_cast<const half8*>(sA + abase[mt] + toff + 32);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bh[nt] = *reinterpret_cast<const half8*>(sB + (tap * BN + nt * 32) * kRec + bbase);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) ah[mt] = *reinterpret_cast<const half8*>(sA + abase[mt] + toff);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) if (NP == 3) bl[nt] = *reinterpret_cast<const half8*>(sB + (tap * BN + nt * 32) * kRec + bbase + 32);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    { if (NP == 3) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc_c[mt][nt], 0, 0, 0); }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    { if (NP == 3) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc_c[mt][nt], 0, 0, 0); }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc_c[mt][nt], 0, 0, 0);
        }
        }
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc_t[mt][nt] += acc_c[mt][nt];
    }
    }

    split_epilogue<BN, ST>(a, acc_t, smem8, n0col, nimg0, ty0, tx0, tpi, tin);
}

// ------------------------------------------------------------------------------------------------------------
// Stride-2 3x3 conv (the strided-conv downsample) in split-fp16.  The input patch of a stride-2 tile is 4x
// larger per output pixel, so chunks are 8 channels and the MFMA's K = 16 is filled with 8 channels x 2 TAPS
// (k = 8h + e: lane half h selects tap 2s + h of k-step s; the 10th tap is zero weights).  Patch columns are stored
// even-columns-first so that lanes on consecutive OUTPUT pixels read consecutive records (conflict-free ds_read_b128).
// LDS: patch record 48 B = [8 hi | 8 lo | pad]; weight record 80 B = [16 hi | 16 lo | pad] per (k-step, column).
// ------------------------------------------------------------------------------------------------------------
constexpr int kRec8 = 48;

template <int BN, int MAXU, typename ST = float, int NP = 3>
__global__ __launch_bounds__(kBlock, 2) void conv3x3s2_f16x3(const ConvArgs a) {
    constexpr int NT = BN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];

    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = bid >> 3;
    const int mtile = (q8 / a.n_ctiles) * 8 + xcd;
    const int ctile = q8 % a.n_ctiles;
    if (mtile >= a.n_mtiles) return;
    const int n0col = ctile * BN;

    const int NIMG = a.NIMG;
    const int tpi = a.tiles_x * a.tiles_y;
    const int grp = mtile / tpi, tin = mtile - grp * tpi;
    const int tyi = tin / a.tiles_x, txi = tin - tyi * a.tiles_x;
    const int nimg0 = grp * NIMG;
    const int ty0 = tyi * a.TH, tx0 = txi * a.TW;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int PHW = a.PH * a.PW, PWe = (a.PW + 1) >> 1;
    const int P = PHW * NIMG;
    unsigned char* sA = smem8;
    unsigned char* sB = smem8 + P * kRec8;

    int goff[MAXU], lrec[MAXU];
    unsigned long long imgbits = 0;
    const float inv_phw = 1.0f / (float)PHW, inv_pw = 1.0f / (float)a.PW;
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int u = tid + it * kBlock;
        int g = -1, lr = 0;
        if (u < P) {
            const int il = (int)(((float)u + 0.5f) * inv_phw), rem = u - il * PHW;
            const int py = (int)(((float)rem + 0.5f) * inv_pw), px = rem - py * a.PW;
            const int n = nimg0 + il, iy = 2 * ty0 - 1 + py, ix = 2 * tx0 - 1 + px;
            if (n < a.B && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) g = (n * a.Hin + iy) * a.Win + ix;
            lr = (il * PHW + py * a.PW + ((px & 1) ? PWe + (px >> 1) : (px >> 1))) * kRec8;
            imgbits |= (unsigned long long)il << (4 * it);
        }
        goff[it] = g; lrec[it] = lr;
    }

    int abase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = 64 * w + 32 * mt + r;
        int il, ty, tx;
        tile_row(a, m, il, ty, tx);
        abase[mt] = (il < NIMG ? (il * PHW + 2 * ty * a.PW + tx) * kRec8 : 0);
    }
    int tofl[5];      // this lane half's tap offset per k-step (tap 9 does not exist: reuse tap 8, its weights are 0)
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int t = (2 * s + h) < 9 ? (2 * s + h) : 8;
        const int dy = t / 3, dx = t - 3 * dy;
        tofl[s] = (dy * a.PW + ((dx & 1) ? PWe : 0) + (dx >> 1)) * kRec8;
    }
    const int bbase = r * kRec + 16 * h;

    f32x16 acc_t[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_t[mt][nt][i] = 0.f;

    const int nchunks = a.C0 / 8;      // the strided conv never reads a concat
    Raw8<ST> pv[MAXU];
    auto prefetch = [&](int ch) {
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            pv[it].zero();
            if (goff[it] >= 0) pv[it].load(a.src0, (size_t)goff[it] * a.C0 + ch * 8);
        }
    };

    const int kper = (nchunks + a.ksplit - 1) / a.ksplit;
    const int kbeg = (int)blockIdx.y * kper, kend = (kbeg + kper < nchunks) ? kbeg + kper : nchunks;
    if (kbeg < kend) prefetch(kbeg);
    for (int ch = kbeg; ch < kend; ++ch) {
        const int cb = ch * 8;
        __syncthreads();
        {
            f32x4 s1a = f32x4{1.f, 1.f, 1.f, 1.f}, s1b = s1a, s2a = f32x4{0.f, 0.f, 0.f, 0.f}, s2b = s2a;
            if (a.sc0 != nullptr && NIMG == 1 && nimg0 < a.B) {
                const size_t o = (size_t)nimg0 * a.C0 + cb;
                s1a = *reinterpret_cast<const f32x4*>(a.sc0 + o); s1b = *reinterpret_cast<const f32x4*>(a.sc0 + o + 4);
                s2a = *reinterpret_cast<const f32x4*>(a.sh0 + o); s2b = *reinterpret_cast<const f32x4*>(a.sh0 + o + 4);
            }
#pragma unroll
            for (int it = 0; it < MAXU; ++it) {
                const int u = tid + it * kBlock;
                if (u < P) {
                    f32x4 va, vb;
                    pv[it].get(va, vb);
                    if (a.sc0 != nullptr && goff[it] >= 0) {
                        if (NIMG != 1) {
                            const size_t o = (size_t)(nimg0 + (int)((imgbits >> (4 * it)) & 15)) * a.C0 + cb;
                            s1a = *reinterpret_cast<const f32x4*>(a.sc0 + o); s1b = *reinterpret_cast<const f32x4*>(a.sc0 + o + 4);
                            s2a = *reinterpret_cast<const f32x4*>(a.sh0 + o); s2b = *reinterpret_cast<const f32x4*>(a.sh0 + o + 4);
                        }
                        va = va * s1a + s2a; vb = vb * s1b + s2b;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            va[e] = fmaxf(va[e], va[e] * a.slope);
                            vb[e] = fmaxf(vb[e], vb[e] * a.slope);
                        }
                    }
                    half8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const _Float16 ha = (_Float16)va[e], hb = (_Float16)vb[e];
                        hi[e] = ha; hi[e + 4] = hb;
                        if (NP == 3) { lo[e] = (_Float16)(va[e] - (float)ha); lo[e + 4] = (_Float16)(vb[e] - (float)hb); }
                    }
                    *reinterpret_cast<half8*>(sA + lrec[it]) = hi;
                    if (NP == 3) *reinterpret_cast<half8*>(sA + lrec[it] + 16) = lo;
                }
            }
        }
        {
            constexpr int WU = 5 * BN * 4;
            // one contiguous [k-step][column][16 hi | 16 lo] block per (chunk, column tile)
            const uint4* wsrc = reinterpret_cast<const uint4*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * (5 * BN * 4);
#pragma unroll
            for (int it = 0; it < (WU + kBlock - 1) / kBlock; ++it) {
                const int idx = tid + it * kBlock;
                if ((WU % kBlock == 0 || idx < WU) && (NP == 3 || (idx & 2) == 0))
                    *reinterpret_cast<uint4*>(sB + (idx >> 2) * kRec + (idx & 3) * 16) = wsrc[idx];
            }
        }
        __syncthreads();
        if (ch + 1 < kend) prefetch(ch + 1);

        f32x16 acc_c[2][NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc_c[mt][nt][i] = 0.f;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            half8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                ah[mt] = *reinterpret_cast<const half8*>(sA + abase[mt] + tofl[s]);
                if (NP == 3) al[mt] = *reinterpret_cast<const half8*>(sA + abase[mt] + tofl[s] + 16);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bh[nt] = *reinterpret_cast<const half8*>(sB + (s * BN + nt * 32) * kRec + bbase);
                if (NP == 3) bl[nt] = *reinterpret_cast<const half8*>(sB + (s * BN + nt * 32) * kRec + bbase + 32);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    { if (NP == 3) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc_c[mt][nt], 0, 0, 0); }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    { if (NP == 3) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc_c[mt][nt], 0, 0, 0); }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc_c[mt][nt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc_t[mt][nt] += acc_c[mt][nt];
    }
    split_epilogue<BN, ST>(a, acc_t, smem8, n0col, nimg0, ty0, tx0, tpi, tin);
}

}  // namespace ts2d

