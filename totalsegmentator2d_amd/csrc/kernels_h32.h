// "mixed fp16" 3x3 stride-1 convolution for fp16-STORED activations (precision mode f16; BASELINE configs 3/5): one
// v_mfma_f32_32x32x16_f16 product per MAC, fp32 accumulation, fp32 InstanceNorm statistics.  Unlike the split kernel run with
// NP = 1 (16 channels per chunk, half of every LDS record unused) a chunk here is 32 input channels: the 80-B LDS record holds
// two real k-steps [ch 0..15 | ch 16..31 | pad], so the barriers, the weight staging and the fragment reads are amortised over
// twice the MACs, and the staging arithmetic works on packed fp16 (norm_lrelu_pk below: 2 VALU operations per element).
// Requires C0 % 32 == 0 and C1 % 32 == 0 (the canonical net; narrower test nets use conv3x3_f16x3<.., _Float16, 1>).
// Tile geometry, XCD-aware block map, split-K and epilogue are those of conv3x3_f16x3 (kernels_f16x3.h).
#pragma once
#include "kernels_f16x3.h"

namespace ts2d {

// Eight fp16 values x (four VGPRs) -> LeakyReLU(x * s + t) as eight fp16 values.  v_fma_mix computes in fp32 from the fp16
// operand and rounds once to fp16; the negative branch is slope16 * y in packed fp16 (slope16 = fp16(slope): 2e-4 relative on
// values that are already scaled by 0.01 - far below the fp16 rounding of the result).  One statement, four interleaved chains.
__device__ __forceinline__ uint4 norm_lrelu_8(uint4 x, f32x4 sa, f32x4 sb, f32x4 ta, f32x4 tb, unsigned slope2) {
    unsigned y0, y1, y2, y3, n0, n1, n2, n3;
    asm volatile(
        "v_fma_mixlo_f16 %0, %8, %12, %20 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %9, %14, %22 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %10, %16, %24 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %11, %18, %26 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %8, %13, %21 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %9, %15, %23 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %10, %17, %25 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %11, %19, %27 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_pk_mul_f16 %4, %0, %28\n\t"
        "v_pk_mul_f16 %5, %1, %28\n\t"
        "v_pk_mul_f16 %6, %2, %28\n\t"
        "v_pk_mul_f16 %7, %3, %28\n\t"
        "v_pk_max_f16 %0, %0, %4\n\t"
        "v_pk_max_f16 %1, %1, %5\n\t"
        "v_pk_max_f16 %2, %2, %6\n\t"
        "v_pk_max_f16 %3, %3, %7"
        : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)
        : "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w),
          "v"(sa[0]), "v"(sa[1]), "v"(sa[2]), "v"(sa[3]), "v"(sb[0]), "v"(sb[1]), "v"(sb[2]), "v"(sb[3]),
          "v"(ta[0]), "v"(ta[1]), "v"(ta[2]), "v"(ta[3]), "v"(tb[0]), "v"(tb[1]), "v"(tb[2]), "v"(tb[3]),
          "v"(slope2));
    return uint4{y0, y1, y2, y3};
}

// Tiles lie inside ONE image (lgNIMG == 0: every level down to 16x16); the engine keeps conv3x3_f16x3<.., _Float16, 1> for the
// multi-image tiles of the 8x8 / 4x4 levels.
template <int BN, int MAXU>
__global__ __launch_bounds__(kBlock, 2) void conv3x3_h32(const ConvArgs a) {
    constexpr int NT = BN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];

    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = bid >> 3;
    const int qm = q8 / a.n_ctiles;                 // (any tile count, any tile shape: round 5)
    const int mtile = qm * 8 + xcd;
    const int ctile = q8 - qm * a.n_ctiles;
    if (mtile >= a.n_mtiles) return;
    const int n0col = ctile * BN;

    const int tpi = a.tiles_x * a.tiles_y;
    const int nimg0 = mtile / tpi, tin = mtile - nimg0 * tpi;
    const int tyi = tin / a.tiles_x, txi = tin - tyi * a.tiles_x;
    const int ty0 = tyi * a.TH, tx0 = txi * a.TW;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int P = a.PH * a.PW;
    unsigned char* sA = smem8;                 // [P patch pixels][80 B]
    unsigned char* sB = smem8 + P * kRec;      // [tap 9][BN columns][80 B]

    // ---- staging plan: unit u = (patch pixel u >> 2, channel octet u & 3); kBlock % 4 == 0: the octet is per-thread constant.
    //      Buffer loads relative to the image base: a padding pixel gets an out-of-range offset and is never staged - its LDS
    //      record is zeroed once, here, and stays zero for every chunk.
    unsigned poff[MAXU];                       // pixel index inside the image, or ~0u (padding / no unit)
    const int total = P * 4;
    const float inv_pw = 1.0f / (float)a.PW;
    const int oct = (tid & 3) * 8;
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int u = tid + it * kBlock;
        unsigned g = ~0u;
        if (u < total) {
            const int pp = u >> 2;
            const int py = (int)(((float)pp + 0.5f) * inv_pw), px = pp - py * a.PW;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) g = (unsigned)(iy * a.Win + ix);
            else *reinterpret_cast<uint4*>(sA + pp * kRec + (u & 3) * 16) = uint4{0u, 0u, 0u, 0u};
        }
        poff[it] = g;
    }

    int abase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = 64 * w + 32 * mt + r;
        int il, ty, tx;
        tile_row(a, m, il, ty, tx);
        abase[mt] = (il == 0 ? (ty * a.PW + tx) * kRec : 0) + 16 * h;      // (rows past TH * TW: a valid dummy record)
    }
    const int bbase = r * kRec + 16 * h;

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const int nchunks = (a.C0 + a.C1) / 32;
    const size_t img_px = (size_t)a.Hin * a.Win;
    // one buffer descriptor per source tensor, based at this tile's image (wave-uniform); 2 bytes per element
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(reinterpret_cast<const _Float16*>(a.src0)) + (size_t)nimg0 * img_px * a.C0,
                                                       0, (int)(img_px * a.C0 * 2), 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(reinterpret_cast<const _Float16*>(a.src1 ? a.src1 : a.src0)) + (size_t)nimg0 * img_px * a.C1,
                                                       0, (int)(a.src1 ? img_px * a.C1 * 2 : 0), 0x00020000);
    unsigned vo0[MAXU], vo1[MAXU];             // per-unit byte offsets (chunk offset goes into the scalar offset)
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        vo0[it] = poff[it] == ~0u ? 0x80000000u : (poff[it] * (unsigned)a.C0 + oct) * 2u;
        vo1[it] = poff[it] == ~0u ? 0x80000000u : (poff[it] * (unsigned)a.C1 + oct) * 2u;
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pv[MAXU];                            // raw fp16 patch values of the next chunk (in flight during the MFMAs)
    f32x4 nsa = f32x4{1.f, 1.f, 1.f, 1.f}, nsb = nsa, nta = f32x4{0.f, 0.f, 0.f, 0.f}, ntb = nta;   // ... and its scale / shift
    const _Float16 slope_h = (_Float16)a.slope;
    const unsigned slope2 = (unsigned)__builtin_bit_cast(unsigned short, slope_h) * 0x10001u;

    const int kper = (nchunks + a.ksplit - 1) / a.ksplit;
    const int kbeg = (int)blockIdx.y * kper, kend = (kbeg + kper < nchunks) ? kbeg + kper : nchunks;
    auto prefetch = [&](int ch) {
        int cb = ch * 32;
        if (cb < a.C0) {
#pragma unroll
            for (int it = 0; it < MAXU; ++it) pv[it] = __builtin_amdgcn_raw_buffer_load_b128(rs0, vo0[it], cb * 2, 0);
            if (a.sc0 != nullptr) {
                const float* ps = a.sc0 + (size_t)nimg0 * a.C0 + cb + oct; const float* pt = a.sh0 + (size_t)nimg0 * a.C0 + cb + oct;
                nsa = *reinterpret_cast<const f32x4*>(ps); nsb = *reinterpret_cast<const f32x4*>(ps + 4);
                nta = *reinterpret_cast<const f32x4*>(pt); ntb = *reinterpret_cast<const f32x4*>(pt + 4);
            }
        } else {
            cb -= a.C0;
#pragma unroll
            for (int it = 0; it < MAXU; ++it) pv[it] = __builtin_amdgcn_raw_buffer_load_b128(rs1, vo1[it], cb * 2, 0);
            if (a.sc1 != nullptr) {
                const float* ps = a.sc1 + (size_t)nimg0 * a.C1 + cb + oct; const float* pt = a.sh1 + (size_t)nimg0 * a.C1 + cb + oct;
                nsa = *reinterpret_cast<const f32x4*>(ps); nsb = *reinterpret_cast<const f32x4*>(ps + 4);
                nta = *reinterpret_cast<const f32x4*>(pt); ntb = *reinterpret_cast<const f32x4*>(pt + 4);
            }
        }
    };

    if (kbeg < kend) prefetch(kbeg);
    for (int ch = kbeg; ch < kend; ++ch) {
        const bool normed = (ch * 32 < a.C0) ? (a.sc0 != nullptr) : (a.sc1 != nullptr);
        __syncthreads();   // the previous chunk's MFMA reads of LDS are done
        // ---- weights of this chunk: loads issued first (named registers), latency hidden behind the patch conversion
        constexpr int WU = 9 * BN * 4, WIT = (WU + kBlock - 1) / kBlock;
        const uint4* wsrc = reinterpret_cast<const uint4*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * (9 * BN * 4);
        uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8;
#define TS2D_WLOAD(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU)) R = wsrc[idx]; }
        TS2D_WLOAD(0, w0) TS2D_WLOAD(1, w1) TS2D_WLOAD(2, w2) TS2D_WLOAD(3, w3) TS2D_WLOAD(4, w4)
        TS2D_WLOAD(5, w5) TS2D_WLOAD(6, w6) TS2D_WLOAD(7, w7) TS2D_WLOAD(8, w8)
#undef TS2D_WLOAD
        // ---- patch: InstanceNorm + LeakyReLU on the fly, fp16 in -> fp16 LDS records (padding records stay zero)
        if (normed) {
#pragma unroll
            for (int it = 0; it < MAXU; ++it) {
                const int u = tid + it * kBlock;
                if (poff[it] != ~0u)
                    *reinterpret_cast<uint4*>(sA + (u >> 2) * kRec + (u & 3) * 16) =
                        norm_lrelu_8(uint4{pv[it][0], pv[it][1], pv[it][2], pv[it][3]}, nsa, nsb, nta, ntb, slope2);
            }
        } else {           // the transposed-conv half of a decoder input: stored activated-free, staged as is
#pragma unroll
            for (int it = 0; it < MAXU; ++it) {
                const int u = tid + it * kBlock;
                if (poff[it] != ~0u)
                    *reinterpret_cast<uint4*>(sA + (u >> 2) * kRec + (u & 3) * 16) = uint4{pv[it][0], pv[it][1], pv[it][2], pv[it][3]};
            }
        }
        // ---- weight registers -> LDS records [tap][col][ch 0..15 | ch 16..31 | pad]
#define TS2D_WSTORE(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU)) \
            *reinterpret_cast<uint4*>(sB + (idx >> 2) * kRec + (idx & 3) * 16) = R; }
        TS2D_WSTORE(0, w0) TS2D_WSTORE(1, w1) TS2D_WSTORE(2, w2) TS2D_WSTORE(3, w3) TS2D_WSTORE(4, w4)
        TS2D_WSTORE(5, w5) TS2D_WSTORE(6, w6) TS2D_WSTORE(7, w7) TS2D_WSTORE(8, w8)
#undef TS2D_WSTORE
        __syncthreads();
        if (ch + 1 < kend) prefetch(ch + 1);       // HBM latency hides behind the MFMA phase

        __builtin_amdgcn_s_setprio(1);
        // software-pipelined over the 9 taps: fragments of tap t+1 are read while the MFMAs of tap t run
        half8 fa[2][2][2], fb[2][NT][2];            // [buffer][tile][k-step]
        auto load_frags = [&](int buf, int tap) {
            const int toff = ((tap / 3) * a.PW + (tap % 3)) * kRec;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                fa[buf][mt][0] = *reinterpret_cast<const half8*>(sA + abase[mt] + toff);
                fa[buf][mt][1] = *reinterpret_cast<const half8*>(sA + abase[mt] + toff + 32);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                fb[buf][nt][0] = *reinterpret_cast<const half8*>(sB + (tap * BN + nt * 32) * kRec + bbase);
                fb[buf][nt][1] = *reinterpret_cast<const half8*>(sB + (tap * BN + nt * 32) * kRec + bbase + 32);
            }
        };
        load_frags(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = tap & 1;
            if (tap + 1 < 9) load_frags(cur ^ 1, tap + 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][ks], fb[cur][nt][ks], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
    }

    split_epilogue_one<BN, _Float16>(a, acc, smem8, n0col, nimg0, ty0, tx0, tpi, tin);
}

}  // namespace ts2d
