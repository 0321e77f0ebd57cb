// conv3x3_f16x3_p: the stride-1 3x3 split kernel of kernels_f16x3_one.h (same tiling, staging pipeline, arithmetic and summation
// order - results are bit-identical) on "k-group major" LDS planes instead of 80-byte records.
//
// Round-1 counters of conv3x3_f16x3_one<64>: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 23 % - the 16-byte ds_write_b128 of the
// staging phases at an 80-byte record stride (a record = [16 hi | 16 lo | pad]) hit the same banks from several lanes.  Here
//   patch   plane[part hi,lo][h][pixel]      16-byte slots (8 channels), unpadded: 21.8 KB instead of 27.2 KB
//   weights plane[tap][part][h][column]      16-byte slots, unpadded: 36.9 KB instead of 46.1 KB
// so that (a) the 32 lanes of a lane half (one h) read 32 CONSECUTIVE slots for a 32x32x16 fragment - conflict-free
// ds_read_b128 at any alignment; (b) 8 consecutive lanes of the staging phase write 8 consecutive slots of one plane -
// conflict-free ds_write_b128; (c) every fragment address is lane base + immediate (PW = 34 is a compile-time constant: the
// kernel takes complete 8 x 32 tiles only; other geometries stay on conv3x3_f16x3_one / conv3x3_f16x3); (d) the weight block of
// a (chunk, column tile) is stored in HBM in LDS order and copied linearly.
// Measured (profiles/r02_sq_counters.txt): conflict ratio of the plane kernels conv3x3_res32 0.4 %, conv3x3s2_v2 0.0 %.
#pragma once
#include "kernels_f16x3_one.h"

namespace ts2d {

constexpr int kPPW = 34, kPP = 340, kPPlane = kPP * 16;      // patch 10 x 34 pixels, one 16-byte slot per pixel and plane

template <int BN, bool PFS>
__global__ __launch_bounds__(kBlock, BN == 32 ? 3 : 2) void conv3x3_f16x3_p(const ConvArgs a) {
    constexpr int NT = BN / 32, MAXU = 3;
    constexpr int WTAP = 4 * BN * 16;                       // bytes per tap: [part][h][column]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];

    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = bid >> 3;
    const int qm = q8 >> a.lg_nct;                          // (power-of-two tilings only, the engine checks)
    const int mtile = qm * 8 + xcd;
    const int ctile = q8 - qm * a.n_ctiles;
    if (mtile >= a.n_mtiles) return;
    const int n0col = ctile * BN;

    const int tpi = a.tiles_x * a.tiles_y;
    const int nimg0 = mtile >> a.lg_tpi, tin = mtile - nimg0 * tpi;
    const int tyi = tin >> a.lg_tx, txi = tin - tyi * a.tiles_x;
    const int ty0 = tyi << 3, tx0 = txi << 5;               // TH = 8, TW = 32

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    unsigned char* sA = smem8;                              // [part][h][pixel] x 16 B
    unsigned char* sB = smem8 + 4 * kPPlane;                // [tap][part][h][column] x 16 B

    // ---- staging plan: a wave instruction covers 32 patch pixels x 2 channel octets: pixel = 32 (4 it + w) + (lane & 7) +
    //      8 (lane >> 4), octet = (lane >> 3) & 1 - 8 consecutive lanes write 8 consecutive slots of one plane.  Buffer loads
    //      relative to the image base: a padding pixel gets an out-of-range offset, its slots are zeroed once and never staged.
    const int octi = (lane >> 3) & 1, oct = octi * 8;
    unsigned poff[MAXU];
    int lw[MAXU];
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int pp = 32 * (4 * it + w) + (lane & 7) + 8 * (lane >> 4);
        unsigned g = ~0u;
        lw[it] = octi * kPPlane + pp * 16;
        if (pp < kPP) {
            const int py = pp / kPPW, px = pp - py * kPPW;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) g = (unsigned)(iy * a.Win + ix);
            else { *reinterpret_cast<uint4*>(sA + lw[it]) = uint4{0u, 0u, 0u, 0u};
                   *reinterpret_cast<uint4*>(sA + lw[it] + 2 * kPPlane) = uint4{0u, 0u, 0u, 0u}; }
        }
        poff[it] = g;
    }

    const int nchunks = (a.C0 + a.C1) / 16;
    const size_t img_px = (size_t)a.Hin * a.Win;
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0) + (size_t)nimg0 * img_px * a.C0, 0, (int)(img_px * a.C0 * 4), 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1 ? a.src1 : a.src0) + (size_t)nimg0 * img_px * a.C1, 0,
                                                       (int)(a.src1 ? img_px * a.C1 * 4 : 0), 0x00020000);
    unsigned vo0[MAXU], vo1[MAXU];
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        vo0[it] = poff[it] == ~0u ? 0x80000000u : (poff[it] * (unsigned)a.C0 + oct) * 4u;
        vo1[it] = poff[it] == ~0u ? 0x80000000u : (poff[it] * (unsigned)a.C1 + oct) * 4u;
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pv[MAXU][2];                         // raw fp32 patch values of the next chunk (in flight during the MFMAs)
    f32x4 nsa = f32x4{1.f, 1.f, 1.f, 1.f}, nsb = nsa, nta = f32x4{0.f, 0.f, 0.f, 0.f}, ntb = nta;   // ... and its scale / shift

    auto load_st = [&](int ch) {              // scale / shift of this thread's 8 channels (nullptr: not normalised)
        int cb = ch * 16;
        const float* ps = nullptr; const float* pt = nullptr;
        if (cb < a.C0) { if (a.sc0 != nullptr) { ps = a.sc0 + (size_t)nimg0 * a.C0 + cb + oct; pt = a.sh0 + (size_t)nimg0 * a.C0 + cb + oct; } }
        else { cb -= a.C0; if (a.sc1 != nullptr) { ps = a.sc1 + (size_t)nimg0 * a.C1 + cb + oct; pt = a.sh1 + (size_t)nimg0 * a.C1 + cb + oct; } }
        if (ps != nullptr) {
            nsa = *reinterpret_cast<const f32x4*>(ps); nsb = *reinterpret_cast<const f32x4*>(ps + 4);
            nta = *reinterpret_cast<const f32x4*>(pt); ntb = *reinterpret_cast<const f32x4*>(pt + 4);
        }
    };
    auto prefetch = [&](int ch) {
        int cb = ch * 16;
        if (cb < a.C0) {
#pragma unroll
            for (int it = 0; it < MAXU; ++it) {
                pv[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rs0, vo0[it], cb * 4, 0);
                pv[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rs0, vo0[it] + 16, cb * 4, 0);
            }
        } else {
            cb -= a.C0;
#pragma unroll
            for (int it = 0; it < MAXU; ++it) {
                pv[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rs1, vo1[it], cb * 4, 0);
                pv[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rs1, vo1[it] + 16, cb * 4, 0);
            }
        }
        if (PFS) load_st(ch);
    };
    if (nchunks > 0) prefetch(0);

    // ---- lane constants of the MFMA phase: output pixel m = 64 w + 32 mt + r = (row 2 w + mt, column r) of the 8 x 32 tile
    const int abase = h * kPPlane + ((2 * w) * kPPW + r) * 16;             // + mt * 34 * 16 + part * 2 * Plane + tap offset
    const int bbase = 4 * kPPlane + h * BN * 16 + r * 16;                  // + tap * WTAP + part * 2 * BN * 16 + nt * 512

    f32x16 acc_t[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_t[mt][nt][i] = 0.f;

    TS2D_PROF_DECL(a.prof);
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool normed = (ch * 16 < a.C0) ? (a.sc0 != nullptr) : (a.sc1 != nullptr);
        __syncthreads();   // the previous chunk's MFMA reads of LDS are done
        TS2D_STAMP_AT(a.prof, 1)
        if (!PFS) load_st(ch);
        // ---- weights of this chunk: one linear block in LDS order; loads issued first (named registers)
        constexpr int WU = 9 * BN * 4, WIT = (WU + kBlock - 1) / kBlock;
        const uint4* wsrc = reinterpret_cast<const uint4*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * (9 * BN * 4);
        uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8;
#define TS2D_WLOAD(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU)) R = wsrc[idx]; }
        TS2D_WLOAD(0, w0) TS2D_WLOAD(1, w1) TS2D_WLOAD(2, w2) TS2D_WLOAD(3, w3) TS2D_WLOAD(4, w4)
        TS2D_WLOAD(5, w5) TS2D_WLOAD(6, w6) TS2D_WLOAD(7, w7) TS2D_WLOAD(8, w8)
#undef TS2D_WLOAD
        TS2D_STAMP_AT(a.prof, 2)
        // ---- patch: InstanceNorm + LeakyReLU on the fly, fp32 in -> fp16 hi / lo planes (padding slots stay zero)
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            if (poff[it] != ~0u) {
                f32x4 va = __builtin_bit_cast(f32x4, pv[it][0]), vb = __builtin_bit_cast(f32x4, pv[it][1]);
                if (normed) {
                    va = va * nsa + nta; vb = vb * nsb + ntb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        va[e] = fmaxf(va[e], va[e] * a.slope);     // LeakyReLU (0 < slope < 1)
                        vb[e] = fmaxf(vb[e], vb[e] * a.slope);
                    }
                }
                uint4 hi, lo;
                split_hi_lo_8(va, vb, hi, lo);
                *reinterpret_cast<uint4*>(sA + lw[it]) = hi;
                *reinterpret_cast<uint4*>(sA + lw[it] + 2 * kPPlane) = lo;
            }
        }
        TS2D_STAMP_AT(a.prof, 4)
#define TS2D_WSTORE(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU)) \
            *reinterpret_cast<uint4*>(sB + idx * 16) = R; }
        TS2D_WSTORE(0, w0) TS2D_WSTORE(1, w1) TS2D_WSTORE(2, w2) TS2D_WSTORE(3, w3) TS2D_WSTORE(4, w4)
        TS2D_WSTORE(5, w5) TS2D_WSTORE(6, w6) TS2D_WSTORE(7, w7) TS2D_WSTORE(8, w8)
#undef TS2D_WSTORE
        TS2D_STAMP_AT(a.prof, 6)
        __syncthreads();
        TS2D_STAMP_AT(a.prof, 0)
        if (ch + 1 < nchunks) prefetch(ch + 1);       // HBM latency hides behind the MFMA phase

        // ---- 9 taps x (hi*lo + lo*hi + hi*hi) into a fresh accumulator (accuracy), added to acc_t;
        //      fragments of tap t+1 are read while the MFMAs of tap t run (BN = 64; BN = 32 keeps the plain loop: registers)
        f32x16 acc_c[2][NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc_c[mt][nt][i] = 0.f;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (BN == 64) {
            half8 fa[2][2][2], fb[2][NT][2];            // [buffer][tile][hi, lo]
#define TS2D_LOAD_FRAGS(BUF, TAP) { \
                constexpr int toff_ = (((TAP) / 3) * kPPW + ((TAP) % 3)) * 16; \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) { \
                    fa[BUF][mt][0] = *reinterpret_cast<const half8*>(smem8 + abase + mt * kPPW * 16 + toff_); \
                    fa[BUF][mt][1] = *reinterpret_cast<const half8*>(smem8 + abase + mt * kPPW * 16 + toff_ + 2 * kPPlane); } \
                _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { \
                    fb[BUF][nt][0] = *reinterpret_cast<const half8*>(smem8 + bbase + (TAP) * WTAP + nt * 512); \
                    fb[BUF][nt][1] = *reinterpret_cast<const half8*>(smem8 + bbase + (TAP) * WTAP + nt * 512 + 2 * BN * 16); } }
#define TS2D_TAP(TAP) { constexpr int cur = (TAP) & 1; \
                if constexpr ((TAP) + 1 < 9) TS2D_LOAD_FRAGS(cur ^ 1, (TAP) + 1) \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                    acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][1], fb[cur][nt][0], acc_c[mt][nt], 0, 0, 0); \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                    acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][1], acc_c[mt][nt], 0, 0, 0); \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                    acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][0], acc_c[mt][nt], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0); }
            TS2D_LOAD_FRAGS(0, 0)
            TS2D_TAP(0) TS2D_TAP(1) TS2D_TAP(2) TS2D_TAP(3) TS2D_TAP(4) TS2D_TAP(5) TS2D_TAP(6) TS2D_TAP(7) TS2D_TAP(8)
#undef TS2D_TAP
#undef TS2D_LOAD_FRAGS
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = ((tap / 3) * kPPW + (tap % 3)) * 16;
                half8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) al[mt] = *reinterpret_cast<const half8*>(smem8 + abase + mt * kPPW * 16 + toff + 2 * kPPlane);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bh[nt] = *reinterpret_cast<const half8*>(smem8 + bbase + tap * WTAP + nt * 512);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) ah[mt] = *reinterpret_cast<const half8*>(smem8 + abase + mt * kPPW * 16 + toff);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bl[nt] = *reinterpret_cast<const half8*>(smem8 + bbase + tap * WTAP + nt * 512 + 2 * BN * 16);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc_c[mt][nt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc_t[mt][nt] += acc_c[mt][nt];
    }

    TS2D_STAMP_AT(a.prof, 3)
    split_epilogue_one<BN, float>(a, acc_t, smem8, n0col, nimg0, ty0, tx0, tpi, tin);
    TS2D_STAMP_AT(a.prof, 5)
    TS2D_PROF_FLUSH(a.prof)
}

}  // namespace ts2d
