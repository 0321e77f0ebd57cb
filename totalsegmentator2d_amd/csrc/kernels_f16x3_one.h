// Split-fp16 3x3 stride-1 convolution for tiles that lie inside ONE image (lgNIMG == 0: every level down to 16x16) - the
// arithmetic, summation order and LDS layout of conv3x3_f16x3 (kernels_f16x3.h; results are bit-identical), with a leaner
// staging path: buffer loads relative to the image base (padding pixels are out-of-range offsets; their LDS records are zeroed
// once and never staged), one exec-mask region per unit, the InstanceNorm scale/shift of the next chunk prefetched with its
// patch data (PFS), and no per-unit statistics loads - which lets the compiler wait with a counted vmcnt for the patch data only
// while the weight loads of the chunk stay in flight behind the conversion (the generic kernel drains them with vmcnt(0)).
#pragma once
#include "kernels_f16x3.h"
#include "kernels_h32.h"

namespace ts2d {

// One chunk of the stride-1 contraction: 9 taps x (lo*hi + hi*lo + hi*hi) into a fresh accumulator, added to acc_t.
template <int BN>
__device__ __forceinline__ void split_mfma_chunk_s1(const unsigned char* sA, const unsigned char* sB, const int (&abase)[2], int bbase,
                                                    const ConvArgs& a, f32x16 (&acc_t)[2][BN / 32]) {
    constexpr int NT = BN / 32;
    f32x16 acc_c[2][NT];                        // fresh accumulator per 16-channel chunk (accuracy)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_c[mt][nt][i] = 0.f;
    __builtin_amdgcn_s_setprio(1);
    if constexpr (BN == 64) {
    // software-pipelined over the 9 taps: fragments of tap t+1 are read while the MFMAs of tap t run
    half8 fa[2][2][2], fb[2][NT][2];            // [buffer][tile][hi, lo]
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
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][1], fb[cur][nt][0], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][1], acc_c[mt][nt], 0, 0, 0);
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
        half8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) al[mt] = *reinterpret_cast<const half8*>(sA + abase[mt] + toff + 32);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bh[nt] = *reinterpret_cast<const half8*>(sB + (tap * BN + nt * 32) * kRec + bbase);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) ah[mt] = *reinterpret_cast<const half8*>(sA + abase[mt] + toff);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bl[nt] = *reinterpret_cast<const half8*>(sB + (tap * BN + nt * 32) * kRec + bbase + 32);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc_c[mt][nt], 0, 0, 0);
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

template <int BN, bool PFS>
__global__ __launch_bounds__(kBlock, 2) void conv3x3_f16x3_one(const ConvArgs a) {
    constexpr int NT = BN / 32, MAXU = 3;
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

    // ---- staging plan: unit u = (patch pixel u >> 1, channel octet u & 1); the octet is per-thread constant.
    //      Buffer loads relative to the image base: a padding pixel gets an out-of-range offset and is never staged - its LDS
    //      record is zeroed once, here, and stays zero for every chunk.
    unsigned poff[MAXU];                       // pixel index inside the image, or ~0u (padding / no unit)
    const int total = P * 2;
    const float inv_pw = 1.0f / (float)a.PW;
    const int oct = (tid & 1) * 8;
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int u = tid + it * kBlock;
        unsigned g = ~0u;
        if (u < total) {
            const int pp = u >> 1;
            const int py = (int)(((float)pp + 0.5f) * inv_pw), px = pp - py * a.PW;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) g = (unsigned)(iy * a.Win + ix);
            else { *reinterpret_cast<uint4*>(sA + pp * kRec + (u & 1) * 16) = uint4{0u, 0u, 0u, 0u};
                   *reinterpret_cast<uint4*>(sA + pp * kRec + (u & 1) * 16 + 32) = uint4{0u, 0u, 0u, 0u}; }
        }
        poff[it] = g;
    }

    const int nchunks = (a.C0 + a.C1) / 16;
    const size_t img_px = (size_t)a.Hin * a.Win;
    // one buffer descriptor per source tensor, based at this tile's image (wave-uniform)
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0) + (size_t)nimg0 * img_px * a.C0, 0, (int)(img_px * a.C0 * 4), 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1 ? a.src1 : a.src0) + (size_t)nimg0 * img_px * a.C1, 0,
                                                       (int)(a.src1 ? img_px * a.C1 * 4 : 0), 0x00020000);
    unsigned vo0[MAXU], vo1[MAXU];             // per-unit byte offsets (chunk offset goes into the scalar offset)
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        vo0[it] = poff[it] == ~0u ? 0x80000000u : (poff[it] * (unsigned)a.C0 + oct) * 4u;
        vo1[it] = poff[it] == ~0u ? 0x80000000u : (poff[it] * (unsigned)a.C1 + oct) * 4u;
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pv[MAXU][2];                         // raw fp32 patch values of the next chunk (in flight during the MFMAs)
    f32x4 nsa = f32x4{1.f, 1.f, 1.f, 1.f}, nsb = nsa, nta = f32x4{0.f, 0.f, 0.f, 0.f}, ntb = nta;   // ... and its scale / shift

    const int kper = (nchunks + a.ksplit - 1) / a.ksplit;
    const int kbeg = (int)blockIdx.y * kper, kend = (kbeg + kper < nchunks) ? kbeg + kper : nchunks;
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

    if (kbeg < kend) prefetch(kbeg);
    // (everything the first loads do not need comes after their issue)
    int abase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = 64 * w + 32 * mt + r;
        int il, ty, tx;
        tile_row(a, m, il, ty, tx);
        abase[mt] = (il == 0 ? (ty * a.PW + tx) * kRec : 0) + 16 * h;      // (rows past TH * TW: a valid dummy record)
    }
    const int bbase = r * kRec + 16 * h;

    f32x16 acc_t[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_t[mt][nt][i] = 0.f;

    for (int ch = kbeg; ch < kend; ++ch) {
        const bool normed = (ch * 16 < a.C0) ? (a.sc0 != nullptr) : (a.sc1 != nullptr);
        __syncthreads();   // the previous chunk's MFMA reads of LDS are done
        if (!PFS) load_st(ch);      // (register budget of the 3-workgroups-per-CU variant: loaded here, ahead of the weights)
        // ---- weights of this chunk: loads issued first (named registers), latency hidden behind the patch conversion
        constexpr int WU = 9 * BN * 4, WIT = (WU + kBlock - 1) / kBlock;
        const uint4* wsrc = reinterpret_cast<const uint4*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * (9 * BN * 4);
        uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8;
#define TS2D_WLOAD(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU)) R = wsrc[idx]; }
        TS2D_WLOAD(0, w0) TS2D_WLOAD(1, w1) TS2D_WLOAD(2, w2) TS2D_WLOAD(3, w3) TS2D_WLOAD(4, w4)
        TS2D_WLOAD(5, w5) TS2D_WLOAD(6, w6) TS2D_WLOAD(7, w7) TS2D_WLOAD(8, w8)
#undef TS2D_WLOAD
        // ---- patch: InstanceNorm + LeakyReLU on the fly, fp16 in -> fp16 LDS records (padding records stay zero)
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            const int u = tid + it * kBlock;
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
                unsigned char* d = sA + (u >> 1) * kRec + (u & 1) * 16;
                *reinterpret_cast<uint4*>(d) = hi;
                *reinterpret_cast<uint4*>(d + 32) = lo;
            }
        }
        // ---- weight registers -> LDS records [tap][col][16 hi | 16 lo | pad]
#define TS2D_WSTORE(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU)) \
            *reinterpret_cast<uint4*>(sB + (idx >> 2) * kRec + (idx & 3) * 16) = R; }
        TS2D_WSTORE(0, w0) TS2D_WSTORE(1, w1) TS2D_WSTORE(2, w2) TS2D_WSTORE(3, w3) TS2D_WSTORE(4, w4)
        TS2D_WSTORE(5, w5) TS2D_WSTORE(6, w6) TS2D_WSTORE(7, w7) TS2D_WSTORE(8, w8)
#undef TS2D_WSTORE
        __syncthreads();
        if (ch + 1 < kend) prefetch(ch + 1);       // HBM latency hides behind the MFMA phase

        split_mfma_chunk_s1<BN>(sA, sB, abase, bbase, a, acc_t);
    }

    split_epilogue_one<BN, float>(a, acc_t, smem8, n0col, nimg0, ty0, tx0, tpi, tin);
}

// ------------------------------------------------------------------------------------------------------------
// Stride-2 counterpart (the strided-conv downsample): arithmetic and LDS layout of conv3x3s2_f16x3, staging as above.
// ------------------------------------------------------------------------------------------------------------
template <int BN, bool PFS, bool PIPE, typename ST = float, int NP = 3>
__global__ __launch_bounds__(kBlock, 2) void conv3x3s2_f16x3_one(const ConvArgs a) {
    constexpr int NT = BN / 32, MAXU = 5, NL = sizeof(ST) == 4 ? 2 : 1;     // NL: 16-byte loads per unit (8 channels)
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

    const int P = a.PH * a.PW, PWe = (a.PW + 1) >> 1;
    unsigned char* sA = smem8;                 // [P patch pixels, even columns first][48 B]
    unsigned char* sB = smem8 + P * kRec8;     // [k-step 5][BN columns][80 B]

    // ---- staging plan: unit u = one patch pixel (8 channels per chunk).  Buffer loads relative to the image base: a padding
    //      pixel gets an out-of-range offset and is never staged - its LDS record is zeroed once, here.
    unsigned poff[MAXU];                       // pixel index inside the image, or ~0u (padding / no unit)
    int lrec[MAXU];
    const float inv_pw = 1.0f / (float)a.PW;
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int u = tid + it * kBlock;
        unsigned g = ~0u; int lr = 0;
        if (u < P) {
            const int py = (int)(((float)u + 0.5f) * inv_pw), px = u - py * a.PW;
            const int iy = 2 * ty0 - 1 + py, ix = 2 * tx0 - 1 + px;
            lr = (py * a.PW + ((px & 1) ? PWe + (px >> 1) : (px >> 1))) * kRec8;
            if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) g = (unsigned)(iy * a.Win + ix);
            else { *reinterpret_cast<uint4*>(sA + lr) = uint4{0u, 0u, 0u, 0u}; if (NP == 3) *reinterpret_cast<uint4*>(sA + lr + 16) = uint4{0u, 0u, 0u, 0u}; }
        }
        poff[it] = g; lrec[it] = lr;
    }

    int abase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = 64 * w + 32 * mt + r;
        int il, ty, tx;
        tile_row(a, m, il, ty, tx);
        abase[mt] = il == 0 ? (2 * ty * a.PW + tx) * kRec8 : 0;
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

    const int nchunks = a.C0 / 8;              // the strided conv never reads a concat
    const size_t img_px = (size_t)a.Hin * a.Win;
    // one buffer descriptor per source tensor, based at this tile's image (wave-uniform)
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(reinterpret_cast<const ST*>(a.src0)) + (size_t)nimg0 * img_px * a.C0, 0,
                                                       (int)(img_px * a.C0 * sizeof(ST)), 0x00020000);
    unsigned vo0[MAXU];                        // per-unit byte offsets (chunk offset goes into the scalar offset)
#pragma unroll
    for (int it = 0; it < MAXU; ++it) vo0[it] = poff[it] == ~0u ? 0x80000000u : poff[it] * (unsigned)a.C0 * (unsigned)sizeof(ST);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pv[MAXU][NL];                        // raw patch values of the next chunk (in flight during the MFMAs)
    const _Float16 slope_h = (_Float16)a.slope;
    const unsigned slope2 = (unsigned)__builtin_bit_cast(unsigned short, slope_h) * 0x10001u;
    f32x4 nsa = f32x4{1.f, 1.f, 1.f, 1.f}, nsb = nsa, nta = f32x4{0.f, 0.f, 0.f, 0.f}, ntb = nta;   // ... and its scale / shift

    const int kper = (nchunks + a.ksplit - 1) / a.ksplit;
    const int kbeg = (int)blockIdx.y * kper, kend = (kbeg + kper < nchunks) ? kbeg + kper : nchunks;
    auto load_st = [&](int ch) {              // scale / shift of the chunk's 8 channels
        if (a.sc0 != nullptr) {
            const float* ps = a.sc0 + (size_t)nimg0 * a.C0 + ch * 8; const float* pt = a.sh0 + (size_t)nimg0 * a.C0 + ch * 8;
            nsa = *reinterpret_cast<const f32x4*>(ps); nsb = *reinterpret_cast<const f32x4*>(ps + 4);
            nta = *reinterpret_cast<const f32x4*>(pt); ntb = *reinterpret_cast<const f32x4*>(pt + 4);
        }
    };
    auto prefetch = [&](int ch) {
#pragma unroll
        for (int it = 0; it < MAXU; ++it)
#pragma unroll
            for (int l = 0; l < NL; ++l) pv[it][l] = __builtin_amdgcn_raw_buffer_load_b128(rs0, vo0[it] + 16 * l, ch * 8 * (int)sizeof(ST), 0);
        if (PFS) load_st(ch);
    };

    if (kbeg < kend) prefetch(kbeg);
    for (int ch = kbeg; ch < kend; ++ch) {
        const bool normed = a.sc0 != nullptr;
        __syncthreads();   // the previous chunk's MFMA reads of LDS are done
        if (!PFS) load_st(ch);      // (register budget of the 3-workgroups-per-CU variant: loaded here, ahead of the weights)
        // ---- weights of this chunk: loads issued first (named registers), latency hidden behind the patch conversion
        constexpr int WU = 5 * BN * 4, WIT = (WU + kBlock - 1) / kBlock;
        const uint4* wsrc = reinterpret_cast<const uint4*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * (5 * BN * 4);
        uint4 w0, w1, w2, w3, w4;
#define TS2D_WLOAD(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU) && (NP == 3 || (idx & 2) == 0)) R = wsrc[idx]; }
        TS2D_WLOAD(0, w0) TS2D_WLOAD(1, w1) TS2D_WLOAD(2, w2) TS2D_WLOAD(3, w3) TS2D_WLOAD(4, w4)
#undef TS2D_WLOAD
        // ---- patch: InstanceNorm + LeakyReLU on the fly, fp16 in -> fp16 LDS records (padding records stay zero)
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            if (poff[it] != ~0u) {
                if constexpr (sizeof(ST) == 4) {
                    f32x4 va = __builtin_bit_cast(f32x4, pv[it][0]), vb = __builtin_bit_cast(f32x4, pv[it][NL - 1]);
                    if (normed) {
                        va = va * nsa + nta; vb = vb * nsb + ntb;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            va[e] = fmaxf(va[e], va[e] * a.slope);     // LeakyReLU (0 < slope < 1)
                            vb[e] = fmaxf(vb[e], vb[e] * a.slope);
                        }
                    }
                    uint4 hi, lo;
                    split_hi_lo_8(va, vb, hi, lo);      // (fp32 storage: NP == 3)
                    *reinterpret_cast<uint4*>(sA + lrec[it]) = hi;
                    if (NP == 3) *reinterpret_cast<uint4*>(sA + lrec[it] + 16) = lo;
                } else {
                    uint4 x = uint4{pv[it][0][0], pv[it][0][1], pv[it][0][2], pv[it][0][3]};
                    if (normed) x = norm_lrelu_8(x, nsa, nsb, nta, ntb, slope2);
                    *reinterpret_cast<uint4*>(sA + lrec[it]) = x;
                }
            }
        }
        // ---- weight registers -> LDS records [k-step][col][16 hi | 16 lo | pad]
#define TS2D_WSTORE(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU) && (NP == 3 || (idx & 2) == 0)) \
            *reinterpret_cast<uint4*>(sB + (idx >> 2) * kRec + (idx & 3) * 16) = R; }
        TS2D_WSTORE(0, w0) TS2D_WSTORE(1, w1) TS2D_WSTORE(2, w2) TS2D_WSTORE(3, w3) TS2D_WSTORE(4, w4)
#undef TS2D_WSTORE
        __syncthreads();
        if (ch + 1 < kend) prefetch(ch + 1);       // HBM latency hides behind the MFMA phase

        f32x16 acc_c[2][NT];                        // fresh accumulator per 8-channel chunk (accuracy)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc_c[mt][nt][i] = 0.f;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (PIPE) {
        // software-pipelined over the 5 k-steps: fragments of step s+1 are read while the MFMAs of step s run
        half8 fa[2][2][2], fb[2][NT][2];            // [buffer][tile][hi, lo]
        auto load_frags = [&](int buf, int s) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                fa[buf][mt][0] = *reinterpret_cast<const half8*>(sA + abase[mt] + tofl[s]);
                if (NP == 3) fa[buf][mt][1] = *reinterpret_cast<const half8*>(sA + abase[mt] + tofl[s] + 16);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                fb[buf][nt][0] = *reinterpret_cast<const half8*>(sB + (s * BN + nt * 32) * kRec + bbase);
                if (NP == 3) fb[buf][nt][1] = *reinterpret_cast<const half8*>(sB + (s * BN + nt * 32) * kRec + bbase + 32);
            }
        };
        load_frags(0, 0);
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int cur = s & 1;
            if (s + 1 < 5) load_frags(cur ^ 1, s + 1);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    { if (NP == 3) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][1], fb[cur][nt][0], acc_c[mt][nt], 0, 0, 0); }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    { if (NP == 3) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][1], acc_c[mt][nt], 0, 0, 0); }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][0], acc_c[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        } else {
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            half8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) if (NP == 3) al[mt] = *reinterpret_cast<const half8*>(sA + abase[mt] + tofl[s] + 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bh[nt] = *reinterpret_cast<const half8*>(sB + (s * BN + nt * 32) * kRec + bbase);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) ah[mt] = *reinterpret_cast<const half8*>(sA + abase[mt] + tofl[s]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) if (NP == 3) bl[nt] = *reinterpret_cast<const half8*>(sB + (s * BN + nt * 32) * kRec + bbase + 32);
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

    split_epilogue_one<BN, ST>(a, acc_t, smem8, n0col, nimg0, ty0, tx0, tpi, tin);
}

}  // namespace ts2d
