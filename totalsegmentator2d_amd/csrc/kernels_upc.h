// Decoder block entry without the upsampled tensor (SURVEY K5 + K6: ConvTranspose2d 2x2 stride 2, torch.cat((up, skip), 1), Conv2d
// 3x3): the transposed conv and the "up" half of the 3x3 conv are COMPOSED into one convolution over the coarse tensor.
//
//   up[cu][2i+a][2j+b] = sum_cb x[cb][i][j] WT[cb][cu][a][b] + bT[cu]        (no norm / activation follows the transposed conv)
//   y[co][Y][X]        = sum_{ky,kx} sum_cu W3[co][cu][ky][kx] up[cu][Y+ky-1][X+kx-1] + (skip half) + b3[co]
// For an output pixel of parity (A, B) = (Y & 1, X & 1) the three taps of a row/column land on TWO coarse pixels, so the up half is
// a 2x2 convolution over the coarse image with parity-specific weights
//   Weff[A][B][dI][dJ][co][cb] = sum_{ky in S(A,dI)} sum_{kx in S(B,dJ)} sum_cu W3[co][cu][ky][kx] WT[cb][cu][a(A,ky)][b(B,kx)]
// (composed on the host in fp64, then split into fp16 hi + lo like every weight).  K per output = 4 Cb instead of 9 Cu, the
// upsampled tensor is never written or read (at level 0: 67 MB per slice of HBM traffic), and the ConvTranspose2d launch is
// gone.  Zero padding of `up` = zero padding of the coarse patch (an out-of-image up pixel maps to an out-of-image coarse pixel);
// the transposed conv's bias reaches an output through the in-image taps only, which gives nine bias variants per channel
// (top / middle / bottom x left / middle / right), also prepared on the host.
//
// Kernel: 256 threads, 8 x 32 output pixels x BN channels, 2-3 workgroups per CU (LDS: the phase-2 image, 62.5 KB at BN = 64).  The M dimension is grouped BY PARITY: wave w
// owns parity class (A, B) = (w >> 1, w & 1), its two 32-row MFMA tiles are the 4 x 16 coarse positions (I, J) of the tile, i.e.
// output pixels (2I + A, 2J + B).  Phase 1 walks the coarse channels in chunks of 32 (2 k-steps x 4 "taps" (dI, dJ) per wave: 96 MFMAs
// for BN = 64 between two barriers), phase 2 the skip channels with the ordinary 9 taps (patch columns stored even-first / odd-second so that the
// stride-2 pixel walk of a fragment is a walk over consecutive LDS slots).  k-group-major planes as in kernels_f16x3_qp.h; row
// pitches (32 / 40 slots) chosen so that the two 16-lane runs of a fragment read fall on disjoint slot residues.
#pragma once
#include "kernels_f16x3_one.h"

namespace ts2d {

struct UpcArgs {
    const float* xc; const float* scc; const float* shc; int Cb;      // coarse tensor [B, H/2, W/2, Cb] + its scale / shift
    const float* xs; const float* scs; const float* shs; int Cs;      // skip tensor [B, H, W, Cs] + its scale / shift
    const void* wc;        // composed weights [chunk Cb/16][column tile][16 = (A,B,dI,dJ)][hi,lo][h][column][8 halves]
    const void* wk;        // skip-half 3x3 weights [chunk Cs/16][column tile][9 taps][hi,lo][h][column][8 halves]
    const float* bvar;     // [9 = (ry, rx)][Cout]: b3 + the transposed conv's bias through the in-image taps
    const float* oscale;   // 1 / (common power-of-two pre-scale of wc and wk)
    float* dst; float* part;
    int B, H, W, Cout;     // output geometry (H % 8 == 0, W % 32 == 0)
    int tiles_x, tiles_y, n_mtiles, n_ctiles;
    int TH, TW;            // FLEX instances: output tile TH x TW (both even, TH * TW <= 256); the fixed-tile instances walk 8 x 32 / 16 x 32
    float inv_twc;         // 1 / (TW / 2)
    float slope;
    unsigned long long* prof;   // diagnostic: phase cycle counters (6 entries) or nullptr
    int dbg;                    // experiment switches (TS2D_DBG; 0 in production)
};

constexpr int kUcPitch = 32, kUcSlots = 6 * kUcPitch, kUcPlane = kUcSlots * 16;        // coarse patch: 6 rows x 18 (pitch 32) slots
constexpr int kUsPitch = 40, kUsSlots = 10 * kUsPitch, kUsPlane = kUsSlots * 16;       // skip patch: 10 rows x (17 even | 17 odd at +20)

// Geometry of a composed tile.  FLEX = false: the 8 x 32 tile of rounds 2-4, every number a compile-time constant.  FLEX (round 5): the
// tile follows the level's extent - TH x TW output pixels, both even, at most 256 - so that a level of 80 x 48 or 28 x 36 pixels
// runs composed too (10 x 24 / 14 x 18 tiles) instead of as transposed conv + conv (the reference runs whatever patch size
// plans.json names: ts2d/core/inference/prediction_worker.py:76-77).  A parity class of the tile is (TH / 2) x (TW / 2) <= 64 coarse
// positions = the wave's two M tiles, row rho -> (I, J) = (rho / TWc, rho % TWc); both patches are stored densely (pitch = width).
// Budgets (the engine checks): coarse patch (TH/2 + 2)(TW/2 + 2) <= 128 pixels (one staging unit per thread pair), skip patch
// (TH + 2)(TW + 2) <= 384 pixels (three units).
template <bool FLEX>
struct UpcGeo {
    int TH, TW, THc, TWc, C1W, C1N, P1, S2W, S2N, NE, P2, HO;
    __device__ __forceinline__ UpcGeo(const UpcArgs& a) {
        TH = FLEX ? a.TH : 8; TW = FLEX ? a.TW : 32; THc = TH >> 1; TWc = TW >> 1;
        C1W = TWc + 2; C1N = (THc + 2) * C1W; P1 = FLEX ? C1W : kUcPitch;            // coarse patch: width, pixels, LDS row pitch
        S2W = TW + 2; S2N = (TH + 2) * S2W; NE = TWc + 1;                              // skip patch: width, pixels, even (= odd) columns per row
        P2 = FLEX ? S2W : kUsPitch; HO = FLEX ? NE : 20;                                // ... LDS row pitch, offset of the odd columns
    }
};

template <int BN, bool FLEX = false>
__global__ __launch_bounds__(kBlock, BN == 32 ? 3 : 2) void conv3x3_upc(const UpcArgs a) {
    constexpr int NT = BN / 32;
    const UpcGeo<FLEX> G(a);
    constexpr int WT1 = 4 * BN * 16;                        // bytes per "tap": [part][h][column]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = bid >> 3;
    const int qm = q8 / a.n_ctiles;                 // (any tile count: round 5)
    const int mtile = qm * 8 + xcd;
    const int ctile = q8 - qm * a.n_ctiles;
    if (mtile >= a.n_mtiles) return;
    const int n0col = ctile * BN;
    const int tpi = a.tiles_x * a.tiles_y;
    const int nimg0 = mtile / tpi, tin = mtile - nimg0 * tpi;
    const int tyi = tin / a.tiles_x, txi = tin - tyi * a.tiles_x;
    const int ty0 = FLEX ? tyi * G.TH : tyi << 3, tx0 = FLEX ? txi * G.TW : txi << 5;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), pA = w >> 1, pB = w & 1;      // this wave's output parity
    const int r = lane & 31, h = lane >> 5;
    const int octi = (lane >> 3) & 1, oct = octi * 8;
    // FLEX: this lane's coarse position per M tile, (I, J) = (rho / TWc, rho % TWc) of row rho = 32 mt + r; rows past the class: (0, 0), masked later
    int fI[2] = {0, 0}, fJ[2] = {0, 0};
    if constexpr (FLEX) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int rho = 32 * mt + r;
            if (rho < G.THc * G.TWc) { fI[mt] = fdiv(rho, a.inv_twc); fJ[mt] = rho - fI[mt] * G.TWc; }
        }
    }

    // LDS: phase 1 [coarse planes 8 x kUcPlane: 32-channel chunks]; phase 2 [skip planes 4 x kUsPlane | weights] (same memory)
    unsigned char* sB2 = smem8 + 4 * kUsPlane;

    f32x16 acc_t[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_t[mt][nt][i] = 0.f;
    // in-kernel phase stamps (a.prof != nullptr: diagnostic runs, scripts/gpu_upc_phases.py): shader-clock cycles of wave 0 spent in
    // [0] phase-1 staging (barrier to barrier), [1] phase-1 MFMAs (+ the wait at the next barrier), [2] / [3] the same for phase 2,
    // [4] epilogue: bias + output stores issued, [5] first LDS barrier + partial sums to LDS, [6] second barrier; [7] workgroups
    TS2D_PROF_DECL(a.prof);
#define TS2D_STAMP(I) TS2D_STAMP_AT(a.prof, I)

    // =================================================================================== phase 1: composed up half (coarse tensor)
    {
        const int Hc = a.H >> 1, Wc = a.W >> 1;
        // staging: one unit per thread: slot = 32 w + (lane & 7) + 8 (lane >> 4) of the 6 x 18 patch (row-major, 108 pixels), octet
        const int pp = 32 * w + (lane & 7) + 8 * (lane >> 4);
        const int py = pp / G.C1W, px = pp - py * G.C1W;
        const int iy = (ty0 >> 1) - 1 + py, ix = (tx0 >> 1) - 1 + px;
        // LDS planes of a 32-channel chunk: [k-step 2][part hi,lo][h][slot]: plane (ks, part, h) at ((ks * 2 + part) * 2 + h) * kUcPlane
        const int lw = octi * kUcPlane + (py * G.P1 + px) * 16;
        unsigned vo = 0x80000000u;
        if (pp < G.C1N) {
            if (iy >= 0 && iy < Hc && ix >= 0 && ix < Wc) vo = (unsigned)(((iy * Wc + ix) * a.Cb + oct) * 4);
            else {
#pragma unroll
                for (int pl = 0; pl < 4; ++pl) *reinterpret_cast<uint4*>(smem8 + lw + 2 * pl * kUcPlane) = uint4{0u, 0u, 0u, 0u};
            }
        }
        const size_t img_px = (size_t)Hc * Wc;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xc) + (size_t)nimg0 * img_px * a.Cb, 0, (int)(img_px * a.Cb * 4), 0x00020000);
        const int nch = a.Cb / 32;                          // chunks of 32 coarse channels = 2 k-steps (the engine checks Cb % 32 == 0)
        u32x4 pv[2][2];
        auto prefetch = [&](int ch) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                pv[ks][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + ks * 64, ch * 128, 0);
                pv[ks][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + ks * 64 + 16, ch * 128, 0);
            }
        };
        prefetch(0);
        // fragments: M-tile row r = coarse position (I = 2 mt + (r >> 4), J = r & 15); tap (dI, dJ) reads coarse patch pixel
        // (I + pA + dI, J + pB + dJ)
        const int abase = h * kUcPlane + (((r >> 4) + pA) * kUcPitch + (r & 15) + pB) * 16;      // + ks * 4 planes + part * 2 planes + mt * 2 * pitch * 16 + (dI * pitch + dJ) * 16
        int ab1[2];          // (per M tile: the FLEX row map has no uniform offset between the two)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) ab1[mt] = FLEX ? h * kUcPlane + ((fI[mt] + pA) * G.P1 + fJ[mt] + pB) * 16 : abase + mt * 2 * kUcPitch * 16;
        // B fragments: the 4 taps of THIS wave's parity - no other wave of the workgroup reads them, so they do not go through
        // LDS (staging all 16 taps made this phase LDS-bound: 65 KB of weight writes per 48 MFMAs of a wave) but straight from
        // L2 into a ring of 4 tap sets, each reloaded for the next k-step right after its MFMAs are issued (4 taps of lookahead).
        // The weight image is already in fragment order: 16 bytes per lane, 512 contiguous bytes per lane half.
        const unsigned char* wgl = reinterpret_cast<const unsigned char*>(a.wc) + ((size_t)ctile * 16 + w * 4) * WT1 + h * BN * 16 + r * 16;
        const size_t wchunk = (size_t)a.n_ctiles * 16 * WT1;       // bytes per 16-channel k-step
        const int nks = 2 * nch;
        half8 rb[4][NT][2];
#pragma unroll
        for (int tap = 0; tap < 4; ++tap)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                rb[tap][nt][0] = *reinterpret_cast<const half8*>(wgl + tap * WT1 + nt * 512);
                rb[tap][nt][1] = *reinterpret_cast<const half8*>(wgl + tap * WT1 + nt * 512 + 2 * BN * 16);
            }
        const bool normed = a.scc != nullptr;
        for (int ch = 0; ch < nch; ++ch) {
            __syncthreads();
            TS2D_STAMP(1)
            if (vo != 0x80000000u) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    f32x4 va = __builtin_bit_cast(f32x4, pv[ks][0]), vb = __builtin_bit_cast(f32x4, pv[ks][1]);
                    if (normed) {
                        const float* ps = a.scc + (size_t)nimg0 * a.Cb + ch * 32 + ks * 16 + oct; const float* pt = a.shc + (size_t)nimg0 * a.Cb + ch * 32 + ks * 16 + oct;
                        const f32x4 nsa = *reinterpret_cast<const f32x4*>(ps), nsb = *reinterpret_cast<const f32x4*>(ps + 4);
                        const f32x4 nta = *reinterpret_cast<const f32x4*>(pt), ntb = *reinterpret_cast<const f32x4*>(pt + 4);
                        va = va * nsa + nta; vb = vb * nsb + ntb;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { va[e] = fmaxf(va[e], va[e] * a.slope); vb[e] = fmaxf(vb[e], vb[e] * a.slope); }
                    }
                    uint4 hi, lo;
                    split_hi_lo_8(va, vb, hi, lo);
                    *reinterpret_cast<uint4*>(smem8 + lw + ks * 4 * kUcPlane) = hi;
                    *reinterpret_cast<uint4*>(smem8 + lw + ks * 4 * kUcPlane + 2 * kUcPlane) = lo;
                }
            }
            __syncthreads();
            TS2D_STAMP(0)
            if (ch + 1 < nch) prefetch(ch + 1);
            f32x16 acc_c[2][NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc_c[mt][nt][i] = 0.f;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int knext = 2 * ch + ks + 1 < nks ? 2 * ch + ks + 1 : 2 * ch + ks;      // (last k-step: reloaded, never used - no branch)
                const unsigned char* wnext = wgl + (size_t)knext * wchunk;
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) {
                    const int toff = ks * 4 * kUcPlane + ((tap >> 1) * G.P1 + (tap & 1)) * 16;
                    half8 ah[2], al[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        ah[mt] = *reinterpret_cast<const half8*>(smem8 + ab1[mt] + toff);
                        al[mt] = *reinterpret_cast<const half8*>(smem8 + ab1[mt] + toff + 2 * kUcPlane);
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], rb[tap][nt][0], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], rb[tap][nt][1], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], rb[tap][nt][0], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        rb[tap][nt][0] = *reinterpret_cast<const half8*>(wnext + tap * WT1 + nt * 512);
                        rb[tap][nt][1] = *reinterpret_cast<const half8*>(wnext + tap * WT1 + nt * 512 + 2 * BN * 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks the reloads next to their use four taps later: kernels_up0.h)
                }
            }
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc_t[mt][nt] += acc_c[mt][nt];
        }
    }

    // =================================================================================== phase 2: skip half (ordinary 3x3 taps)
    {
        __syncthreads();                                    // phase-1 LDS reads are done: the memory is re-laid out
        constexpr int MAXU = 3;
        unsigned vo[MAXU];
        int lw[MAXU];
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            // unit enumeration: patch row, then its 17 even columns, then its 17 odd columns (LDS slots +20)
            const int q = 32 * (4 * it + w) + (lane & 7) + 8 * (lane >> 4);
            const int py = q / G.S2W, rem = q - py * G.S2W;
            const int half = rem >= G.NE ? 1 : 0, idx = rem - G.NE * half, px = 2 * idx + half;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            unsigned v = 0x80000000u;
            lw[it] = octi * kUsPlane + (py * G.P2 + G.HO * half + idx) * 16;
            if (q < G.S2N) {
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = (unsigned)(((iy * a.W + ix) * a.Cs + oct) * 4);
                else { *reinterpret_cast<uint4*>(smem8 + lw[it]) = uint4{0u, 0u, 0u, 0u};
                       *reinterpret_cast<uint4*>(smem8 + lw[it] + 2 * kUsPlane) = uint4{0u, 0u, 0u, 0u}; }
            }
            vo[it] = v;
        }
        const size_t img_px = (size_t)a.H * a.W;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xs) + (size_t)nimg0 * img_px * a.Cs, 0, (int)(img_px * a.Cs * 4), 0x00020000);
        const int nch = a.Cs / 16;
        u32x4 pv[MAXU][2];
        auto prefetch = [&](int ch) {
#pragma unroll
            for (int it = 0; it < MAXU; ++it) {
                pv[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[it], ch * 64, 0);
                pv[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[it] + 16, ch * 64, 0);
            }
        };
        prefetch(0);
        // fragments: row r = (I = 2 mt + (r >> 4), J = r & 15) -> output pixel (2I + pA, 2J + pB); tap (ky, kx) reads patch pixel
        // (2I + pA + ky, 2J + pB + kx): patch row 2I + pA + ky, column parity (pB + kx) & 1, index J + ((pB + kx) >> 1)
        const int abase = h * kUsPlane + ((2 * (r >> 4) + pA) * kUsPitch + (r & 15)) * 16;       // + mt * 4 * pitch * 16 + tap offset + part * 2 * Plane
        int ab2[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) ab2[mt] = FLEX ? h * kUsPlane + ((2 * fI[mt] + pA) * G.P2 + fJ[mt]) * 16 : abase + mt * 4 * kUsPitch * 16;
        const int bbase = 4 * kUsPlane + h * BN * 16 + r * 16;
        int tofs[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) tofs[kx] = ((((pB + kx) & 1) ? G.HO : 0) + ((pB + kx) >> 1)) * 16;
        for (int ch = 0; ch < nch; ++ch) {
            __syncthreads();
            TS2D_STAMP(3)
            f32x4 nsa = f32x4{1.f, 1.f, 1.f, 1.f}, nsb = nsa, nta = f32x4{0.f, 0.f, 0.f, 0.f}, ntb = nta;
            const bool normed = a.scs != nullptr;
            if (normed) {
                const float* ps = a.scs + (size_t)nimg0 * a.Cs + ch * 16 + oct; const float* pt = a.shs + (size_t)nimg0 * a.Cs + ch * 16 + oct;
                nsa = *reinterpret_cast<const f32x4*>(ps); nsb = *reinterpret_cast<const f32x4*>(ps + 4);
                nta = *reinterpret_cast<const f32x4*>(pt); ntb = *reinterpret_cast<const f32x4*>(pt + 4);
            }
            constexpr int WU = 9 * BN * 4, WIT = (WU + kBlock - 1) / kBlock;
            const uint4* wsrc = reinterpret_cast<const uint4*>(a.wk) + ((size_t)ch * a.n_ctiles + ctile) * WU;
            uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8;
#define TS2D_WL(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU)) R = wsrc[idx]; }
            TS2D_WL(0, w0) TS2D_WL(1, w1) TS2D_WL(2, w2) TS2D_WL(3, w3) TS2D_WL(4, w4) TS2D_WL(5, w5) TS2D_WL(6, w6) TS2D_WL(7, w7) TS2D_WL(8, w8)
#undef TS2D_WL
#pragma unroll
            for (int it = 0; it < MAXU; ++it) {
                if (vo[it] != 0x80000000u) {
                    f32x4 va = __builtin_bit_cast(f32x4, pv[it][0]), vb = __builtin_bit_cast(f32x4, pv[it][1]);
                    if (normed) {
                        va = va * nsa + nta; vb = vb * nsb + ntb;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { va[e] = fmaxf(va[e], va[e] * a.slope); vb[e] = fmaxf(vb[e], vb[e] * a.slope); }
                    }
                    uint4 hi, lo;
                    split_hi_lo_8(va, vb, hi, lo);
                    *reinterpret_cast<uint4*>(smem8 + lw[it]) = hi;
                    *reinterpret_cast<uint4*>(smem8 + lw[it] + 2 * kUsPlane) = lo;
                }
            }
#define TS2D_WS(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU)) *reinterpret_cast<uint4*>(sB2 + idx * 16) = R; }
            TS2D_WS(0, w0) TS2D_WS(1, w1) TS2D_WS(2, w2) TS2D_WS(3, w3) TS2D_WS(4, w4) TS2D_WS(5, w5) TS2D_WS(6, w6) TS2D_WS(7, w7) TS2D_WS(8, w8)
#undef TS2D_WS
            __syncthreads();
            TS2D_STAMP(2)
            if (ch + 1 < nch) prefetch(ch + 1);
            f32x16 acc_c[2][NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc_c[mt][nt][i] = 0.f;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
                const int toff = ky * G.P2 * 16 + tofs[kx];
                half8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    ah[mt] = *reinterpret_cast<const half8*>(smem8 + ab2[mt] + toff);
                    al[mt] = *reinterpret_cast<const half8*>(smem8 + ab2[mt] + toff + 2 * kUsPlane);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bh[nt] = *reinterpret_cast<const half8*>(smem8 + bbase + tap * WT1 + nt * 512);
                    bl[nt] = *reinterpret_cast<const half8*>(smem8 + bbase + tap * WT1 + nt * 512 + 2 * BN * 16);
                }
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
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc_t[mt][nt] += acc_c[mt][nt];
        }
    }

    TS2D_STAMP(3)
    // =================================================================================== epilogue: scatter by parity, statistics
    // C/D map: column = lane & 31 (channel), row rho = (i & 3) + 8 (i >> 2) + 4 h -> (I = 2 mt + (rho >> 4), J = rho & 15)
    const float oscale = *a.oscale;
    const size_t img_el = (size_t)a.H * a.W * a.Cout;
    const auto rsd = __builtin_amdgcn_make_buffer_rsrc(a.dst + (size_t)nimg0 * img_el, 0, (int)(img_el * 4), 0x00020000);
    const bool edge = FLEX || tyi == 0 || tyi == a.tiles_y - 1 || txi == 0 || txi == a.tiles_x - 1;       // wave-uniform
    float st_s[NT], st_q[NT], st_k[NT];
    float nvalid = 64.f;                                    // this wave's pixels inside the tile and the image
    // bias variants of this lane's channels, all loaded before the first store (a load per element serialised the 64 stores of a
    // wave behind 64 round trips to memory: measured 35 000 cycles per workgroup, in-kernel stamps of gpurun r2 upc_ph3; a load
    // between stores still waits - in-order vmcnt - for the stores ahead of it)
    // (nine separate arrays, not one [9]: a select among elements of one array becomes an indexed load from scratch)
    float bv0[NT], bv1[NT], bv2[NT], bv3[NT], bv4[NT], bv5[NT], bv6[NT], bv7[NT], bv8[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float* pb = a.bvar + n0col + nt * 32 + r;
        bv4[nt] = pb[4 * a.Cout];
        bv0[nt] = bv1[nt] = bv2[nt] = bv3[nt] = bv5[nt] = bv6[nt] = bv7[nt] = bv8[nt] = 0.f;
        if (edge) {
            bv0[nt] = pb[0]; bv1[nt] = pb[a.Cout]; bv2[nt] = pb[2 * a.Cout]; bv3[nt] = pb[3 * a.Cout];
            bv5[nt] = pb[5 * a.Cout]; bv6[nt] = pb[6 * a.Cout]; bv7[nt] = pb[7 * a.Cout]; bv8[nt] = pb[8 * a.Cout];
        }
    }
    if constexpr (FLEX) {
        // per row: (I, J) by division, output pixel (ty0 + 2 I + pA, tx0 + 2 J + pB); rows past the parity class or outside the image are
        // dropped (out-of-range store offset) and kept out of the statistics; bias variant by the pixel's border position
        const int ncls = G.THc * G.TWc;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { st_k[nt] = stat_pivot(__builtin_fmaf(acc_t[0][nt][0], oscale, bv4[nt])); st_s[nt] = 0.f; st_q[nt] = 0.f; }
        int cnt = 0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int rho = 32 * mt + 4 * h + (i & 3) + 8 * (i >> 2);
                asm volatile("" : "+v"(rho));              // (not hoisted into 32 live registers)
                const int I = fdiv(rho, a.inv_twc), J = rho - I * G.TWc;
                const int Y = ty0 + 2 * I + pA, X = tx0 + 2 * J + pB;
                const bool ok = (rho < ncls) & (Y < a.H) & (X < a.W);
                const unsigned vpix = ok ? (unsigned)(((Y * a.W + X) * a.Cout + n0col + r) * 4) : 0x80000000u;
                cnt += ok ? 1 : 0;
                const bool top = Y == 0, bot = Y == a.H - 1, lft = X == 0, rgt = X == a.W - 1;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float b0 = top ? bv0[nt] : (bot ? bv6[nt] : bv3[nt]);
                    const float b1 = top ? bv1[nt] : (bot ? bv7[nt] : bv4[nt]);
                    const float b2 = top ? bv2[nt] : (bot ? bv8[nt] : bv5[nt]);
                    const float bv = lft ? b0 : (rgt ? b2 : b1);
                    const float v = __builtin_fmaf(acc_t[mt][nt][i], oscale, bv);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsd, vpix, nt * 128, 0);
                    const float d = ok ? v - st_k[nt] : 0.f;
                    st_s[nt] += d; st_q[nt] = __builtin_fmaf(d, d, st_q[nt]);
                }
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        cnt += __shfl_xor(cnt, 32);
        nvalid = (float)cnt;
    } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0col + nt * 32 + r;
        const float kv = stat_pivot(__builtin_fmaf(acc_t[0][nt][0], oscale, bv4[nt]));      // shifted statistics (kernels.h); any finite pivot near the data
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            // lane part: J offset 4 h -> X offset 8 h pixels
            const unsigned voff = (unsigned)((((ty0 + 4 * mt + pA) * a.W + tx0 + 8 * h + pB) * a.Cout + co) * 4);
            if (!edge) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int dI = i >> 3, dJ = (i & 3) + 8 * ((i >> 2) & 1);          // rho = (i&3) + 8 (i>>2): rho >> 4 = i >> 3, rho & 15 as here (+ 4 h)
                    const unsigned soff = (unsigned)(((2 * dI * a.W + 2 * dJ) * a.Cout) * 4);      // scalar
                    const float v = __builtin_fmaf(acc_t[mt][nt][i], oscale, bv4[nt]);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsd, voff, soff, 0);
                    const float d = v - kv;
                    s += d; q = __builtin_fmaf(d, d, q);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int dI = i >> 3, dJ = (i & 3) + 8 * ((i >> 2) & 1);
                    const unsigned soff = (unsigned)(((2 * dI * a.W + 2 * dJ) * a.Cout) * 4);
                    const int Y = ty0 + 4 * mt + pA + 2 * dI, X = tx0 + pB + 2 * (dJ + 4 * h);      // Y wave-uniform, X per lane half
                    const bool top = Y == 0, bot = Y == a.H - 1;
                    const float b0 = top ? bv0[nt] : (bot ? bv6[nt] : bv3[nt]);
                    const float b1 = top ? bv1[nt] : (bot ? bv7[nt] : bv4[nt]);
                    const float b2 = top ? bv2[nt] : (bot ? bv8[nt] : bv5[nt]);
                    const float bv = X == 0 ? b0 : (X == a.W - 1 ? b2 : b1);
                    const float v = __builtin_fmaf(acc_t[mt][nt][i], oscale, bv);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsd, voff, soff, 0);
                    const float d = v - kv;
                    s += d; q = __builtin_fmaf(d, d, q);
                }
            }
        }
        st_s[nt] = s; st_q[nt] = q; st_k[nt] = kv;
    }
    }
    TS2D_STAMP(4)
    lds_barrier();
    float* red = reinterpret_cast<float*>(smem8);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float s = st_s[nt], q = st_q[nt];
        s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
        if (h == 0) stat_wave_put(red, w * BN + nt * 32 + r, s, q, st_k[nt], nvalid);
    }
    TS2D_STAMP(5)
    lds_barrier();
    TS2D_STAMP(6)
    if (tid < BN) stat_tile_store(red, 4, BN, tid, a.part + ((size_t)(nimg0 * tpi + tin) * a.Cout + n0col + tid) * 4);
    TS2D_PROF_FLUSH(a.prof)
#undef TS2D_STAMP
}

}  // namespace ts2d
