// conv3x3_upc_h: the composed decoder block entry of kernels_upc.h for the 16-bit mode (ts2d_engine_set_precision F16: activations
// stored as fp16, weights rounded to fp16 - here the fp64-COMPOSED weights rounded once -, ONE fp16 MFMA product, fp32 accumulation
// and statistics).  Same tiling as conv3x3_upc (256 threads, 8 x 32 output pixels, wave = output parity), with the chunk sizes the
// single product asks for: phase 1 walks the coarse channels KS k-steps (16 channels each) per barrier pair (KS = 4: 64 MFMAs per
// wave for BN = 64), phase 2 the skip channels in chunks of 32.  The weight images are the split mode's (engine.hip:pack_weights:
// [.. tap][hi, lo][h][column][8 halves]); only the hi part is read.  The two-kernel path of this mode spent 1.17 ms of a
// 12.9 ms step in the five transposed-conv launches alone (gpurun r2 f16ops).
#pragma once
#include "kernels_upc.h"
#include "kernels_h32.h"

namespace ts2d {

// FLEX: extent-following tiles, exactly as in conv3x3_upc (kernels_upc.h: UpcGeo).
template <int BN, int KS, bool FLEX = false>
__global__ __launch_bounds__(kBlock, BN == 32 ? 3 : 2) void conv3x3_upc_h(const UpcArgs a) {
    constexpr int NT = BN / 32;
    const UpcGeo<FLEX> G(a);
    constexpr int WT1 = 4 * BN * 16;                        // bytes per tap of the weight images: [hi, lo][h][column]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef _Float16 ST;

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
    int fI[2] = {0, 0}, fJ[2] = {0, 0};                     // FLEX: this lane's coarse position per M tile (kernels_upc.h)
    if constexpr (FLEX) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int rho = 32 * mt + r;
            if (rho < G.THc * G.TWc) { fI[mt] = fdiv(rho, a.inv_twc); fJ[mt] = rho - fI[mt] * G.TWc; }
        }
    }
    const _Float16 slope_h = (_Float16)a.slope;
    const unsigned slope2 = (unsigned)__builtin_bit_cast(unsigned short, slope_h) * 0x10001u;

    f32x16 acc_t[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_t[mt][nt][i] = 0.f;

    // =================================================================================== phase 1: composed up half (coarse tensor)
    {
        const int Hc = a.H >> 1, Wc = a.W >> 1;
        const int pp = 32 * w + (lane & 7) + 8 * (lane >> 4);              // pixel of the 6 x 18 coarse patch, octet (lane >> 3) & 1
        const int py = pp / G.C1W, px = pp - py * G.C1W;
        const int iy = (ty0 >> 1) - 1 + py, ix = (tx0 >> 1) - 1 + px;
        const int lw = octi * kUcPlane + (py * G.P1 + px) * 16;             // planes [k-step][h]
        unsigned vo = 0x80000000u;
        if (pp < G.C1N) {
            if (iy >= 0 && iy < Hc && ix >= 0 && ix < Wc) vo = (unsigned)(((iy * Wc + ix) * a.Cb + oct) * 2);
            else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<uint4*>(smem8 + lw + ks * 2 * kUcPlane) = uint4{0u, 0u, 0u, 0u};
            }
        }
        const size_t img_px = (size_t)Hc * Wc;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(reinterpret_cast<const ST*>(a.xc)) + (size_t)nimg0 * img_px * a.Cb, 0,
                                                          (int)(img_px * a.Cb * 2), 0x00020000);
        const int nch = a.Cb / (16 * KS), nks = KS * nch;
        u32x4 pv[KS];
        auto prefetch = [&](int ch) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) pv[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + ks * 32, ch * 32 * KS, 0);
        };
        prefetch(0);
        const int abase = h * kUcPlane + (((r >> 4) + pA) * kUcPitch + (r & 15) + pB) * 16;      // + ks * 2 planes + mt * 2 * pitch * 16 + (dI * pitch + dJ) * 16
        int ab1[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) ab1[mt] = FLEX ? h * kUcPlane + ((fI[mt] + pA) * G.P1 + fJ[mt] + pB) * 16 : abase + mt * 2 * kUcPitch * 16;
        const unsigned char* wgl = reinterpret_cast<const unsigned char*>(a.wc) + ((size_t)ctile * 16 + w * 4) * WT1 + h * BN * 16 + r * 16;
        const size_t wchunk = (size_t)a.n_ctiles * 16 * WT1;       // bytes per 16-channel k-step
        half8 rb[4][NT];                                            // this wave's 4 taps of the NEXT k-step to be used (ring, hi part only)
#pragma unroll
        for (int tap = 0; tap < 4; ++tap)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) rb[tap][nt] = *reinterpret_cast<const half8*>(wgl + tap * WT1 + nt * 512);
        for (int ch = 0; ch < nch; ++ch) {
            __syncthreads();
            if (vo != 0x80000000u) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const float* ps = a.scc + (size_t)nimg0 * a.Cb + (ch * KS + ks) * 16 + oct; const float* pt = a.shc + (size_t)nimg0 * a.Cb + (ch * KS + ks) * 16 + oct;
                    const f32x4 nsa = *reinterpret_cast<const f32x4*>(ps), nsb = *reinterpret_cast<const f32x4*>(ps + 4);
                    const f32x4 nta = *reinterpret_cast<const f32x4*>(pt), ntb = *reinterpret_cast<const f32x4*>(pt + 4);
                    *reinterpret_cast<uint4*>(smem8 + lw + ks * 2 * kUcPlane) =
                        norm_lrelu_8(uint4{pv[ks][0], pv[ks][1], pv[ks][2], pv[ks][3]}, nsa, nsb, nta, ntb, slope2);
                }
            }
            __syncthreads();
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
            for (int ks = 0; ks < KS; ++ks) {
                const int knext = ch * KS + ks + 1 < nks ? ch * KS + ks + 1 : ch * KS + ks;       // (last k-step: reloaded, never used)
                const unsigned char* wnext = wgl + (size_t)knext * wchunk;
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) {
                    const int toff = ks * 2 * kUcPlane + ((tap >> 1) * G.P1 + (tap & 1)) * 16;
                    half8 ah[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) ah[mt] = *reinterpret_cast<const half8*>(smem8 + ab1[mt] + toff);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], rb[tap][nt], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) rb[tap][nt] = *reinterpret_cast<const half8*>(wnext + tap * WT1 + nt * 512);
                    __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks the reload next to its use four taps later: kernels_up0.h)
                }
            }
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc_t[mt][nt] += acc_c[mt][nt];
        }
    }

    // =================================================================================== phase 2: skip half, 32-channel chunks
    {
        __syncthreads();                                    // phase-1 LDS reads are done: the memory is re-laid out
        constexpr int MAXU = 3;
        unsigned char* sB2 = smem8 + 4 * kUsPlane;          // planes [k-step 2][h 2], then weights [k-step][tap][h][column]
        unsigned vo[MAXU];
        int lw[MAXU];
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            const int q = 32 * (4 * it + w) + (lane & 7) + 8 * (lane >> 4);
            const int py = q / G.S2W, rem = q - py * G.S2W;
            const int half = rem >= G.NE ? 1 : 0, idx = rem - G.NE * half, px = 2 * idx + half;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            unsigned v = 0x80000000u;
            lw[it] = octi * kUsPlane + (py * G.P2 + G.HO * half + idx) * 16;
            if (q < G.S2N) {
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = (unsigned)(((iy * a.W + ix) * a.Cs + oct) * 2);
                else { *reinterpret_cast<uint4*>(smem8 + lw[it]) = uint4{0u, 0u, 0u, 0u};
                       *reinterpret_cast<uint4*>(smem8 + lw[it] + 2 * kUsPlane) = uint4{0u, 0u, 0u, 0u}; }
            }
            vo[it] = v;
        }
        const size_t img_px = (size_t)a.H * a.W;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(reinterpret_cast<const ST*>(a.xs)) + (size_t)nimg0 * img_px * a.Cs, 0,
                                                          (int)(img_px * a.Cs * 2), 0x00020000);
        const int nch = a.Cs / 32;
        u32x4 pv[MAXU][2];                                  // [unit][k-step]
        auto prefetch = [&](int ch) {
#pragma unroll
            for (int it = 0; it < MAXU; ++it) {
                pv[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[it], ch * 64, 0);
                pv[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[it] + 32, ch * 64, 0);
            }
        };
        prefetch(0);
        const int abase = h * kUsPlane + ((2 * (r >> 4) + pA) * kUsPitch + (r & 15)) * 16;       // + ks * 2 planes + mt * 4 * pitch * 16 + tap offset
        int ab2[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) ab2[mt] = FLEX ? h * kUsPlane + ((2 * fI[mt] + pA) * G.P2 + fJ[mt]) * 16 : abase + mt * 4 * kUsPitch * 16;
        const int bbase = 4 * kUsPlane + h * BN * 16 + r * 16;                                   // + (ks * 9 + tap) * 2 BN 16 + nt * 512
        int tofs[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) tofs[kx] = ((((pB + kx) & 1) ? G.HO : 0) + ((pB + kx) >> 1)) * 16;
        for (int ch = 0; ch < nch; ++ch) {
            __syncthreads();
            // weights of this chunk: 2 k-steps x 9 taps x the hi part [h][column] (2 BN slots per tap, contiguous in the image)
            constexpr int WU = 2 * 9 * 2 * BN, WIT = (WU + kBlock - 1) / kBlock;
            const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wk);
            uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8;
#define TS2D_WL(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU)) { \
                const int ks_ = idx / (18 * BN), rem_ = idx - ks_ * (18 * BN), tap_ = rem_ / (2 * BN), sl_ = rem_ - tap_ * (2 * BN); \
                R = *reinterpret_cast<const uint4*>(wsrc + (((size_t)(2 * ch + ks_) * a.n_ctiles + ctile) * 9 + tap_) * WT1 + sl_ * 16); } }
            TS2D_WL(0, w0) TS2D_WL(1, w1) TS2D_WL(2, w2) TS2D_WL(3, w3) TS2D_WL(4, w4) TS2D_WL(5, w5) TS2D_WL(6, w6) TS2D_WL(7, w7) TS2D_WL(8, w8)
#undef TS2D_WL
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float* ps = a.scs + (size_t)nimg0 * a.Cs + ch * 32 + ks * 16 + oct; const float* pt = a.shs + (size_t)nimg0 * a.Cs + ch * 32 + ks * 16 + oct;
                const f32x4 nsa = *reinterpret_cast<const f32x4*>(ps), nsb = *reinterpret_cast<const f32x4*>(ps + 4);
                const f32x4 nta = *reinterpret_cast<const f32x4*>(pt), ntb = *reinterpret_cast<const f32x4*>(pt + 4);
#pragma unroll
                for (int it = 0; it < MAXU; ++it)
                    if (vo[it] != 0x80000000u)
                        *reinterpret_cast<uint4*>(smem8 + lw[it] + ks * 2 * kUsPlane) =
                            norm_lrelu_8(uint4{pv[it][ks][0], pv[it][ks][1], pv[it][ks][2], pv[it][ks][3]}, nsa, nsb, nta, ntb, slope2);
            }
#define TS2D_WS(K, R) { const int idx = tid + K * kBlock; if (K < WIT && (WU % kBlock == 0 || idx < WU)) *reinterpret_cast<uint4*>(sB2 + idx * 16) = R; }
            TS2D_WS(0, w0) TS2D_WS(1, w1) TS2D_WS(2, w2) TS2D_WS(3, w3) TS2D_WS(4, w4) TS2D_WS(5, w5) TS2D_WS(6, w6) TS2D_WS(7, w7) TS2D_WS(8, w8)
#undef TS2D_WS
            __syncthreads();
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
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ky = tap / 3, kx = tap - 3 * ky;
                    const int toff = ks * 2 * kUsPlane + ky * G.P2 * 16 + tofs[kx];
                    half8 ah[2], bh[NT];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) ah[mt] = *reinterpret_cast<const half8*>(smem8 + ab2[mt] + toff);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bh[nt] = *reinterpret_cast<const half8*>(smem8 + bbase + (ks * 9 + tap) * 2 * BN * 16 + nt * 512);
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

    // =================================================================================== epilogue: scatter by parity (fp16 stores), statistics
    const float oscale = *a.oscale;
    const size_t img_el = (size_t)a.H * a.W * a.Cout;
    const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<ST*>(a.dst) + (size_t)nimg0 * img_el, 0, (int)(img_el * 2), 0x00020000);
    const bool edge = FLEX || tyi == 0 || tyi == a.tiles_y - 1 || txi == 0 || txi == a.tiles_x - 1;       // wave-uniform
    float st_s[NT], st_q[NT], st_k[NT];
    float nvalid = 64.f;
    float bv0[NT], bv1[NT], bv2[NT], bv3[NT], bv4[NT], bv5[NT], bv6[NT], bv7[NT], bv8[NT];       // (nine arrays: see kernels_upc.h)
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
    if constexpr (FLEX) {          // (the epilogue of conv3x3_upc<.., FLEX>, fp16 stores)
        const int ncls = G.THc * G.TWc;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { st_k[nt] = stat_pivot(round_act<ST>(__builtin_fmaf(acc_t[0][nt][0], oscale, bv4[nt]))); st_s[nt] = 0.f; st_q[nt] = 0.f; }
        int cnt = 0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int rho = 32 * mt + 4 * h + (i & 3) + 8 * (i >> 2);
                asm volatile("" : "+v"(rho));
                const int I = fdiv(rho, a.inv_twc), J = rho - I * G.TWc;
                const int Y = ty0 + 2 * I + pA, X = tx0 + 2 * J + pB;
                const bool ok = (rho < ncls) & (Y < a.H) & (X < a.W);
                const unsigned vpix = ok ? (unsigned)(((Y * a.W + X) * a.Cout + n0col + r) * 2) : 0x80000000u;
                cnt += ok ? 1 : 0;
                const bool top = Y == 0, bot = Y == a.H - 1, lft = X == 0, rgt = X == a.W - 1;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float b0 = top ? bv0[nt] : (bot ? bv6[nt] : bv3[nt]);
                    const float b1 = top ? bv1[nt] : (bot ? bv7[nt] : bv4[nt]);
                    const float b2 = top ? bv2[nt] : (bot ? bv8[nt] : bv5[nt]);
                    const float bv = lft ? b0 : (rgt ? b2 : b1);
                    const float v = __builtin_fmaf(acc_t[mt][nt][i], oscale, bv);
                    buffer_store_act<ST>(v, rsd, vpix, nt * 64);
                    const float d = ok ? round_act<ST>(v) - st_k[nt] : 0.f;
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
        const float kv = stat_pivot(round_act<ST>(__builtin_fmaf(acc_t[0][nt][0], oscale, bv4[nt])));      // shifted statistics (kernels.h)
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const unsigned voff = (unsigned)((((ty0 + 4 * mt + pA) * a.W + tx0 + 8 * h + pB) * a.Cout + co) * 2);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int dI = i >> 3, dJ = (i & 3) + 8 * ((i >> 2) & 1);
                const unsigned soff = (unsigned)(((2 * dI * a.W + 2 * dJ) * a.Cout) * 2);      // scalar
                float bv = bv4[nt];
                if (edge) {
                    const int Y = ty0 + 4 * mt + pA + 2 * dI, X = tx0 + pB + 2 * (dJ + 4 * h);
                    const bool top = Y == 0, bot = Y == a.H - 1;
                    const float b0 = top ? bv0[nt] : (bot ? bv6[nt] : bv3[nt]);
                    const float b1 = top ? bv1[nt] : (bot ? bv7[nt] : bv4[nt]);
                    const float b2 = top ? bv2[nt] : (bot ? bv8[nt] : bv5[nt]);
                    bv = X == 0 ? b0 : (X == a.W - 1 ? b2 : b1);
                }
                float v = __builtin_fmaf(acc_t[mt][nt][i], oscale, bv);
                buffer_store_act<ST>(v, rsd, voff, soff);
                const float d = round_act<ST>(v) - kv;                               // statistics of what is stored
                s += d; q = __builtin_fmaf(d, d, q);
            }
        }
        st_s[nt] = s; st_q[nt] = q; st_k[nt] = kv;
    }
    }
    lds_barrier();
    float* red = reinterpret_cast<float*>(smem8);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float s = st_s[nt], q = st_q[nt];
        s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
        if (h == 0) stat_wave_put(red, w * BN + nt * 32 + r, s, q, st_k[nt], nvalid);
    }
    lds_barrier();
    if (tid < BN) stat_tile_store(red, 4, BN, tid, a.part + ((size_t)(nimg0 * tpi + tin) * a.Cout + n0col + tid) * 4);
}

}  // namespace ts2d
