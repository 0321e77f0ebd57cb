// conv3x3_upc_h2: the 16-bit composed decoder block entry (kernels_upc_h.h) on 16 x 32 output tiles.
//
// conv3x3_upc_h does a third of the MFMA work of the split kernel with the same per-tile overheads (eight barriers, 74 KB of
// skip-half weights staged through registers and the composed-weight stream from L2 per 256 pixels): 0.87 ms for dec1.c0 against
// floors of 0.29 (HBM) / 0.36 ms (MFMA).  Here a wave owns its parity class of a 16 x 32 tile - 8 x 16 coarse positions, FOUR
// 32-row MFMA tiles (the accumulators of a single-product kernel leave room for 128 registers of them) - so every weight byte
// and every barrier serves twice the pixels; the skip-half weights of a chunk go to LDS by global_load_lds (no registers).
// Geometry of the patches as conv3x3_upq: coarse patch 10 x 18 at pitch 24, skip patch 18 x (17 even | 17 odd) at pitch 36, an M
// tile's two 16-lane runs are coarse rows I and I + 2 (conflict-free ds_read_b128 at those pitches).  256 threads, two workgroups
// per CU (78 336 B of LDS).  Arithmetic: as conv3x3_upc_h (one fp16 product, fp32 accumulation over the whole K; statistics of the
// stored fp16 values).
#pragma once
#include "kernels_upc_h.h"
#include "kernels_upq.h"

namespace ts2d {

constexpr int kUh2Lds = 4 * kUq2Plane + 2 * 9 * 2 * 64 * 16;      // phase 2: planes [k-step 2][h 2] + weights [k-step][tap][h][column]

// UP = false: the same kernel as a PLAIN 3x3 conv of the 16-bit mode (no coarse tensor, no phase 1): xs / scs / shs / Cs = the source,
// wk = the layer's plane-order weight image (engine.hip: dev_wp, the same layout), bvar = its bias [Cout].
template <int KS, bool UP = true>
__global__ __launch_bounds__(kBlock, 2) void conv3x3_upc_h2(const UpcArgs a) {
    constexpr int BN = 64, NT = 2, MT = 4;
    constexpr int WT1 = 4 * BN * 16;                        // bytes per tap of the weight images: [hi, lo][h][column]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void* lds_ptr;
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
    const int ty0 = tyi << 4, tx0 = txi << 5;               // 16 x 32 output pixels

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), pA = w >> 1, pB = w & 1;      // this wave's output parity
    const int r = lane & 31, h = lane >> 5;
    const int octi = (lane >> 3) & 1, oct = octi * 8;
    const _Float16 slope_h = (_Float16)a.slope;
    const unsigned slope2 = (unsigned)__builtin_bit_cast(unsigned short, slope_h) * 0x10001u;

    // M tile mt, row rho of the 32: coarse position I = (mt & 1) + 4 (mt >> 1) + 2 (rho >> 4), J = rho & 15
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    // =================================================================================== phase 1: composed up half (coarse tensor)
    if constexpr (UP) {
        const int Hc = a.H >> 1, Wc = a.W >> 1;
        // staging: unit it = pixel pp = 128 it + 32 w + (lane & 7) + 8 (lane >> 4) of the 10 x 18 coarse patch, octet (lane >> 3) & 1
        unsigned vo[2]; int lw[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int pp = 128 * it + 32 * w + (lane & 7) + 8 * (lane >> 4);
            const int py = pp / 18, px = pp - py * 18;
            const int iy = (ty0 >> 1) - 1 + py, ix = (tx0 >> 1) - 1 + px;
            lw[it] = octi * kUq1Plane + (py * kUq1Pitch + px) * 16;          // planes [k-step][h]
            vo[it] = 0x80000000u;
            if (pp < 180) {
                if (iy >= 0 && iy < Hc && ix >= 0 && ix < Wc) vo[it] = (unsigned)(((iy * Wc + ix) * a.Cb + oct) * 2);
                else {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<uint4*>(smem8 + lw[it] + ks * 2 * kUq1Plane) = uint4{0u, 0u, 0u, 0u};
                }
            }
        }
        const size_t img_px = (size_t)Hc * Wc;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(reinterpret_cast<const ST*>(a.xc)) + (size_t)nimg0 * img_px * a.Cb, 0,
                                                          (int)(img_px * a.Cb * 2), 0x00020000);
        const int nch = a.Cb / (16 * KS), nks = KS * nch;
        u32x4 pv[2][KS];
        auto prefetch = [&](int ch) {
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) pv[it][ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[it] + ks * 32, ch * 32 * KS, 0);
        };
        prefetch(0);
        // tap (dI, dJ) of M tile mt reads coarse patch pixel (I + pA + dI, J + pB + dJ)
        const int abase = h * kUq1Plane + ((2 * (r >> 4) + pA) * kUq1Pitch + (r & 15) + pB) * 16;      // + ks * 2 planes + ((mt & 1) + 4 (mt >> 1)) * pitch * 16 + (dI * pitch + dJ) * 16
        const unsigned char* wgl = reinterpret_cast<const unsigned char*>(a.wc) + ((size_t)ctile * 16 + w * 4) * WT1 + h * BN * 16 + r * 16;
        const size_t wchunk = (size_t)a.n_ctiles * 16 * WT1;       // bytes per 16-channel k-step
        half8 rb[4][NT];                                            // this wave's 4 taps of the NEXT k-step to be used (ring, hi part only)
#pragma unroll
        for (int tap = 0; tap < 4; ++tap)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) rb[tap][nt] = *reinterpret_cast<const half8*>(wgl + tap * WT1 + nt * 512);
        for (int ch = 0; ch < nch; ++ch) {
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 2; ++it)
                if (vo[it] != 0x80000000u) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const float* ps = a.scc + (size_t)nimg0 * a.Cb + (ch * KS + ks) * 16 + oct; const float* pt = a.shc + (size_t)nimg0 * a.Cb + (ch * KS + ks) * 16 + oct;
                        const f32x4 nsa = *reinterpret_cast<const f32x4*>(ps), nsb = *reinterpret_cast<const f32x4*>(ps + 4);
                        const f32x4 nta = *reinterpret_cast<const f32x4*>(pt), ntb = *reinterpret_cast<const f32x4*>(pt + 4);
                        *reinterpret_cast<uint4*>(smem8 + lw[it] + ks * 2 * kUq1Plane) =
                            norm_lrelu_8(uint4{pv[it][ks][0], pv[it][ks][1], pv[it][ks][2], pv[it][ks][3]}, nsa, nsb, nta, ntb, slope2);
                    }
                }
            __syncthreads();
            if (ch + 1 < nch) prefetch(ch + 1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int knext = ch * KS + ks + 1 < nks ? ch * KS + ks + 1 : ch * KS + ks;       // (last k-step: reloaded, never used)
                const unsigned char* wnext = wgl + (size_t)knext * wchunk;
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) {
                    const int toff = ks * 2 * kUq1Plane + ((tap >> 1) * kUq1Pitch + (tap & 1)) * 16;
                    half8 ah[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) ah[mt] = *reinterpret_cast<const half8*>(smem8 + abase + ((mt & 1) + 4 * (mt >> 1)) * kUq1Pitch * 16 + toff);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], rb[tap][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) rb[tap][nt] = *reinterpret_cast<const half8*>(wnext + tap * WT1 + nt * 512);
                    __builtin_amdgcn_sched_barrier(0);      // (keeps the reload here, four taps ahead of its use: kernels_up0.h)
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }

    // =================================================================================== phase 2: skip half, 32-channel chunks
    {
        __syncthreads();                                    // phase-1 LDS reads are done: the memory is re-laid out
        constexpr int MAXU = 5;                             // 18 x 34 = 612 patch pixels, 128 per pass
        unsigned char* sB2 = smem8 + 4 * kUq2Plane;         // planes [k-step 2][h 2], then weights [k-step][tap][h][column]
        unsigned vo[MAXU];
        int lw[MAXU];
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            // unit enumeration: patch row, then its 17 even columns, then its 17 odd columns (LDS slots + 18)
            const int q = 32 * (4 * it + w) + (lane & 7) + 8 * (lane >> 4);
            const int py = q / 34, rem = q - py * 34;
            const int half = rem >= 17 ? 1 : 0, idx = rem - 17 * half, px = 2 * idx + half;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            unsigned v = 0x80000000u;
            lw[it] = octi * kUq2Plane + (py * kUq2Pitch + 18 * half + idx) * 16;
            if (q < 18 * 34) {
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = (unsigned)(((iy * a.W + ix) * a.Cs + oct) * 2);
                else { *reinterpret_cast<uint4*>(smem8 + lw[it]) = uint4{0u, 0u, 0u, 0u};
                       *reinterpret_cast<uint4*>(smem8 + lw[it] + 2 * kUq2Plane) = uint4{0u, 0u, 0u, 0u}; }
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
        // weights of a chunk: 2 k-steps x 9 taps x the hi part [h][column] (2 KB per (k-step, tap), contiguous in the image): 36 pieces of
        // 1 KB by global_load_lds, 9 per wave
        auto weights_dma = [&](int ch) {
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int p = w * 9 + j, kt = p >> 1, ks_ = kt / 9, tap_ = kt - 9 * ks_;          // (wave-uniform)
                const unsigned char* src = reinterpret_cast<const unsigned char*>(a.wk) + (((size_t)(2 * ch + ks_) * a.n_ctiles + ctile) * 9 + tap_) * WT1 + (p & 1) * 1024 + lane * 16;
                __builtin_amdgcn_global_load_lds(src, (lds_ptr)(sB2 + p * 1024), 16, 0, 0);
            }
        };
        prefetch(0);
        // M tile mt, row rho -> output pixel (2I + pA, 2J + pB); tap (ky, kx) reads patch row 2I + pA + ky, half (pB + kx) & 1, index J + ((pB + kx) >> 1)
        const int abase = h * kUq2Plane + ((4 * (r >> 4) + pA) * kUq2Pitch + (r & 15)) * 16;       // + ks * 2 planes + 2 ((mt & 1) + 4 (mt >> 1)) * pitch * 16 + tap offset
        const int bbase = 4 * kUq2Plane + h * BN * 16 + r * 16;                                   // + (ks * 9 + tap) * 2 BN 16 + nt * 512
        int tofs[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) tofs[kx] = ((((pB + kx) & 1) ? 18 : 0) + ((pB + kx) >> 1)) * 16;
        for (int ch = 0; ch < nch; ++ch) {
            __syncthreads();                                // the previous chunk's MFMA reads are done
            weights_dma(ch);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float* ps = a.scs + (size_t)nimg0 * a.Cs + ch * 32 + ks * 16 + oct; const float* pt = a.shs + (size_t)nimg0 * a.Cs + ch * 32 + ks * 16 + oct;
                const f32x4 nsa = *reinterpret_cast<const f32x4*>(ps), nsb = *reinterpret_cast<const f32x4*>(ps + 4);
                const f32x4 nta = *reinterpret_cast<const f32x4*>(pt), ntb = *reinterpret_cast<const f32x4*>(pt + 4);
#pragma unroll
                for (int it = 0; it < MAXU; ++it)
                    if (vo[it] != 0x80000000u)
                        *reinterpret_cast<uint4*>(smem8 + lw[it] + ks * 2 * kUq2Plane) =
                            norm_lrelu_8(uint4{pv[it][ks][0], pv[it][ks][1], pv[it][ks][2], pv[it][ks][3]}, nsa, nsb, nta, ntb, slope2);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the weight DMA has landed (nothing else is in flight here)
            __syncthreads();
            // (round 4: the next chunk's ten patch loads ride one per tap behind the first ten taps' MFMAs instead of being issued in
            //  one burst in front of them - each wave instruction touches 32 lines, see kernels_s2v2.h)
            const bool burst = a.dbg & 2048;                   // diagnostic A/B (TS2D_DBG=2048): the round-3 order
            if (burst && ch + 1 < nch) prefetch(ch + 1);
            const bool more = ch + 1 < nch && !burst;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    if (ks * 9 + tap < 2 * MAXU && more) {
                        const int u_ = (ks * 9 + tap) >> 1, l_ = (ks * 9 + tap) & 1;
                        pv[u_][l_] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[u_] + 32 * l_, (ch + 1) * 64, 0);
                    }
                    const int ky = tap / 3, kx = tap - 3 * ky;
                    const int toff = ks * 2 * kUq2Plane + ky * kUq2Pitch * 16 + tofs[kx];
                    half8 ah[MT], bh[NT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) ah[mt] = *reinterpret_cast<const half8*>(smem8 + abase + 2 * ((mt & 1) + 4 * (mt >> 1)) * kUq2Pitch * 16 + toff);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bh[nt] = *reinterpret_cast<const half8*>(smem8 + bbase + (ks * 9 + tap) * 2 * BN * 16 + nt * 512);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
        }
    }

    // =================================================================================== epilogue: scatter by parity (fp16 stores), statistics
    // C/D map: column = lane & 31 (channel), row rho = (i & 3) + 8 (i >> 2) + 4 h -> (I = (mt & 1) + 4 (mt >> 1) + 2 (rho >> 4), J = rho & 15)
    const float oscale = *a.oscale;
    const size_t img_el = (size_t)a.H * a.W * a.Cout;
    const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<ST*>(a.dst) + (size_t)nimg0 * img_el, 0, (int)(img_el * 2), 0x00020000);
    const bool edge = UP && (tyi == 0 || tyi == a.tiles_y - 1 || txi == 0 || txi == a.tiles_x - 1);       // wave-uniform
    float st_s[NT], st_q[NT], st_k[NT];
    float bv0[NT], bv1[NT], bv2[NT], bv3[NT], bv4[NT], bv5[NT], bv6[NT], bv7[NT], bv8[NT];       // (nine arrays: see kernels_upc.h)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float* pb = a.bvar + n0col + nt * 32 + r;
        bv4[nt] = UP ? pb[4 * a.Cout] : pb[0];
        bv0[nt] = bv1[nt] = bv2[nt] = bv3[nt] = bv5[nt] = bv6[nt] = bv7[nt] = bv8[nt] = 0.f;
        if (edge) {
            bv0[nt] = pb[0]; bv1[nt] = pb[a.Cout]; bv2[nt] = pb[2 * a.Cout]; bv3[nt] = pb[3 * a.Cout];
            bv5[nt] = pb[5 * a.Cout]; bv6[nt] = pb[6 * a.Cout]; bv7[nt] = pb[7 * a.Cout]; bv8[nt] = pb[8 * a.Cout];
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0col + nt * 32 + r;
        const float kv = stat_pivot(round_act<ST>(__builtin_fmaf(acc[0][nt][0], oscale, bv4[nt])));      // shifted statistics (kernels.h)
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int Y0 = ty0 + 2 * ((mt & 1) + 4 * (mt >> 1)) + pA;                          // + 4 (i >> 3)
            const unsigned voff = (unsigned)(((Y0 * a.W + tx0 + 8 * h + pB) * a.Cout + co) * 2);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int dI = i >> 3, dJ = (i & 3) + 8 * ((i >> 2) & 1);
                const unsigned soff = (unsigned)(((4 * dI * a.W + 2 * dJ) * a.Cout) * 2);      // scalar
                float bv = bv4[nt];
                if (edge) {
                    const int Y = Y0 + 4 * dI, X = tx0 + pB + 2 * (dJ + 4 * h);
                    const bool top = Y == 0, bot = Y == a.H - 1;
                    const float b0 = top ? bv0[nt] : (bot ? bv6[nt] : bv3[nt]);
                    const float b1 = top ? bv1[nt] : (bot ? bv7[nt] : bv4[nt]);
                    const float b2 = top ? bv2[nt] : (bot ? bv8[nt] : bv5[nt]);
                    bv = X == 0 ? b0 : (X == a.W - 1 ? b2 : b1);
                }
                float v = __builtin_fmaf(acc[mt][nt][i], oscale, bv);
                buffer_store_act<ST>(v, rsd, voff, soff);
                const float d = round_act<ST>(v) - kv;                               // statistics of what is stored
                s += d; q = __builtin_fmaf(d, d, q);
            }
        }
        st_s[nt] = s; st_q[nt] = q; st_k[nt] = kv;
    }
    lds_barrier();
    float* red = reinterpret_cast<float*>(smem8);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float s = st_s[nt], q = st_q[nt];
        s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
        if (h == 0) stat_wave_put(red, w * BN + nt * 32 + r, s, q, st_k[nt], 128.f);
    }
    lds_barrier();
    if (tid < BN) stat_tile_store(red, 4, BN, tid, a.part + ((size_t)(nimg0 * tpi + tin) * a.Cout + n0col + tid) * 4);
}

}  // namespace ts2d
