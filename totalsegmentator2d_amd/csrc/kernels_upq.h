// conv3x3_upq: the composed decoder block entry of kernels_upc.h (ConvTranspose2d folded into the 3x3 conv: same weights, same
// arithmetic per output) as ONE 512-thread workgroup per CU with double-buffered staging - the structure of conv3x3_f16x3_q.
//
// Phase stamps of conv3x3_upc<64> (profiles/r02_phase_stamps.txt, dec2.c0): 180 k cycles per workgroup of which 47 k are staging
// between barriers and 125 k the MFMA phases (104 k = the matrix pipe shared by two workgroups) - 58 % of the pipe.  Here
//   * tile 16 x 32 output pixels x 64 columns, 8 waves: wave w owns output parity (w & 3) and half (w >> 2) of the 8 x 16 coarse
//     positions; an M tile's two 16-lane runs are coarse rows I and I + 2, so that their LDS slots differ by a multiple of 16
//     at a row pitch of 24 slots (coarse patch) / 36 slots (skip patch: 17 even | 17 odd columns) - conflict-free ds_read_b128
//     without padding the pitch to 32 / 40 (which would not fit twice);
//   * phase 1 (coarse tensor, 32-channel chunks): patch double-buffered, chunk c+1 converted and chunk c+2 requested between the
//     MFMAs of chunk c, ONE LDS-only barrier per chunk (no DMA in this phase: the global loads stay in flight across it); the
//     B operand comes straight from L2 through a register ring, as in conv3x3_upc;
//   * phase 2 (skip tensor, 16-channel chunks, 9 taps): exactly the pipeline of conv3x3_f16x3_q - patch and weights
//     double-buffered, weights by global_load_lds, one raw barrier per chunk;
//   * scale / shift of the source channels live in an LDS table (filled once per phase) instead of prefetch registers.
// Measured (gpurun r2 upq1/upq2): dec4.c0 1.20 -> 1.12 ms, dec3.c0 1.62 -> 1.50, dec2.c0 1.69 -> 1.59, dec1.c0 1.79 -> 1.80 (stays on
// conv3x3_upc).  Stamps: wave 0 spends a third of its time at the per-chunk barrier, as in conv3x3_f16x3_q - it gets through its
// own chunk (96 / 108 MFMAs + ~260 other instructions) in the time the matrix pipe needs for BOTH waves of the SIMD, and its
// partner then needs as long again for the rest: the other instructions of the two resident waves do not ride in the shadow of
// the MFMAs (the additive model of DESIGN.md section 4), so what is left is their count, not their placement.
// LDS: phase 2 [patch 0 | patch 1 | weights 0 | weights 1] = 2 x 41472 + 2 x 36864 = 156672 B, phase 1 uses the first 61440 B
// (2 x 8 planes x 3840), table 4096 B behind: 160768 B.
#pragma once
#include "kernels_upc.h"

namespace ts2d {

constexpr int kUqThreads = 512;
constexpr int kUq1Pitch = 24, kUq1Plane = 10 * kUq1Pitch * 16, kUq1Buf = 8 * kUq1Plane;          // coarse patch 10 x 18 (pitch 24), 8 planes (ks, part, h)
constexpr int kUq2Pitch = 36, kUq2Plane = 18 * kUq2Pitch * 16, kUq2Buf = 4 * kUq2Plane;          // skip patch 18 x (17 | 17) at pitch 36, 4 planes (part, h)
constexpr int kUqWts = 9 * 4 * 64 * 16, kUqTable = 2 * kUq2Buf + 2 * kUqWts, kUqLds = kUqTable + 4096;

__global__ __launch_bounds__(kUqThreads, 1) void conv3x3_upq(const UpcArgs a) {
    constexpr int BN = 64, NT = 2, WT1 = 4 * BN * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void* lds_ptr;

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
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    if (!(a.dbg & 512)) { if (w >= 4) __builtin_amdgcn_s_setprio(1); }      // static issue priority for waves 4-7 (kernels_f16x3_qp.h; here within noise)
    const int pA = (w >> 1) & 1, pB = w & 1, hw = w >> 2;   // this wave's output parity and half of the coarse rows
    const int r = lane & 31, h = lane >> 5;
    const int octi = (lane >> 3) & 1, oct = octi * 8;
    float* const table = reinterpret_cast<float*>(smem8 + kUqTable);      // [scale C | shift C] of the phase's source

    TS2D_PROF_DECL(a.prof);
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
        for (int i = tid; i < a.Cb; i += kUqThreads) {
            table[i] = a.scc[(size_t)nimg0 * a.Cb + i]; table[a.Cb + i] = a.shc[(size_t)nimg0 * a.Cb + i];
        }
        // staging: pixel pp = 32 w + (lane & 7) + 8 (lane >> 4) of the 10 x 18 coarse patch, octet (lane >> 3) & 1, two units (k-steps)
        const int pp = 32 * w + (lane & 7) + 8 * (lane >> 4);
        const int py = pp / 18, px = pp - py * 18;
        const int iy = (ty0 >> 1) - 1 + py, ix = (tx0 >> 1) - 1 + px;
        const int lw = octi * kUq1Plane + (py * kUq1Pitch + px) * 16;
        const bool unit = pp < 180;
        unsigned vo = 0x80000000u;
        bool real = false;
        if (unit) {
            real = iy >= 0 && iy < Hc && ix >= 0 && ix < Wc;
            if (real) vo = (unsigned)(((iy * Wc + ix) * a.Cb + oct) * 4);
        }
        const size_t img_px = (size_t)Hc * Wc;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xc) + (size_t)nimg0 * img_px * a.Cb, 0, (int)(img_px * a.Cb * 4), 0x00020000);
        const int nch = a.Cb / 32, nks = 2 * nch;
        u32x4 pv[2][2];
        auto prefetch = [&](int ch) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                pv[ks][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + ks * 64, ch * 128, 0);
                pv[ks][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + ks * 64 + 16, ch * 128, 0);
            }
        };
        auto convert = [&](int ks, int ch, unsigned char* pb) {      // branch-free arithmetic; a padding pixel stores zeros
            const float* ps = table + ch * 32 + ks * 16 + oct;
            const f32x4 nsa = *reinterpret_cast<const f32x4*>(ps), nsb = *reinterpret_cast<const f32x4*>(ps + 4);
            const f32x4 nta = *reinterpret_cast<const f32x4*>(ps + a.Cb), ntb = *reinterpret_cast<const f32x4*>(ps + a.Cb + 4);
            f32x4 va = __builtin_bit_cast(f32x4, pv[ks][0]), vb = __builtin_bit_cast(f32x4, pv[ks][1]);
            va = va * nsa + nta; vb = vb * nsb + ntb;
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] = fmaxf(va[e], va[e] * a.slope); vb[e] = fmaxf(vb[e], vb[e] * a.slope); }
            uint4 hi, lo;
            split_hi_lo_8(va, vb, hi, lo);
            hi.x = real ? hi.x : 0u; hi.y = real ? hi.y : 0u; hi.z = real ? hi.z : 0u; hi.w = real ? hi.w : 0u;
            lo.x = real ? lo.x : 0u; lo.y = real ? lo.y : 0u; lo.z = real ? lo.z : 0u; lo.w = real ? lo.w : 0u;
            if (unit) {
                *reinterpret_cast<uint4*>(pb + lw + ks * 4 * kUq1Plane) = hi;
                *reinterpret_cast<uint4*>(pb + lw + ks * 4 * kUq1Plane + 2 * kUq1Plane) = lo;
            }
        };
        // M-tile row r = coarse position (I = 4 hw + mt + 2 (r >> 4), J = r & 15); tap (dI, dJ) reads patch pixel (I + pA + dI, J + pB + dJ)
        const int abase = h * kUq1Plane + ((4 * hw + 2 * (r >> 4) + pA) * kUq1Pitch + (r & 15) + pB) * 16;
        const unsigned char* wgl = reinterpret_cast<const unsigned char*>(a.wc) + ((size_t)ctile * 16 + (w & 3) * 4) * WT1 + h * BN * 16 + r * 16;
        const size_t wchunk = (size_t)a.n_ctiles * 16 * WT1;       // bytes per 16-channel k-step
        half8 rb[4][NT][2];
#pragma unroll
        for (int tap = 0; tap < 4; ++tap)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                rb[tap][nt][0] = *reinterpret_cast<const half8*>(wgl + tap * WT1 + nt * 512);
                rb[tap][nt][1] = *reinterpret_cast<const half8*>(wgl + tap * WT1 + nt * 512 + 2 * BN * 16);
            }
        prefetch(0);
        lds_barrier();                                       // the table is complete
        convert(0, 0, smem8); convert(1, 0, smem8);
        prefetch(nch > 1 ? 1 : 0);
        lds_barrier();
        TS2D_STAMP_AT(a.prof, 0)
        for (int ch = 0; ch < nch; ++ch) {
            const int b = ch & 1;
            const unsigned char* pa = smem8 + b * kUq1Buf + abase;
            unsigned char* pb_next = smem8 + (b ^ 1) * kUq1Buf;
            const int ch1 = ch + 1 < nch ? ch + 1 : nch - 1, ch2 = ch + 2 < nch ? ch + 2 : nch - 1;
            f32x16 acc_c[2][NT];                         // fresh per chunk: its first MFMA takes the constant 0 as C (no 64 v_mov per chunk)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int knext = 2 * ch + ks + 1 < nks ? 2 * ch + ks + 1 : 2 * ch + ks;      // (last k-step: reloaded, never used)
                const unsigned char* wnext = wgl + (size_t)knext * wchunk;
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) {
                    const int toff = ks * 4 * kUq1Plane + ((tap >> 1) * kUq1Pitch + (tap & 1)) * 16;
                    half8 ah[2], al[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        ah[mt] = *reinterpret_cast<const half8*>(pa + mt * kUq1Pitch * 16 + toff);
                        al[mt] = *reinterpret_cast<const half8*>(pa + mt * kUq1Pitch * 16 + toff + 2 * kUq1Plane);
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], rb[tap][nt][0], (ks == 0 && tap == 0) ? kZero16 : acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], rb[tap][nt][1], acc_c[mt][nt], 0, 0, 0);
                    // the next chunk's staging rides between the MFMAs of the first tap steps
                    if (ks == 0 && tap == 0) convert(0, ch1, pb_next);
                    if (ks == 0 && tap == 1) convert(1, ch1, pb_next);
                    if (ks == 0 && tap == 2) prefetch(ch2);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], rb[tap][nt][0], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        rb[tap][nt][0] = *reinterpret_cast<const half8*>(wnext + tap * WT1 + nt * 512);
                        rb[tap][nt][1] = *reinterpret_cast<const half8*>(wnext + tap * WT1 + nt * 512 + 2 * BN * 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc_t[mt][nt] += acc_c[mt][nt];
            TS2D_STAMP_AT(a.prof, 1)
            lds_barrier();                                   // (no DMA in this phase: patch prefetch and weight ring stay in flight)
            TS2D_STAMP_AT(a.prof, 0)
        }
    }

    // =================================================================================== phase 2: skip half (ordinary 3x3 taps)
    {
        constexpr int MAXU = 3;
        for (int i = tid; i < a.Cs; i += kUqThreads) {
            table[i] = a.scs[(size_t)nimg0 * a.Cs + i]; table[a.Cs + i] = a.shs[(size_t)nimg0 * a.Cs + i];
        }
        unsigned vo[MAXU];
        int lw[MAXU];
        bool real[MAXU], unit[MAXU];
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            // unit enumeration: patch row, then its 17 even columns, then its 17 odd columns (LDS slots +18)
            const int q = 32 * (8 * it + w) + (lane & 7) + 8 * (lane >> 4);
            const int py = q / 34, rem = q - py * 34;
            const int half = rem >= 17 ? 1 : 0, idx = rem - 17 * half, px = 2 * idx + half;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            unit[it] = q < 18 * 34;
            real[it] = unit[it] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            lw[it] = octi * kUq2Plane + (py * kUq2Pitch + 18 * half + idx) * 16;
            vo[it] = real[it] ? (unsigned)(((iy * a.W + ix) * a.Cs + oct) * 4) : 0x80000000u;
        }
        const size_t img_px = (size_t)a.H * a.W;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xs) + (size_t)nimg0 * img_px * a.Cs, 0, (int)(img_px * a.Cs * 4), 0x00020000);
        const int nch = a.Cs / 16;
        u32x4 pv[MAXU][2];
        auto prefetch_unit = [&](int it, int ch) {
            pv[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[it], ch * 64, 0);
            pv[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[it] + 16, ch * 64, 0);
        };
        auto convert = [&](int it, int ch, unsigned char* pb) {
            const float* ps = table + ch * 16 + oct;
            const f32x4 nsa = *reinterpret_cast<const f32x4*>(ps), nsb = *reinterpret_cast<const f32x4*>(ps + 4);
            const f32x4 nta = *reinterpret_cast<const f32x4*>(ps + a.Cs), ntb = *reinterpret_cast<const f32x4*>(ps + a.Cs + 4);
            f32x4 va = __builtin_bit_cast(f32x4, pv[it][0]), vb = __builtin_bit_cast(f32x4, pv[it][1]);
            va = va * nsa + nta; vb = vb * nsb + ntb;
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] = fmaxf(va[e], va[e] * a.slope); vb[e] = fmaxf(vb[e], vb[e] * a.slope); }
            uint4 hi, lo;
            split_hi_lo_8(va, vb, hi, lo);
            const bool rl = real[it];
            hi.x = rl ? hi.x : 0u; hi.y = rl ? hi.y : 0u; hi.z = rl ? hi.z : 0u; hi.w = rl ? hi.w : 0u;
            lo.x = rl ? lo.x : 0u; lo.y = rl ? lo.y : 0u; lo.z = rl ? lo.z : 0u; lo.w = rl ? lo.w : 0u;
            if (it < 2 || unit[it]) {
                *reinterpret_cast<uint4*>(pb + lw[it]) = hi;
                *reinterpret_cast<uint4*>(pb + lw[it] + 2 * kUq2Plane) = lo;
            }
        };
        unsigned char* const wbuf0 = smem8 + 2 * kUq2Buf;
        auto weights_dma = [&](int ch, unsigned char* wb) {          // 36 pieces of 1 KiB; every wave issues exactly 5 (pieces 32..35 twice)
            const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wk) + ((size_t)ch * a.n_ctiles + ctile) * kUqWts + lane * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds(wsrc + (w + 8 * j) * 1024, (lds_ptr)(wb + (w + 8 * j) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(wsrc + ((w & 3) + 32) * 1024, (lds_ptr)(wb + ((w & 3) + 32) * 1024), 16, 0, 0);
        };
        // fragments: row r = (I = 4 hw + mt + 2 (r >> 4), J = r & 15) -> output pixel (2I + pA, 2J + pB); tap (ky, kx) reads patch row
        // 2I + pA + ky, column parity (pB + kx) & 1, index J + ((pB + kx) >> 1)
        const int abase = h * kUq2Plane + ((8 * hw + 4 * (r >> 4) + pA) * kUq2Pitch + (r & 15)) * 16;      // + mt * 2 * pitch * 16 + tap + part * 2 * Plane
        const int bbase = h * BN * 16 + r * 16;
        int tofs[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) tofs[kx] = ((((pB + kx) & 1) ? 18 : 0) + ((pB + kx) >> 1)) * 16;

        // prologue of the phase (exposed once per workgroup): chunk 0 staged, chunk 1 requested.  The phase-1 buffers are dead: every
        // wave passed the last phase-1 barrier
#pragma unroll
        for (int it = 0; it < MAXU; ++it) prefetch_unit(it, 0);
        weights_dma(0, wbuf0);
        lds_barrier();                                       // table complete (LDS writes only; the DMA is not read before the next barrier)
#pragma unroll
        for (int it = 0; it < MAXU; ++it) convert(it, 0, smem8);
#pragma unroll
        for (int it = 0; it < MAXU; ++it) prefetch_unit(it, nch > 1 ? 1 : 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        TS2D_STAMP_AT(a.prof, 2)

        for (int ch = 0; ch < nch; ++ch) {
            const int b = ch & 1;
            const unsigned char* pa = smem8 + b * kUq2Buf + abase;
            const unsigned char* pw = wbuf0 + b * kUqWts + bbase;
            unsigned char* pb_next = smem8 + (b ^ 1) * kUq2Buf;
            unsigned char* wb_next = wbuf0 + (b ^ 1) * kUqWts;
            const int ch1 = ch + 1 < nch ? ch + 1 : nch - 1, ch2 = ch + 2 < nch ? ch + 2 : nch - 1;
            f32x16 acc_c[2][NT];                         // (first MFMA of the chunk: C = 0)
            half8 fa[2][2][2], fb[2][NT][2];            // [buffer][tile][hi, lo]
#define TS2D_LOAD_FRAGS(BUF, TAP) { \
                const int toff_ = ((TAP) / 3) * kUq2Pitch * 16 + tofs[(TAP) % 3]; \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) { \
                    fa[BUF][mt][0] = *reinterpret_cast<const half8*>(pa + mt * 2 * kUq2Pitch * 16 + toff_); \
                    fa[BUF][mt][1] = *reinterpret_cast<const half8*>(pa + mt * 2 * kUq2Pitch * 16 + toff_ + 2 * kUq2Plane); } \
                _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { \
                    fb[BUF][nt][0] = *reinterpret_cast<const half8*>(pw + (TAP) * WT1 + nt * 512); \
                    fb[BUF][nt][1] = *reinterpret_cast<const half8*>(pw + (TAP) * WT1 + nt * 512 + 2 * BN * 16); } }
#define TS2D_TAP(TAP, EXTRA) { constexpr int cur = (TAP) & 1; \
                if constexpr ((TAP) + 1 < 9) TS2D_LOAD_FRAGS(cur ^ 1, (TAP) + 1) \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                    acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][1], fb[cur][nt][0], (TAP) == 0 ? kZero16 : acc_c[mt][nt], 0, 0, 0); \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                    acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][1], acc_c[mt][nt], 0, 0, 0); \
                EXTRA \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                    acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][0], acc_c[mt][nt], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0); }
            TS2D_LOAD_FRAGS(0, 0)
            // (memory operations of a chunk: every use of a loaded register first, THEN the weight DMA - see kernels_f16x3_qp.h)
            TS2D_TAP(0, convert(0, ch1, pb_next);)
            TS2D_TAP(1, convert(1, ch1, pb_next);)
            TS2D_TAP(2, convert(2, ch1, pb_next);)
            TS2D_TAP(3, weights_dma(ch1, wb_next);)
            TS2D_TAP(4, prefetch_unit(0, ch2); prefetch_unit(1, ch2); prefetch_unit(2, ch2);)
            TS2D_TAP(5, ) TS2D_TAP(6, ) TS2D_TAP(7, ) TS2D_TAP(8, )
#undef TS2D_TAP
#undef TS2D_LOAD_FRAGS
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc_t[mt][nt] += acc_c[mt][nt];
            TS2D_STAMP_AT(a.prof, 3)
            // the DMA (older) has landed; the 6 patch loads of chunk ch+2 (younger) stay in flight across the barrier
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TS2D_STAMP_AT(a.prof, 2)
        }
    }

    // =================================================================================== epilogue: scatter by parity, statistics
    // C/D map: column = lane & 31 (channel), row rho = (i & 3) + 8 (i >> 2) + 4 h -> (I = 4 hw + mt + 2 (rho >> 4), J = rho & 15)
    const float oscale = *a.oscale;
    const size_t img_el = (size_t)a.H * a.W * a.Cout;
    const auto rsd = __builtin_amdgcn_make_buffer_rsrc(a.dst + (size_t)nimg0 * img_el, 0, (int)(img_el * 4), 0x00020000);
    const bool edge = tyi == 0 || tyi == a.tiles_y - 1 || txi == 0 || txi == a.tiles_x - 1;       // wave-uniform
    float st_s[NT], st_q[NT], st_k[NT];
    float bv0[NT], bv1[NT], bv2[NT], bv3[NT], bv4[NT], bv5[NT], bv6[NT], bv7[NT], bv8[NT];       // (see kernels_upc.h)
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
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0col + nt * 32 + r;
        const float kv = stat_pivot(__builtin_fmaf(acc_t[0][nt][0], oscale, bv4[nt]));      // shifted statistics (kernels.h)
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int Y0 = ty0 + 8 * hw + 2 * mt + pA;                                 // + 4 (i >> 3)
            const unsigned voff = (unsigned)(((Y0 * a.W + tx0 + 8 * h + pB) * a.Cout + co) * 4);
            if (!edge) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int dI = i >> 3, dJ = (i & 3) + 8 * ((i >> 2) & 1);
                    const unsigned soff = (unsigned)(((4 * dI * a.W + 2 * dJ) * a.Cout) * 4);      // scalar
                    const float v = __builtin_fmaf(acc_t[mt][nt][i], oscale, bv4[nt]);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsd, voff, soff, 0);
                    const float d = v - kv;
                    s += d; q = __builtin_fmaf(d, d, q);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int dI = i >> 3, dJ = (i & 3) + 8 * ((i >> 2) & 1);
                    const unsigned soff = (unsigned)(((4 * dI * a.W + 2 * dJ) * a.Cout) * 4);
                    const int Y = Y0 + 4 * dI, X = tx0 + pB + 2 * (dJ + 4 * h);          // Y wave-uniform, X per lane half
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
    TS2D_STAMP_AT(a.prof, 4)
    float* red = reinterpret_cast<float*>(smem8);           // (all LDS reads ended at the loop's last barrier)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float s = st_s[nt], q = st_q[nt];
        s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
        if (h == 0) stat_wave_put(red, w * BN + nt * 32 + r, s, q, st_k[nt], 64.f);
    }
    lds_barrier();
    if (tid < BN) stat_tile_store(red, 8, BN, tid, a.part + ((size_t)(nimg0 * tpi + tin) * a.Cout + n0col + tid) * 4);
    TS2D_STAMP_AT(a.prof, 5)
    TS2D_PROF_FLUSH(a.prof)
}

}  // namespace ts2d
