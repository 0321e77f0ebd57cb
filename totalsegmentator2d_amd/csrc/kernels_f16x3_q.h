// conv3x3_f16x3_q: the stride-1 3x3 split kernel as ONE 512-thread workgroup per CU with everything double-buffered.
//
// Ablation of conv3x3_f16x3_p<64> on the 512 -> 512 block (gpurun r2, p_abl): whole kernel 0.80 ms, its MFMAs alone 0.60 ms (the
// matrix-pipe rate), its staging alone 0.44 ms - the two phases of a chunk overlap only through the other workgroup of the CU,
// and the weight block (36.9 KB per chunk, L2 -> registers -> LDS between two barriers) is the largest staging item.  Here
//   * the tile is 16 x 32 pixels x 64 channels, 8 waves (wave w = rows 2w, 2w + 1): one weight block serves 512 pixels;
//   * the weights of chunk c+1 travel L2 -> LDS by global_load_lds (no registers, no ds_write) while chunk c computes;
//   * the patch of chunk c+1 (raw values prefetched one chunk earlier) is normalised, split and written to the OTHER patch
//     buffer between the MFMAs of chunk c; the raw values of chunk c+2 are requested right after;
//   * one barrier per chunk (raw s_barrier; the counted s_waitcnt leaves the patch prefetch in flight across it).
// Measured (gpurun r2, q3/q5/q6): 512 -> 512 block 0.80 -> 0.73 ms, 256 -> 256 0.84 -> 0.76, 128 -> 128 0.87 -> 0.83; the 64 -> 64
// block (4 chunks: the two surplus prefetches of the tail weigh more) 0.93 -> 1.00 and stays on conv3x3_f16x3_p.  The same
// kernel with register-staged weights (DBG 64) or with every workgroup reading ONE hot weight block (DBG 128) runs at the same
// speed: neither the DMA nor L2 misses of the 9.4 MB weight set are the limit; the practical roof of an LDS-fed 32x32x16 loop
// (probes/mfma_shape_probe.hip: 1.45-1.5 PFLOP/s, register-fed 1.7) puts this block at 0.62 ms.  (Ablations that leave the
// weight buffer unwritten read lower - zero operands raise the clock - and are not evidence.)
// LDS: 2 x 39168 (patch 18 x 34 slots x 4 planes) + 2 x 36864 (weights) = 152064 bytes.  Tap order, chunking and the
// per-chunk fresh accumulator are those of conv3x3_f16x3_one / _p: the conv outputs are bit-identical; the statistics partials
// are summed over other tiles (16 rows, 8 waves), so scale / shift agree to fp32 rounding only.
#pragma once
#include "kernels_f16x3_p.h"

namespace ts2d {

constexpr int kQThreads = 512, kQRows = 18, kQSlots = kQRows * kPPW, kQPlane = kQSlots * 16, kQPatch = 4 * kQPlane;
constexpr int kQWts = 9 * 4 * 64 * 16, kQLds = 2 * kQPatch + 2 * kQWts;

template <int DBG = 0>      // DBG: timing ablations (1 no MFMA, 2 no conversion, 4 no weight DMA, 8 no patch prefetch) - diagnostic only
__global__ __launch_bounds__(kQThreads, 1) void conv3x3_f16x3_q(const ConvArgs a) {
    constexpr int BN = 64, NT = 2, MAXU = 3, WTAP = 4 * BN * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = bid >> 3;
    const int qm = q8 >> a.lg_nct;
    const int mtile = qm * 8 + xcd;
    const int ctile = q8 - qm * a.n_ctiles;
    if (mtile >= a.n_mtiles) return;
    const int n0col = ctile * BN;
    const int tpi = a.tiles_x * a.tiles_y;
    const int nimg0 = mtile >> a.lg_tpi, tin = mtile - nimg0 * tpi;
    const int tyi = tin >> a.lg_tx, txi = tin - tyi * a.tiles_x;
    const int ty0 = tyi << 4, tx0 = txi << 5;               // TH = 16, TW = 32

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int octi = (lane >> 3) & 1, oct = octi * 8;

    // ---- staging plan (as conv3x3_f16x3_p): pixel = 32 (8 it + w) + (lane & 7) + 8 (lane >> 4), octet = (lane >> 3) & 1
    unsigned poff[MAXU];
    int lw[MAXU];
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int pp = 32 * (8 * it + w) + (lane & 7) + 8 * (lane >> 4);
        unsigned g = ~0u;
        lw[it] = octi * kQPlane + pp * 16;
        if (pp < kQSlots) {
            const int py = pp / kPPW, px = pp - py * kPPW;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) g = (unsigned)(iy * a.Win + ix);
            else {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    *reinterpret_cast<uint4*>(smem8 + b * kQPatch + lw[it]) = uint4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<uint4*>(smem8 + b * kQPatch + lw[it] + 2 * kQPlane) = uint4{0u, 0u, 0u, 0u};
                }
            }
        }
        poff[it] = g;
    }

    const int nchunks = (a.C0 + a.C1) / 16;
    const size_t img_px = (size_t)a.Hin * a.Win;
    unsigned vo0[MAXU], vo1[MAXU];
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        vo0[it] = poff[it] == ~0u ? 0x80000000u : (poff[it] * (unsigned)a.C0 + oct) * 4u;
        vo1[it] = poff[it] == ~0u ? 0x80000000u : (poff[it] * (unsigned)a.C1 + oct) * 4u;
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pv[MAXU][2];
    f32x4 nsa, nsb, nta, ntb;

    // (every source of this kernel is normalised: the engine checks; branch-free - exactly 6 + 4 vector loads per call, the
    //  s_waitcnt below counts on it, and hipcc's own waitcnt pass stays exact instead of falling back to vmcnt(0) at joins)
    const float* const base0 = a.src0 + (size_t)nimg0 * img_px * a.C0;
    const float* const base1 = (a.src1 ? a.src1 : a.src0) + (size_t)nimg0 * img_px * a.C1;
    const float* const sc1p = a.src1 ? a.sc1 : a.sc0;
    const float* const sh1p = a.src1 ? a.sh1 : a.sh0;
    auto prefetch_unit = [&](int it, int ch) {          // 2 buffer loads
        const int cb0 = ch * 16;
        const bool first = cb0 < a.C0;
        const int cb = first ? cb0 : cb0 - a.C0, C = first ? a.C0 : a.C1;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(first ? base0 : base1), 0, (int)(img_px * C * 4), 0x00020000);
        const unsigned vo = first ? vo0[it] : vo1[it];
        pv[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, cb * 4, 0);
        pv[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 16, cb * 4, 0);
    };
    auto prefetch_norm = [&](int ch) {                  // 4 global loads: scale / shift of this thread's 8 channels
        const int cb0 = ch * 16;
        const bool first = cb0 < a.C0;
        const int cb = first ? cb0 : cb0 - a.C0, C = first ? a.C0 : a.C1;
        const float* ps = (first ? a.sc0 : sc1p) + (size_t)nimg0 * C + cb + oct;
        const float* pt = (first ? a.sh0 : sh1p) + (size_t)nimg0 * C + cb + oct;
        nsa = *reinterpret_cast<const f32x4*>(ps); nsb = *reinterpret_cast<const f32x4*>(ps + 4);
        nta = *reinterpret_cast<const f32x4*>(pt); ntb = *reinterpret_cast<const f32x4*>(pt + 4);
    };
    auto prefetch = [&](int ch) {
#pragma unroll
        for (int it = 0; it < MAXU; ++it) prefetch_unit(it, ch);
        prefetch_norm(ch);
    };
    // normalise + LeakyReLU + hi/lo split of unit `it` of the prefetched chunk into patch buffer `pb`
    auto convert = [&](int it, unsigned char* pb) {      // branch-free arithmetic (a padding pixel stores zeros)
        f32x4 va = __builtin_bit_cast(f32x4, pv[it][0]), vb = __builtin_bit_cast(f32x4, pv[it][1]);
        va = va * nsa + nta; vb = vb * nsb + ntb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            va[e] = fmaxf(va[e], va[e] * a.slope);
            vb[e] = fmaxf(vb[e], vb[e] * a.slope);
        }
        uint4 hi, lo;
        split_hi_lo_8(va, vb, hi, lo);
        const bool real = poff[it] != ~0u;
        hi.x = real ? hi.x : 0u; hi.y = real ? hi.y : 0u; hi.z = real ? hi.z : 0u; hi.w = real ? hi.w : 0u;
        lo.x = real ? lo.x : 0u; lo.y = real ? lo.y : 0u; lo.z = real ? lo.z : 0u; lo.w = real ? lo.w : 0u;
        if (it < 2 || lw[it] < octi * kQPlane + kQSlots * 16) {      // (units 0, 1 always exist; 612 .. 767 of unit 2 do not)
            *reinterpret_cast<uint4*>(pb + lw[it]) = hi;
            *reinterpret_cast<uint4*>(pb + lw[it] + 2 * kQPlane) = lo;
        }
    };
    // weight block of chunk `ch` -> weight buffer `wb`: 36 pieces of 1 KiB, wave w takes pieces w, w + 8, ... (LDS-DMA)
    auto weights_dma = [&](int ch, unsigned char* wb) {
        const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wph) + ((DBG & 128) ? (size_t)0 : ((size_t)ch * a.n_ctiles + ctile) * kQWts) + lane * 16;
#pragma unroll
        for (int j = 0; j < ((DBG & 32) ? 1 : 4); ++j) __builtin_amdgcn_global_load_lds(wsrc + (w + 8 * j) * 1024, (lds_ptr)(wb + (w + 8 * j) * 1024), 16, 0, 0);
        // (pieces 32 .. 35 are copied twice - waves w and w + 4, same bytes: every wave issues exactly 5 DMAs, no branch)
        if constexpr (!(DBG & 32)) __builtin_amdgcn_global_load_lds(wsrc + ((w & 3) + 32) * 1024, (lds_ptr)(wb + ((w & 3) + 32) * 1024), 16, 0, 0);
    };

    // register-staged alternative (DBG & 64): 4.5 uint4 per thread, loaded in one tap, written two taps later
    uint4 wr0, wr1, wr2, wr3, wr4;
    auto weights_load = [&](int ch) {
        const uint4* wsrc = reinterpret_cast<const uint4*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * (kQWts / 16);
        wr0 = wsrc[tid]; wr1 = wsrc[tid + 512]; wr2 = wsrc[tid + 1024]; wr3 = wsrc[tid + 1536];
        wr4 = wsrc[2048 + (tid & 255)];           // (units 2048 .. 2303: read twice, written twice with the same bytes - no branch)
    };
    auto weights_store = [&](unsigned char* wb) {
        *reinterpret_cast<uint4*>(wb + tid * 16) = wr0; *reinterpret_cast<uint4*>(wb + (tid + 512) * 16) = wr1;
        *reinterpret_cast<uint4*>(wb + (tid + 1024) * 16) = wr2; *reinterpret_cast<uint4*>(wb + (tid + 1536) * 16) = wr3;
        *reinterpret_cast<uint4*>(wb + (2048 + (tid & 255)) * 16) = wr4;
    };

    unsigned char* const pbuf0 = smem8;
    unsigned char* const wbuf0 = smem8 + 2 * kQPatch;

    // ---- prologue: chunk 0 staged synchronously, chunk 1 requested
    prefetch(0);
    if constexpr (DBG & 64) { weights_load(0); weights_store(wbuf0); } else weights_dma(0, wbuf0);
#pragma unroll
    for (int it = 0; it < MAXU; ++it) convert(it, pbuf0);
    prefetch(nchunks > 1 ? 1 : 0);          // (past the end: the last chunk again - loaded, never used; keeps the loop branch-free)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    TS2D_PROF_DECL(a.prof);
    const int abase = h * kQPlane + ((2 * w) * kPPW + r) * 16;
    const int bbase = h * BN * 16 + r * 16;

    f32x16 acc_t[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_t[mt][nt][i] = 0.f;

    for (int ch = 0; ch < nchunks; ++ch) {
        const int b = ch & 1;
        const unsigned char* pa = smem8 + b * kQPatch + abase;
        const unsigned char* pw = wbuf0 + b * kQWts + bbase;
        unsigned char* pb_next = smem8 + (b ^ 1) * kQPatch;
        unsigned char* wb_next = wbuf0 + (b ^ 1) * kQWts;
        const int ch1 = ch + 1 < nchunks ? ch + 1 : nchunks - 1, ch2 = ch + 2 < nchunks ? ch + 2 : nchunks - 1;

        f32x16 acc_c[2][NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc_c[mt][nt][i] = 0.f;
        half8 fa[2][2][2], fb[2][NT][2];            // [buffer][tile][hi, lo]
#define TS2D_LOAD_FRAGS(BUF, TAP) { \
            constexpr int toff_ = (((TAP) / 3) * kPPW + ((TAP) % 3)) * 16; \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) { \
                fa[BUF][mt][0] = *reinterpret_cast<const half8*>(pa + mt * kPPW * 16 + toff_); \
                fa[BUF][mt][1] = *reinterpret_cast<const half8*>(pa + mt * kPPW * 16 + toff_ + 2 * kQPlane); } \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { \
                fb[BUF][nt][0] = *reinterpret_cast<const half8*>(pw + (TAP) * WTAP + nt * 512); \
                fb[BUF][nt][1] = *reinterpret_cast<const half8*>(pw + (TAP) * WTAP + nt * 512 + 2 * BN * 16); } }
#define TS2D_TAP(TAP, EXTRA) { constexpr int cur = (TAP) & 1; \
            if constexpr ((TAP) + 1 < 9) TS2D_LOAD_FRAGS(cur ^ 1, (TAP) + 1) \
            if constexpr (!(DBG & 1)) { _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][1], fb[cur][nt][0], acc_c[mt][nt], 0, 0, 0); \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][1], acc_c[mt][nt], 0, 0, 0); } \
            EXTRA \
            if constexpr (!(DBG & 1)) { _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][0], acc_c[mt][nt], 0, 0, 0); } \
            __builtin_amdgcn_sched_barrier(0); }
        TS2D_LOAD_FRAGS(0, 0)
        // order of the memory operations of a chunk: the raw values of chunk ch+2 unit by unit as soon as a unit's registers are
        // free, THEN the weight DMA (hipcc answers any use of a loaded register with vmcnt(0) while a DMA is in flight: all uses
        // come first); the s_waitcnt at the end retires everything - the loads are 6+ taps old by then
        if constexpr (DBG & 64) {
            TS2D_TAP(0, convert(0, pb_next); weights_load(ch1);)
            TS2D_TAP(1, convert(1, pb_next);)
            TS2D_TAP(2, convert(2, pb_next);)
            TS2D_TAP(3, )
            TS2D_TAP(4, weights_store(wb_next);)
            TS2D_TAP(5, prefetch_unit(0, ch2); prefetch_unit(1, ch2);)
            TS2D_TAP(6, prefetch_unit(2, ch2); prefetch_norm(ch2);)
            TS2D_TAP(7, ) TS2D_TAP(8, )
        } else {
        TS2D_TAP(0, if constexpr (!(DBG & 2)) convert(0, pb_next); if constexpr (!(DBG & 8)) prefetch_unit(0, ch2);)
        TS2D_TAP(1, if constexpr (!(DBG & 2)) convert(1, pb_next); if constexpr (!(DBG & 8)) prefetch_unit(1, ch2);)
        TS2D_TAP(2, if constexpr (!(DBG & 2)) convert(2, pb_next); if constexpr (!(DBG & 8)) { prefetch_unit(2, ch2); prefetch_norm(ch2); })
        TS2D_TAP(3, if constexpr (!(DBG & 4) && !(DBG & 16)) weights_dma(ch1, wb_next);)
        TS2D_TAP(4, )
        TS2D_TAP(5, ) TS2D_TAP(6, if constexpr (!(DBG & 4) && (DBG & 16)) weights_dma(ch1, wb_next);) TS2D_TAP(7, ) TS2D_TAP(8, )
        }
#undef TS2D_TAP
#undef TS2D_LOAD_FRAGS
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc_t[mt][nt] += acc_c[mt][nt];
        // the DMA of chunk ch+1 has landed (only the patch prefetch of chunk ch+2 may stay in flight), this wave's LDS writes are done
        TS2D_STAMP_AT(a.prof, 1)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        TS2D_STAMP_AT(a.prof, 0)
    }
    TS2D_STAMP_AT(a.prof, 3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: bias, store (C/D map of the 32x32 MFMA: column = lane & 31, rows (i & 3) + 8 (i >> 2) + 4 h), tile statistics
    const float oscale = *a.oscale;
    const size_t img_el = (size_t)a.Ht * a.Wt * a.Cout;
    const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(a.dst) + (size_t)nimg0 * img_el, 0, (int)(img_el * 4), 0x00020000);
    float st_s[NT], st_q[NT];
    float bvs[NT];       // every bias value before the first store: a load issued between stores waits (in-order vmcnt) for the stores ahead of it
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bvs[nt] = a.bias[n0col + nt * 32 + r];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0col + nt * 32 + r;
        const float bv = bvs[nt];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int oy = ty0 + 2 * w + mt, ox = tx0 + 4 * h;
            const unsigned voff = (unsigned)(((oy * a.Wt + ox) * a.Cout + co) * 4);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rowoff = (i & 3) + 8 * (i >> 2);
                const unsigned soff = (unsigned)(rowoff * a.Cout * 4);
                const float v = __builtin_fmaf(acc_t[mt][nt][i], oscale, bv);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsd, voff, soff, 0);
                s += v; q = __builtin_fmaf(v, v, q);
            }
        }
        st_s[nt] = s; st_q[nt] = q;
    }
    TS2D_STAMP_AT(a.prof, 4)
    float* red = reinterpret_cast<float*>(smem8);       // (all LDS reads ended at the loop's last barrier)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float s = st_s[nt], q = st_q[nt];
        s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
        if (h == 0) { red[(w * BN + nt * 32 + r) * 2] = s; red[(w * BN + nt * 32 + r) * 2 + 1] = q; }
    }
    lds_barrier();
    if (tid < BN) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) { s += red[(ww * BN + tid) * 2]; q += red[(ww * BN + tid) * 2 + 1]; }
        float* p = a.part + ((size_t)(nimg0 * tpi + tin) * a.Cout + n0col + tid) * 2;
        p[0] = s; p[1] = q;
    }
    TS2D_STAMP_AT(a.prof, 5)
    TS2D_PROF_FLUSH(a.prof)
}

}  // namespace ts2d
