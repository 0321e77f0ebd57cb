// conv3x3s2_q (round 4): the strided-conv downsample (Conv2d 3x3 stride 2 pad 1; enc2.c0 .. enc4.c0 of the canonical net, 128 output
// columns per workgroup) in the structure of conv3x3_f16x3_qp: ONE persistent 512-thread workgroup per CU, a stream of (tile, chunk)
// items with the MFMAs of item i, the conversion + weight DMA of item i + 1 and the raw loads of item i + 2 in one barrier interval.
//
// conv3x3s2_v2 (kernels_s2v2.h) converts an item's whole patch BETWEEN two barriers and then issues its MFMAs: a stride-2 conv stages four
// input pixels per output pixel, the conversion is a third of an item (stamps: 4.6 k of 14.4 k cycles, enc2.c0) and the matrix pipe idles
// through it - its 72 KB patch and 72 KB weight block leave no room for a second buffer.  Here a chunk is EIGHT input channels:
//   * patch buffer = [part hi, lo][slot] x 16 bytes (a slot = the 8 channels of one patch pixel), 17 rows x (33 even | 33 odd columns),
//     36 KB - two of them fit beside two weight buffers;
//   * one MFMA k-step = 8 channels x TWO taps: lane half h reads tap 2 s + h (a per-lane address constant), five k-steps per chunk, the
//     second half of the fifth multiplies zero weights (10 % of the MFMAs; the price of the second buffer);
//   * weights [k-step 5][part][h][column 128] x 16 bytes = 40 KB per (chunk, column tile), stored in HBM in LDS order, 40 pieces of 1 KiB
//     by global_load_lds (5 per wave);
//   * staging: two lanes per pixel (16 bytes = 4 channels each), five units per thread: two converted + re-requested per k-step in
//     k-steps 0-1, one in k-step 2, the weight DMA in k-step 3 (every use of a loaded register precedes it, kernels_f16x3_qp.h);
//   * 8 waves = 4 (pixel rows 2 wm, 2 wm + 1 of the 8 x 32 tile) x 2 (64 columns each), 60 MFMAs per wave and item, one raw barrier per item.
// Arithmetic: split mode only (fp32 storage, x = hi + lo, three products, fresh accumulator per chunk of 8 channels x 9 taps).
#pragma once
#include "kernels_s2v2.h"

namespace ts2d {

constexpr int kSqThreads = 512, kSqPW = 66, kSqSlots = 17 * kSqPW, kSqPlane = (kSqSlots + 2) * 16, kSqPatch = 2 * kSqPlane;
constexpr int kSqWts = 5 * 2 * 2 * 128 * 16;                                      // weight bytes per (chunk, column tile)
constexpr int kSqRed = 2 * kSqPatch + 2 * kSqWts, kSqLds = kSqRed + 8192;          // 2 x 35968 + 2 x 40960 + the statistics exchange [4][128] x 16 B = 162048 bytes

// ABL: timing ablations of diagnostic runs (TS2D_DBG bits 12..15; results are WRONG): 1 = no MFMAs, 2 = no patch loads in the loop, 4 = no weight
// DMA in the loop, 8 = no conversion / LDS writes in the loop
template <int ABL = 0>
__global__ __launch_bounds__(kSqThreads, 1) void conv3x3s2_q(const ConvArgs a) {
    constexpr int BN = 128, NT = 2, MAXU = 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6) & 7, wm = w & 3, wn = w >> 2;
    const int r = lane & 31, h = lane >> 5;
    const int sub = tid & 1;                                 // this thread's half of a pixel's 8-channel slice: channels 4 sub .. 4 sub + 3

    // ---- this workgroup's tiles (as conv3x3_f16x3_qp / conv3x3s2_v2)
    const int xcd = blockIdx.x & 7, q80 = blockIdx.x >> 3;
    const int ctile = q80 & (a.n_ctiles - 1), n0col = ctile * BN;
    const int mtile0 = (q80 >> a.lg_nct) * 8 + xcd, mstep = ((int)(gridDim.x >> 3) >> a.lg_nct) * 8;
    if (mtile0 >= a.n_mtiles) return;
    const int ntl = (a.n_mtiles - 1 - mtile0) / mstep + 1;
    const int nchunks = a.C0 / 8;
    const int tpi = a.tiles_x * a.tiles_y;
    const size_t img_px = (size_t)a.Hin * a.Win;

    // ---- staging plan (tile-independent): unit it = patch slot (tid >> 1) + 256 it; slot q = row q / 66, then the 33 even columns, then the odd
    //      ones (the 66th slot of a row is unused and read by no fragment)
    unsigned rel[MAXU];                                      // byte offset of the unit from the patch origin, 0x80000000 = no unit
    unsigned emask = 0;                                      // per unit: bit 0 = patch row 0, bit 1 = patch column 0 (the padding candidates)
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int q = (tid >> 1) + 256 * it;
        const int py = q / kSqPW, rem = q - py * kSqPW;
        const int half = rem >= 33 ? 1 : 0, px = 2 * (rem - 33 * half) + half;
        const bool exists = q < kSqSlots && px <= 64;
        rel[it] = exists ? (unsigned)(((py * a.Win + px) * a.C0) * 4 + 16 * sub) : 0x80000000u;
        if (exists) emask |= ((py == 0 ? 1u : 0u) | (px == 0 ? 2u : 0u)) << (2 * it);
    }
    const int lw0 = (tid >> 1) * 16 + sub * 8;               // LDS write address of unit 0 (hi plane); unit it: + 4096 it; lo plane: + kSqPlane

    struct Item { int k, c; };
    auto advance = [&](Item& t) {                            // next item of the stream; the last item repeats (loaded / staged, never used)
        int c = t.c + 1, k = t.k;
        if (c == nchunks) { c = 0; ++k; }
        if (k < ntl) { t.k = k; t.c = c; }
    };
    auto tile_origin = [&](int k, int& nimg, int& tyi, int& txi, int& tin) {
        const int mtile = mtile0 + k * mstep;
        nimg = mtile >> a.lg_tpi; tin = mtile - nimg * tpi;
        tyi = tin >> a.lg_tx; txi = tin - tyi * a.tiles_x;
    };

    u32x4 pv[MAXU];
    f32x4 nsa, nta;
    unsigned real_pf = 0;                                    // bit it: unit it of the item in the registers lies inside the image (set by its load)
    struct Req { __amdgpu_buffer_rsrc_t rs; unsigned org, pad; int soff; const float* ps; const float* pt; };
    auto request = [&](const Item& t) {
        int nimg, tyi, txi, tin;
        tile_origin(t.k, nimg, tyi, txi, tin);
        Req q;
        q.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0) + (size_t)nimg * img_px * a.C0, 0, (int)(img_px * a.C0 * 4), 0x00020000);
        // patch origin (2 ty0 - 1, 2 tx0 - 1) may lie one row / column outside the image: unsigned wrap-around is fine, the affected units are padding
        q.org = (unsigned)((((16 * tyi - 1) * a.Win + 64 * txi - 1) * a.C0) * 4);
        q.pad = (tyi == 0 ? 0x155u : 0u) | (txi == 0 ? 0x2AAu : 0u);      // which emask bits mean "outside the image" for this tile
        q.soff = t.c * 32;
        q.ps = a.sc0 + (size_t)nimg * a.C0 + t.c * 8 + 4 * sub;
        q.pt = a.sh0 + (size_t)nimg * a.C0 + t.c * 8 + 4 * sub;
        return q;
    };
    auto load_unit = [&](const Req& q, int it) {
        const bool in = rel[it] != 0x80000000u && ((emask & q.pad) >> (2 * it) & 3u) == 0u;
        const unsigned vo = in ? q.org + rel[it] : 0x80000000u;
        pv[it] = __builtin_amdgcn_raw_buffer_load_b128(q.rs, vo, q.soff, 0);
        real_pf = (real_pf & ~(1u << it)) | (in ? (1u << it) : 0u);
    };
    auto load_norm = [&](const Req& q) {
        nsa = *reinterpret_cast<const f32x4*>(q.ps); nta = *reinterpret_cast<const f32x4*>(q.pt);
    };
    auto convert = [&](int it, unsigned char* pb) {          // branch-free arithmetic (a padding pixel stores zeros AFTER norm + activation)
        f32x4 va = __builtin_bit_cast(f32x4, pv[it]);
        va = va * nsa + nta;
#pragma unroll
        for (int e = 0; e < 4; ++e) va[e] = fmaxf(va[e], va[e] * a.slope);
        uint2 hi, lo;
        split_hi_lo_4(va, hi, lo);
        const bool real = (real_pf >> it) & 1u;
        hi.x = real ? hi.x : 0u; hi.y = real ? hi.y : 0u;
        lo.x = real ? lo.x : 0u; lo.y = real ? lo.y : 0u;
        if (it < MAXU - 1 || (tid >> 1) + 256 * (MAXU - 1) < kSqSlots) {
            *reinterpret_cast<uint2*>(pb + lw0 + it * 4096) = hi;
            *reinterpret_cast<uint2*>(pb + lw0 + it * 4096 + kSqPlane) = lo;
        }
    };
    auto weights_dma = [&](int ch, unsigned char* wb) {     // 40 pieces of 1 KiB; every wave issues exactly 5
        const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * kSqWts + lane * 16;
#pragma unroll
        for (int j = 0; j < 5; ++j) __builtin_amdgcn_global_load_lds(wsrc + (w + 8 * j) * 1024, (lds_ptr)(wb + (w + 8 * j) * 1024), 16, 0, 0);
    };

    unsigned char* const wbuf0 = smem8 + 2 * kSqPatch;
    // ---- fill the pipeline: item 0 staged synchronously (once per workgroup), item 1 requested
    Item cur{0, 0}, nx1{0, 0}, nx2{0, 0};
    {
        const Req q0 = request(cur);
#pragma unroll
        for (int it = 0; it < MAXU; ++it) load_unit(q0, it);
        load_norm(q0);
    }
    weights_dma(0, wbuf0);
#pragma unroll
    for (int it = 0; it < MAXU; ++it) convert(it, smem8);
    advance(nx1);
    nx2 = nx1;
    {
        const Req q1 = request(nx1);
#pragma unroll
        for (int it = 0; it < MAXU; ++it) load_unit(q1, it);
        load_norm(q1);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    advance(nx2);

    // ---- lane constants of the MFMA phase: output pixel (2 wm + mt, r), tap (dy, dx) reads patch row 2 (2 wm + mt) + dy, slot (dx & 1) 33 + r + (dx >> 1);
    //      lane half h feeds tap 2 s + h of k-step s (tap 9 does not exist: its weights are zero, the lane re-reads tap 8)
    const int abase = ((4 * wm) * kSqPW + r) * 16;           // + mt * 2 * 66 * 16 + part * kSqPlane + tap offset
    const int bbase = h * 2048 + (wn * 64 + r) * 16;         // + s * 8192 + part * 4096 + nt * 512
    auto tap_off = [](int t) { const int dy = t / 3, dx = t - 3 * dy; return (dy * kSqPW + (dx & 1) * 33 + (dx >> 1)) * 16; };

    f32x16 acc_t[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_t[mt][nt][i] = 0.f;

    if (!(a.dbg & 512)) { if (w >= 4) __builtin_amdgcn_s_setprio(1); }      // static issue priority for the younger half (kernels_f16x3_qp.h)
    const int nitems = ntl * nchunks;
    int pend = -1;                                           // statistics of a finished tile waiting for the item barrier: its entry in a.part
    for (int i = 0; i < nitems; ++i) {
        const int b = i & 1;
        const unsigned char* pa = smem8 + b * kSqPatch + abase;
        const unsigned char* pw = wbuf0 + b * kSqWts + bbase;
        unsigned char* pb_next = smem8 + (b ^ 1) * kSqPatch;
        unsigned char* wb_next = wbuf0 + (b ^ 1) * kSqWts;

        f32x16 acc_c[2][NT];
        half8 fa[2][2][2], fb[2][NT][2];                     // [buffer][tile][hi, lo]
#define TS2D_LOAD_FRAGS(BUF, S) { \
            const int toff_ = h ? tap_off((2 * (S) + 1) < 9 ? 2 * (S) + 1 : 8) : tap_off(2 * (S)); \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) { \
                fa[BUF][mt][0] = *reinterpret_cast<const half8*>(pa + mt * 2 * kSqPW * 16 + toff_); \
                fa[BUF][mt][1] = *reinterpret_cast<const half8*>(pa + mt * 2 * kSqPW * 16 + toff_ + kSqPlane); } \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { \
                fb[BUF][nt][0] = *reinterpret_cast<const half8*>(pw + (S) * 8192 + nt * 512); \
                fb[BUF][nt][1] = *reinterpret_cast<const half8*>(pw + (S) * 8192 + nt * 512 + 4096); } }
#define TS2D_STEP(S, EXTRA) { constexpr int cur_ = (S) & 1; \
            if constexpr ((S) + 1 < 5) TS2D_LOAD_FRAGS(cur_ ^ 1, (S) + 1) \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                if constexpr (!(ABL & 1)) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur_][mt][1], fb[cur_][nt][0], (S) == 0 ? kZero16 : acc_c[mt][nt], 0, 0, 0); \
                else { if ((S) == 0) acc_c[mt][nt] = kZero16; acc_c[mt][nt][0] += (float)fa[cur_][mt][1][0] * (float)fb[cur_][nt][0][0] + (float)fa[cur_][mt][0][0] * (float)fb[cur_][nt][1][0]; } \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                if constexpr (!(ABL & 1)) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur_][mt][0], fb[cur_][nt][1], acc_c[mt][nt], 0, 0, 0); \
            EXTRA \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                if constexpr (!(ABL & 1)) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur_][mt][0], fb[cur_][nt][0], acc_c[mt][nt], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); }
        const Req rq = request(nx2);                         // the item after next: each unit re-requested right behind its conversion
        TS2D_LOAD_FRAGS(0, 0)
        TS2D_STEP(0, if constexpr (!(ABL & 8)) convert(0, pb_next); if constexpr (!(ABL & 2)) load_unit(rq, 0); if constexpr (!(ABL & 8)) convert(1, pb_next); if constexpr (!(ABL & 2)) load_unit(rq, 1);)
        TS2D_STEP(1, if constexpr (!(ABL & 8)) convert(2, pb_next); if constexpr (!(ABL & 2)) load_unit(rq, 2); if constexpr (!(ABL & 8)) convert(3, pb_next); if constexpr (!(ABL & 2)) load_unit(rq, 3);)
        TS2D_STEP(2, if constexpr (!(ABL & 8)) convert(4, pb_next); if constexpr (!(ABL & 2)) load_unit(rq, 4); load_norm(rq);)
        TS2D_STEP(3, if constexpr (!(ABL & 4)) weights_dma(nx1.c, wb_next);)
        TS2D_STEP(4, )
#undef TS2D_STEP
#undef TS2D_LOAD_FRAGS
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc_t[mt][nt] += acc_c[mt][nt];

        if (cur.c == nchunks - 1) {                          // (uniform) the tile is complete: bias, store, statistics; the next items' staging is in flight
            int nimg, tyi, txi, tin;
            tile_origin(cur.k, nimg, tyi, txi, tin);
            const int ty0 = tyi * 8, tx0 = txi * 32;
            const size_t img_el = (size_t)a.Ht * a.Wt * a.Cout;
            const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(a.dst) + (size_t)nimg * img_el, 0, (int)(img_el * 4), 0x00020000);
            float st_s[NT], st_q[NT], st_k[NT], bvs[NT];
            const float oscale = *a.oscale;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bvs[nt] = a.bias[n0col + wn * 64 + nt * 32 + r];       // (both before the first store)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = n0col + wn * 64 + nt * 32 + r;
                const float kv = stat_pivot(__builtin_fmaf(acc_t[0][nt][0], oscale, bvs[nt]));      // shifted statistics (kernels.h)
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int oy = ty0 + 2 * wm + mt, ox = tx0 + 4 * h;
                    const unsigned voff = (unsigned)(((oy * a.Wt + ox) * a.Cout + co) * 4);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const unsigned soff = (unsigned)((((e & 3) + 8 * (e >> 2)) * a.Cout) * 4);      // scalar
                        const float v = __builtin_fmaf(acc_t[mt][nt][e], oscale, bvs[nt]);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsd, voff, soff, 0);
                        const float d = v - kv;
                        s += d; q = __builtin_fmaf(d, d, q);
                        acc_t[mt][nt][e] = 0.f;
                    }
                }
                st_s[nt] = s; st_q[nt] = q; st_k[nt] = kv;
            }
            float* red = reinterpret_cast<float*>(smem8 + kSqRed);      // [wm 4][column BN] x (S, Q, K, n)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float s = st_s[nt], q = st_q[nt];
                s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
                if (h == 0) stat_wave_put(red, wm * BN + wn * 64 + nt * 32 + r, s, q, st_k[nt], 64.f);
            }
            pend = (nimg * tpi + tin) * a.Cout + n0col;      // merged behind the item barrier (kernels_f16x3_qp.h); `red` is not written again before the next tile's epilogue
        }
        advance(cur); advance(nx1); advance(nx2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (pend >= 0) {                                     // (uniform)
            if (tid < BN) stat_tile_store(reinterpret_cast<const float*>(smem8 + kSqRed), 4, BN, tid, a.part + ((size_t)pend + tid) * 4);
            pend = -1;
        }
    }
}

}  // namespace ts2d
