// conv3x3_hq (round 4): the plain stride-1 3x3 conv of the 16-bit mode (TS2D_PRECISION_F16: fp16-stored activations, fp16 weights, ONE
// v_mfma_f32_32x32x16_f16 product per MAC, fp32 accumulation over the whole K, fp32 statistics of the stored values) in the structure
// of conv3x3_f16x3_qp: ONE persistent 512-thread workgroup per CU, tile = 16 x 32 pixels x 64 columns, a stream of (tile, chunk) items
// with the MFMAs of item i, the conversion + weight DMA of item i + 1 and the raw loads of item i + 2 in one barrier interval.
// conv3x3_h2 (the skip phase of conv3x3_upc_h2, kernels_upc_h2.h) served these layers un-pipelined - convert | barrier | 72 MFMAs with
// two 256-thread workgroups per CU - at 0.35 of the fp16 MFMA roof.
//
// A chunk is 32 channels = 64 bytes of a pixel record: exactly the bytes of a split-mode chunk of 16 (hi + lo), so the LDS map of
// conv3x3_f16x3_qp is reused one to one with the PARTS re-read as K-STEPS: patch plane[ks 2][h 2][slot] (k-step ks = channels 16 ks ..,
// h = the 8-channel half a lane half feeds), weights plane[ks][tap][h][column], 2 x 39168 + 2 x 36864 bytes.  The weight block of a
// chunk = the hi parts of two 16-channel blocks of the split image (engine.hip: dev_wp), fetched as 36 pieces of 1 KiB by
// global_load_lds.  Staging: four lanes per pixel, 16 bytes = 8 channels each (a wave load covers 16 pixels = 16 lines, kernels_s2v2.h),
// five units per thread, converted one per tap (norm_lrelu_8: fp32 FMA from the fp16 value, one rounding, LeakyReLU in packed fp16) and
// re-requested for the item after next right behind the conversion; weight DMA at tap 5.  72 MFMAs and 72 ds_read_b128 per wave and item.
#pragma once
#include "kernels_f16x3_qp.h"
#include "kernels_h32.h"

namespace ts2d {

constexpr int kHqBias = kQLds + 8192, kHqLds = kHqBias + 256;      // LDS map of conv3x3_f16x3_qp + the statistics exchange + the column tile's bias

// ABL: timing ablations of diagnostic runs (TS2D_DBG bits 12..15 -> ABL bits 0..3; results are WRONG): 1 = no MFMAs, 2 = no patch loads in
// the loop, 4 = no weight DMA in the loop, 8 = no conversion / LDS writes in the loop
template <int ABL = 0>
__global__ __launch_bounds__(kQThreads, 1) void conv3x3_hq(const ConvArgs a) {
    constexpr int BN = 64, NT = 2, MAXU = 5, WTAP = 2 * BN * 16;         // bytes per (k-step, tap) of the LDS weight image: [h][column]
    constexpr int WSRC = 4 * BN * 16;                                    // bytes per (chunk16, tap) of the split image in HBM: [hi, lo][h][column]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef _Float16 ST;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int r = lane & 31, h = lane >> 5;
    const int sub = tid & 3;                                 // this thread's quarter of a chunk slice: channels 8 sub .. 8 sub + 7 = plane (ks, h) = (sub >> 1, sub & 1)

    // ---- this workgroup's tiles (as conv3x3_f16x3_qp)
    const int xcd = blockIdx.x & 7, q80 = blockIdx.x >> 3;
    const int ctile = q80 & (a.n_ctiles - 1), n0col = ctile * BN;
    const int mtile0 = (q80 >> a.lg_nct) * 8 + xcd, mstep = ((int)(gridDim.x >> 3) >> a.lg_nct) * 8;
    if (mtile0 >= a.n_mtiles) return;
    const int ntl = (a.n_mtiles - 1 - mtile0) / mstep + 1;
    const int nchunks = a.C0 / 32;
    const int tpi = a.tiles_x * a.tiles_y;
    const size_t img_px = (size_t)a.Hin * a.Win;

    // ---- staging units: unit it = patch slot (tid >> 2) + 128 it -> (py, px), tile-independent; packed 5 + 6 bits per unit (py = 31: no unit)
    unsigned upk0 = 0, upk1 = 0;
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int pp = (tid >> 2) + 128 * it;
        const int py = pp < kQSlots ? pp / kPPW : 31, px = pp < kQSlots ? pp - py * kPPW : 0;
        const unsigned f = (unsigned)((py << 6) | px);
        if (it < 3) upk0 |= f << (11 * it); else upk1 |= f << (11 * (it - 3));
    }
    auto unit_py = [&](int it) { return (int)(((it < 3 ? upk0 >> (11 * it) : upk1 >> (11 * (it - 3))) >> 6) & 31u); };
    auto unit_px = [&](int it) { return (int)((it < 3 ? upk0 >> (11 * it) : upk1 >> (11 * (it - 3))) & 63u); };
    const int lw0 = sub * kQPlane + (tid >> 2) * 16;         // LDS write address of unit 0; unit it: + 2048 it
    struct Item { int k, c; };
    auto advance = [&](Item& t) {                            // next item of the stream; the last item repeats (loaded / staged, never used)
        int c = t.c + 1, k = t.k;
        if (c == nchunks) { c = 0; ++k; }
        if (k < ntl) { t.k = k; t.c = c; }
    };
    auto tile_origin = [&](int k, int& nimg, int& ty0, int& tx0, int& tin) {
        const int mtile = mtile0 + k * mstep;
        nimg = mtile >> a.lg_tpi; tin = mtile - nimg * tpi;
        const int tyi = tin >> a.lg_tx, txi = tin - tyi * a.tiles_x;
        ty0 = tyi << 4; tx0 = txi << 5;
    };

    u32x4 pv[MAXU];
    f32x4 nsa, nsb, nta, ntb;
    unsigned real_pf = 0;                                    // bit it: unit it of the item in the registers lies inside the image
    const _Float16 slope_h = (_Float16)a.slope;
    const unsigned slope2 = (unsigned)__builtin_bit_cast(unsigned short, slope_h) * 0x10001u;
    struct Req { __amdgpu_buffer_rsrc_t rs; int ty0, tx0, cb; const float* ps; const float* pt; };
    auto request = [&](const Item& t) {
        int nimg, tin;
        Req q;
        tile_origin(t.k, nimg, q.ty0, q.tx0, tin);
        q.cb = t.c * 32;
        q.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(reinterpret_cast<const ST*>(a.src0)) + (size_t)nimg * img_px * a.C0, 0,
                                                 (int)(img_px * a.C0 * 2), 0x00020000);
        q.ps = a.sc0 + (size_t)nimg * a.C0 + q.cb + 8 * sub;
        q.pt = a.sh0 + (size_t)nimg * a.C0 + q.cb + 8 * sub;
        return q;
    };
    auto load_unit = [&](const Req& q, int it) {             // one buffer load, branch-free
        const int py = unit_py(it), px = unit_px(it);
        const int iy = q.ty0 - 1 + py, ix = q.tx0 - 1 + px;
        const bool in = (py < kQRows) & ((unsigned)iy < (unsigned)a.Hin) & ((unsigned)ix < (unsigned)a.Win);
        const unsigned vo = in ? (unsigned)(((iy * a.Win + ix) * a.C0) * 2 + 16 * sub) : 0x80000000u;
        pv[it] = __builtin_amdgcn_raw_buffer_load_b128(q.rs, vo, q.cb * 2, 0);
        real_pf = (real_pf & ~(1u << it)) | (in ? (1u << it) : 0u);
    };
    auto load_norm = [&](const Req& q) {
        nsa = *reinterpret_cast<const f32x4*>(q.ps); nsb = *reinterpret_cast<const f32x4*>(q.ps + 4);
        nta = *reinterpret_cast<const f32x4*>(q.pt); ntb = *reinterpret_cast<const f32x4*>(q.pt + 4);
    };
    auto convert = [&](int it, unsigned char* pb) {          // a padding pixel stores zeros (AFTER norm + activation)
        uint4 x = norm_lrelu_8(uint4{pv[it][0], pv[it][1], pv[it][2], pv[it][3]}, nsa, nsb, nta, ntb, slope2);
        const bool real = (real_pf >> it) & 1u;
        x.x = real ? x.x : 0u; x.y = real ? x.y : 0u; x.z = real ? x.z : 0u; x.w = real ? x.w : 0u;
        if (it < MAXU - 1 || unit_py(MAXU - 1) < kQRows) *reinterpret_cast<uint4*>(pb + lw0 + it * 2048) = x;
    };
    // weights of chunk ch: LDS piece p (1 KiB) = half (p & 1) of the [h][column] block of (k-step, tap) = (p >> 1) / 9, (p >> 1) % 9; its
    // source = the hi part of 16-channel block 2 ch + ks of the split image.  36 pieces; every wave issues exactly 5 (pieces 32..35 twice)
    auto weights_dma = [&](int ch, unsigned char* wb) {
        auto piece = [&](int p) {                            // (wave-uniform)
            const int kt = p >> 1, ks = kt / 9, tap = kt - 9 * ks;
            const unsigned char* src = reinterpret_cast<const unsigned char*>(a.wph) + (((size_t)(2 * ch + ks) * a.n_ctiles + ctile) * 9 + tap) * WSRC + (p & 1) * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds(src, (lds_ptr)(wb + p * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) piece(w + 8 * j);
        piece((w & 3) + 32);
    };

    unsigned char* const wbuf0 = smem8 + 2 * kQPatch;
    float* const lbias = reinterpret_cast<float*>(smem8 + kHqBias);      // the 64 bias values of this column tile (visible behind the fill barrier)
    if (tid < BN) lbias[tid] = a.bias[n0col + tid];
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

    const int abase = h * kQPlane + ((2 * w) * kPPW + r) * 16;           // + ks * 2 planes + mt * row + tap offset
    const int bbase = h * BN * 16 + r * 16;                              // + (ks * 9 + tap) * WTAP + nt * 512

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    if (!(a.dbg & 512)) { if (w >= 4) __builtin_amdgcn_s_setprio(1); }      // static issue priority for the younger half (kernels_f16x3_qp.h)
    const int nitems = ntl * nchunks;
    int pend = -1;                                           // statistics of a finished tile waiting for the item barrier
    const float oscale = *a.oscale;
    for (int i = 0; i < nitems; ++i) {
        const int b = i & 1;
        const unsigned char* pa = smem8 + b * kQPatch + abase;
        const unsigned char* pw = wbuf0 + b * kQWts + bbase;
        unsigned char* pb_next = smem8 + (b ^ 1) * kQPatch;
        unsigned char* wb_next = wbuf0 + (b ^ 1) * kQWts;

        half8 fa[2][2][2], fb[2][NT][2];                     // [buffer][tile][k-step]
#define TS2D_LOAD_FRAGS(BUF, TAP) { \
            constexpr int toff_ = (((TAP) / 3) * kPPW + ((TAP) % 3)) * 16; \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) { \
                fa[BUF][mt][0] = *reinterpret_cast<const half8*>(pa + mt * kPPW * 16 + toff_); \
                fa[BUF][mt][1] = *reinterpret_cast<const half8*>(pa + mt * kPPW * 16 + toff_ + 2 * kQPlane); } \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { \
                fb[BUF][nt][0] = *reinterpret_cast<const half8*>(pw + (TAP) * WTAP + nt * 512); \
                fb[BUF][nt][1] = *reinterpret_cast<const half8*>(pw + (9 + (TAP)) * WTAP + nt * 512); } }
#define TS2D_TAP(TAP, EXTRA) { constexpr int cur_ = (TAP) & 1; \
            if constexpr ((TAP) + 1 < 9) TS2D_LOAD_FRAGS(cur_ ^ 1, (TAP) + 1) \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                if constexpr (!(ABL & 1)) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur_][mt][0], fb[cur_][nt][0], acc[mt][nt], 0, 0, 0); \
                else { acc[mt][nt][0] += (float)fa[cur_][mt][0][0] * (float)fb[cur_][nt][0][0]; } \
            EXTRA \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                if constexpr (!(ABL & 1)) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur_][mt][1], fb[cur_][nt][1], acc[mt][nt], 0, 0, 0); \
                else { acc[mt][nt][1] += (float)fa[cur_][mt][1][0] * (float)fb[cur_][nt][1][0]; } \
            __builtin_amdgcn_sched_barrier(0); }
        const Req rq = request(nx2);                         // the item after next: each unit re-requested right behind its conversion
        TS2D_LOAD_FRAGS(0, 0)
        // (every use of a loaded register first, THEN the weight DMA - see kernels_f16x3_qp.h)
        TS2D_TAP(0, if constexpr (!(ABL & 8)) convert(0, pb_next); if constexpr (!(ABL & 2)) load_unit(rq, 0);)
        TS2D_TAP(1, if constexpr (!(ABL & 8)) convert(1, pb_next); if constexpr (!(ABL & 2)) load_unit(rq, 1);)
        TS2D_TAP(2, if constexpr (!(ABL & 8)) convert(2, pb_next); if constexpr (!(ABL & 2)) load_unit(rq, 2);)
        TS2D_TAP(3, if constexpr (!(ABL & 8)) convert(3, pb_next); if constexpr (!(ABL & 2)) load_unit(rq, 3);)
        TS2D_TAP(4, if constexpr (!(ABL & 8)) convert(4, pb_next); if constexpr (!(ABL & 2)) load_unit(rq, 4); load_norm(rq);)
        TS2D_TAP(5, if constexpr (!(ABL & 4)) weights_dma(nx1.c, wb_next);)
        TS2D_TAP(6, ) TS2D_TAP(7, ) TS2D_TAP(8, )
#undef TS2D_TAP
#undef TS2D_LOAD_FRAGS

        if (cur.c == nchunks - 1) {                          // (uniform) the tile is complete: bias, fp16 stores, statistics of the stored values
            int nimg, ty0, tx0, tin;
            tile_origin(cur.k, nimg, ty0, tx0, tin);
            const size_t img_el = (size_t)a.Ht * a.Wt * a.Cout;
            const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<ST*>(a.dst) + (size_t)nimg * img_el, 0, (int)(img_el * 2), 0x00020000);
            float st_s[NT], st_q[NT], st_k[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = n0col + nt * 32 + r;
                const float bv = lbias[nt * 32 + r];
                const float kv = stat_pivot(round_act<ST>(__builtin_fmaf(acc[0][nt][0], oscale, bv)));      // shifted statistics (kernels.h)
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int oy = ty0 + 2 * w + mt, ox = tx0 + 4 * h;
                    const unsigned voff = (unsigned)(((oy * a.Wt + ox) * a.Cout + co) * 2);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rowoff = (e & 3) + 8 * (e >> 2);
                        const unsigned soff = (unsigned)(rowoff * a.Cout * 2);
                        const float v = __builtin_fmaf(acc[mt][nt][e], oscale, bv);
                        buffer_store_act<ST>(v, rsd, voff, soff);
                        const float d = round_act<ST>(v) - kv;
                        s += d; q = __builtin_fmaf(d, d, q);
                        acc[mt][nt][e] = 0.f;
                    }
                }
                st_s[nt] = s; st_q[nt] = q; st_k[nt] = kv;
            }
            float* red = reinterpret_cast<float*>(smem8 + kQpRed);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float s = st_s[nt], q = st_q[nt];
                s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
                if (h == 0) stat_wave_put(red, w * BN + nt * 32 + r, s, q, st_k[nt], 64.f);
            }
            pend = (nimg * tpi + tin) * a.Cout + n0col;      // merged behind the item barrier (kernels_f16x3_qp.h)
        }
        advance(cur); advance(nx1); advance(nx2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (pend >= 0) {                                     // (uniform)
            if (tid < BN) stat_tile_store(reinterpret_cast<const float*>(smem8 + kQpRed), 8, BN, tid, a.part + ((size_t)pend + tid) * 4);
            pend = -1;
        }
    }
}

}  // namespace ts2d
