// conv3x3_h_qp16: the 16-bit mode (fp16 storage, ONE fp16 product, fp32 accumulation and statistics: BASELINE configs 3 / 5) in the
// persistent pipeline of conv3x3_f16x3_qp16 - VERDICT r2 item 7: "its stride-1 layers still run the un-pipelined conv3x3_h32".
// Here v_mfma_f32_16x16x32_f16 fits without any packing: a chunk is 32 REAL channels, k-group g = lane >> 4 = channels 8 g .. 8 g + 7
// of the chunk on both operands, 9 k-steps x 16 blocks = 144 MFMAs of 16 cycles per chunk and wave (2 304 cycles for 32 channels; the
// split form needs 3 584 for 16), 72 ds_read_b128.  A 32-channel chunk of fp16 is exactly as large as a 16-channel hi + lo chunk:
// the same LDS map (2 x 39 936 patch + 2 x 36 864 weights + statistics), the same 36-piece weight DMA.
//   patch    plane[g][pixel slot], 624 slots of 16 bytes (a multiple of 256 bytes: conflict-free 16-lane fragment rows)
//   weights  [chunk32][column tile][tap][g][column 64][8 halves] in HBM = LDS order (engine.hip: dev_wq16h; the hi parts of the split
//            weights, pre-scaled by the layer's power of two like every image)
// Staging: a pixel's chunk is 64 contiguous bytes of the NHWC fp16 tensor = 4 units of 16 bytes; a wave instruction covers 16 pixels
// x 4 k-groups (8 consecutive lanes write 8 consecutive slots of one plane); 5 units per thread; InstanceNorm + LeakyReLU as packed
// fp16 (norm_lrelu_8, kernels_h32.h).  Transposed product D[cout][pixel] as in qp16: 16 stores of 8 bytes per lane and tile.
#pragma once
#include "kernels_f16x3_qp16.h"
#include "kernels_h32.h"

namespace ts2d {

__global__ __launch_bounds__(kQThreads, 1) void conv3x3_h_qp16(const ConvArgs a) {
    constexpr int BN = 64, MAXU = 5, WTAP = 4 * BN * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int j = lane & 15, g = lane >> 4;                  // MFMA lane roles: row / column index, k-group
    const int sg = (lane >> 3) & 3;                          // staging role: k-group of the unit

    // ---- this workgroup's tiles: virtual block v = blockIdx.x + k * gridDim.x -> (xcd, column tile) fixed, pixel tile mtile0 + k * mstep
    const int xcd = blockIdx.x & 7, q80 = blockIdx.x >> 3;
    const int ctile = q80 & (a.n_ctiles - 1), n0col = ctile * BN;
    const int mtile0 = (q80 >> a.lg_nct) * 8 + xcd, mstep = ((int)(gridDim.x >> 3) >> a.lg_nct) * 8;
    if (mtile0 >= a.n_mtiles) return;
    const int ntl = (a.n_mtiles - 1 - mtile0) / mstep + 1;                 // tiles of this workgroup
    const int nchunks = (a.C0 + a.C1) / 32;                                // (the engine checks C0 % 32 == C1 % 32 == 0)
    const int tpi = a.tiles_x * a.tiles_y;
    const size_t img_px = (size_t)a.Hin * a.Win;
    const _Float16* const src0h = reinterpret_cast<const _Float16*>(a.src0);
    const _Float16* const src1h = reinterpret_cast<const _Float16*>(a.src1 ? a.src1 : a.src0);
    const float* const sc1p = a.src1 ? a.sc1 : a.sc0;
    const float* const sh1p = a.src1 ? a.sh1 : a.sh0;
    if (w >= 4) __builtin_amdgcn_s_setprio(1);               // static issue priority for the younger half (kernels_f16x3_qp.h)

    // ---- staging units: patch pixel pp = 16 (8 it + w) + (lane & 7) + 8 (lane >> 5), k-group sg: LDS slot plane sg, pixel pp
    const int pp0 = 16 * w + (lane & 7) + 8 * (lane >> 5);
    const int lw0 = sg * kQ16Plane + pp0 * 16;               // unit it: + 2048 it
    struct Item { int k, c; };                               // tile number within the workgroup, chunk
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
    unsigned real_pf = 0;                                    // bit it: unit it of the prefetched item lies inside the image
    const _Float16 slope_h = (_Float16)a.slope;
    const unsigned slope2 = (unsigned)__builtin_bit_cast(unsigned short, slope_h) * 0x10001u;
    auto prefetch = [&](const Item& t) {                     // 5 buffer loads + 4 global loads, branch-free
        int nimg, ty0, tx0, tin;
        tile_origin(t.k, nimg, ty0, tx0, tin);
        const int cb0 = t.c * 32;
        const bool first = cb0 < a.C0;
        const int cb = first ? cb0 : cb0 - a.C0, C = first ? a.C0 : a.C1;
        const _Float16* base = (first ? src0h : src1h) + (size_t)nimg * img_px * C;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base), 0, (int)(img_px * C * 2), 0x00020000);
        unsigned m = 0;
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            const int pp = pp0 + 128 * it;
            const int py = pp / kPPW, px = pp - py * kPPW;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            const bool in = (py < kQRows) & ((unsigned)iy < (unsigned)a.Hin) & ((unsigned)ix < (unsigned)a.Win);
            const unsigned vo = in ? (unsigned)(((iy * a.Win + ix) * C + 8 * sg) * 2) : 0x80000000u;
            pv[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, cb * 2, 0);
            m |= in ? (1u << it) : 0u;
        }
        real_pf = m;
        const float* ps = (first ? a.sc0 : sc1p) + (size_t)nimg * C + cb + 8 * sg;
        const float* pt = (first ? a.sh0 : sh1p) + (size_t)nimg * C + cb + 8 * sg;
        nsa = *reinterpret_cast<const f32x4*>(ps); nsb = *reinterpret_cast<const f32x4*>(ps + 4);
        nta = *reinterpret_cast<const f32x4*>(pt); ntb = *reinterpret_cast<const f32x4*>(pt + 4);
    };
    auto convert = [&](int it, unsigned char* pb) {          // InstanceNorm + LeakyReLU in packed fp16; a padding pixel stores zeros
        uint4 x = norm_lrelu_8(uint4{pv[it][0], pv[it][1], pv[it][2], pv[it][3]}, nsa, nsb, nta, ntb, slope2);
        const bool real = (real_pf >> it) & 1u;
        x.x = real ? x.x : 0u; x.y = real ? x.y : 0u; x.z = real ? x.z : 0u; x.w = real ? x.w : 0u;
        if (it < 4 || pp0 + 128 * 4 < kQSlots) *reinterpret_cast<uint4*>(pb + lw0 + it * 2048) = x;      // (pixels 612 .. 639 of unit 4 do not exist)
    };
    auto weights_dma = [&](int ch, unsigned char* wb) {     // 36 pieces of 1 KiB; every wave issues exactly 5 (pieces 32..35 twice)
        const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * kQWts + lane * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_amdgcn_global_load_lds(wsrc + (w + 8 * k) * 1024, (lds_ptr)(wb + (w + 8 * k) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(wsrc + ((w & 3) + 32) * 1024, (lds_ptr)(wb + ((w & 3) + 32) * 1024), 16, 0, 0);
    };

    unsigned char* const wbuf0 = smem8 + 2 * kQ16Patch;
    // ---- fill the pipeline: item 0 staged synchronously (once per workgroup), item 1 requested
    Item cur{0, 0}, nx1{0, 0}, nx2{0, 0};
    prefetch(cur);
    weights_dma(0, wbuf0);
#pragma unroll
    for (int it = 0; it < MAXU; ++it) convert(it, smem8);
    advance(nx1);
    nx2 = nx1;
    prefetch(nx1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    advance(nx2);

    // ---- lane constants of the MFMA phase: X plane g, pixel (2 w + (pb >> 1) + dy, 16 (pb & 1) + dx + j); W [tap][g][column 16 cb + j]
    const int xA = g * kQ16Plane + ((2 * w) * kPPW + j) * 16;
    const int wA = g * 1024 + j * 16;

    f32x4 acc_t[4][4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) acc_t[cb][pb] = kZero4;

    const int nitems = ntl * nchunks;
    int pend = -1;                                           // statistics of a finished tile waiting for the item barrier: its entry in a.part
    for (int i = 0; i < nitems; ++i) {
        const int b = i & 1;
        const unsigned char* pA = smem8 + b * kQ16Patch + xA;
        const unsigned char* qA = wbuf0 + b * kQWts + wA;
        unsigned char* pb_next = smem8 + (b ^ 1) * kQ16Patch;
        unsigned char* wb_next = wbuf0 + (b ^ 1) * kQWts;

        f32x4 acc_c[4][4];
        // 9 k-steps (taps) x 4 groups of 4 MFMAs (one 16-column block each).  Registers as in conv3x3_f16x3_qp16: the X fragments of
        // tap t+1 are read during tap t (two sets of 4), the W fragments travel through a ring of four, read two groups ahead.
#define TS2D_XL(SET, PX) { _Pragma("unroll") for (int pb = 0; pb < 4; ++pb) \
            fx[SET][pb] = *reinterpret_cast<const half8*>((PX) + ((pb >> 1) * kPPW + 16 * (pb & 1)) * 16); }
#define TS2D_G(USE, FILL, WN, SET, CB, FIRST, EXTRA) { \
            wr[FILL] = *reinterpret_cast<const half8*>(WN); \
            EXTRA \
            _Pragma("unroll") for (int pb = 0; pb < 4; ++pb) \
                acc_c[CB][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[USE], fx[SET][pb], (FIRST) ? kZero4 : acc_c[CB][pb], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); }
#define TS2D_GL(USE, SET, CB, EXTRA) { \
            EXTRA \
            _Pragma("unroll") for (int pb = 0; pb < 4; ++pb) \
                acc_c[CB][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[USE], fx[SET][pb], acc_c[CB][pb], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); }
        half8 wr[4], fx[2][4];
        wr[0] = *reinterpret_cast<const half8*>(qA + 0);
        wr[1] = *reinterpret_cast<const half8*>(qA + 256);
        TS2D_XL(0, pA)
        // (memory operations of an item: every use of a loaded register first, THEN the weight DMA - kernels_f16x3_qp.h)
        TS2D_XL(1, pA + (0 * kPPW + 1) * 16)                                   // ---- tap 0
        TS2D_G(0, 2, qA + 0 * WTAP + 512, 0, 0, true, convert(0, pb_next);)
        TS2D_G(1, 3, qA + 0 * WTAP + 768, 0, 1, true, convert(1, pb_next);)
        TS2D_G(2, 0, qA + 1 * WTAP + 0, 0, 2, true, convert(2, pb_next);)
        TS2D_G(3, 1, qA + 1 * WTAP + 256, 0, 3, true, convert(3, pb_next);)
        TS2D_XL(0, pA + (0 * kPPW + 2) * 16)                                   // ---- tap 1
        TS2D_G(0, 2, qA + 1 * WTAP + 512, 1, 0, false, convert(4, pb_next);)
        TS2D_G(1, 3, qA + 1 * WTAP + 768, 1, 1, false, prefetch(nx2);)
        TS2D_G(2, 0, qA + 2 * WTAP + 0, 1, 2, false, weights_dma(nx1.c, wb_next);)
        TS2D_G(3, 1, qA + 2 * WTAP + 256, 1, 3, false, )
        TS2D_XL(1, pA + (1 * kPPW + 0) * 16)                                   // ---- tap 2
        TS2D_G(0, 2, qA + 2 * WTAP + 512, 0, 0, false, )
        TS2D_G(1, 3, qA + 2 * WTAP + 768, 0, 1, false, )
        TS2D_G(2, 0, qA + 3 * WTAP + 0, 0, 2, false, )
        TS2D_G(3, 1, qA + 3 * WTAP + 256, 0, 3, false, )
        TS2D_XL(0, pA + (1 * kPPW + 1) * 16)                                   // ---- tap 3
        TS2D_G(0, 2, qA + 3 * WTAP + 512, 1, 0, false, )
        TS2D_G(1, 3, qA + 3 * WTAP + 768, 1, 1, false, )
        TS2D_G(2, 0, qA + 4 * WTAP + 0, 1, 2, false, )
        TS2D_G(3, 1, qA + 4 * WTAP + 256, 1, 3, false, )
        TS2D_XL(1, pA + (1 * kPPW + 2) * 16)                                   // ---- tap 4
        TS2D_G(0, 2, qA + 4 * WTAP + 512, 0, 0, false, )
        TS2D_G(1, 3, qA + 4 * WTAP + 768, 0, 1, false, )
        TS2D_G(2, 0, qA + 5 * WTAP + 0, 0, 2, false, )
        TS2D_G(3, 1, qA + 5 * WTAP + 256, 0, 3, false, )
        TS2D_XL(0, pA + (2 * kPPW + 0) * 16)                                   // ---- tap 5
        TS2D_G(0, 2, qA + 5 * WTAP + 512, 1, 0, false, )
        TS2D_G(1, 3, qA + 5 * WTAP + 768, 1, 1, false, )
        TS2D_G(2, 0, qA + 6 * WTAP + 0, 1, 2, false, )
        TS2D_G(3, 1, qA + 6 * WTAP + 256, 1, 3, false, )
        TS2D_XL(1, pA + (2 * kPPW + 1) * 16)                                   // ---- tap 6
        TS2D_G(0, 2, qA + 6 * WTAP + 512, 0, 0, false, )
        TS2D_G(1, 3, qA + 6 * WTAP + 768, 0, 1, false, )
        TS2D_G(2, 0, qA + 7 * WTAP + 0, 0, 2, false, )
        TS2D_G(3, 1, qA + 7 * WTAP + 256, 0, 3, false, )
        TS2D_XL(0, pA + (2 * kPPW + 2) * 16)                                   // ---- tap 7
        TS2D_G(0, 2, qA + 7 * WTAP + 512, 1, 0, false, )
        TS2D_G(1, 3, qA + 7 * WTAP + 768, 1, 1, false, )
        TS2D_G(2, 0, qA + 8 * WTAP + 0, 1, 2, false, )
        TS2D_G(3, 1, qA + 8 * WTAP + 256, 1, 3, false, )
        // ---- tap 8
        TS2D_G(0, 2, qA + 8 * WTAP + 512, 0, 0, false, )
        TS2D_G(1, 3, qA + 8 * WTAP + 768, 0, 1, false, )
        TS2D_GL(2, 0, 2, )
        TS2D_GL(3, 0, 3, )
#undef TS2D_GL
#undef TS2D_G
#undef TS2D_XL
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) acc_t[cb][pb] += acc_c[cb][pb];

        if (cur.c == nchunks - 1) {                          // (uniform) the tile is complete: bias, store, statistics; the next items' staging is in flight
            int nimg, ty0, tx0, tin;
            tile_origin(cur.k, nimg, ty0, tx0, tin);
            const size_t img_el = (size_t)a.Ht * a.Wt * a.Cout;
            const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<_Float16*>(a.dst) + (size_t)nimg * img_el, 0, (int)(img_el * 2), 0x00020000);
            const float oscale = *a.oscale;
            f32x4 bv[4];                                     // (every bias value before the first store)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) bv[cb] = *reinterpret_cast<const f32x4*>(a.bias + n0col + 16 * cb + 4 * g);
            // lane = pixel j of block pb, channels n0col + 16 cb + 4 g .. + 3: one 8-byte store per 16x16 block
            const unsigned vst = (unsigned)((((ty0 + 2 * w) * a.Wt + tx0 + j) * a.Cout + n0col + 4 * g) * 2);
            float* red = reinterpret_cast<float*>(smem8 + kQ16Red);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                // shifted statistics (kernels.h) of what is stored: pivot = the stored value of pixel 0 of the wave's first block, per channel
                f32x4 kv, ss = kZero4, qq = kZero4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = (float)(_Float16)__builtin_fmaf(acc_t[cb][0][e], oscale, bv[cb][e]);
                    kv[e] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v0), 0x150, 0xF, 0xF, true));      // row_newbcast:0
                }
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    half4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hv[e] = (_Float16)__builtin_fmaf(acc_t[cb][pb][e], oscale, bv[cb][e]);
                        const float d = (float)hv[e] - kv[e];
                        ss[e] += d; qq[e] = __builtin_fmaf(d, d, qq[e]);
                    }
                    const unsigned off = vst + (unsigned)((((pb >> 1) * a.Wt + 16 * (pb & 1)) * a.Cout + 16 * cb) * 2);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), rsd, off, 0, 0);
                    acc_t[cb][pb] = kZero4;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {                // sum over the 16 pixels of the lane row (DPP row rotations)
                    float s = ss[e], q = qq[e];
#define TS2D_ROR_ADD(X, N) X += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, X), 0x120 + N, 0xF, 0xF, true))
                    TS2D_ROR_ADD(s, 8); TS2D_ROR_ADD(q, 8); TS2D_ROR_ADD(s, 4); TS2D_ROR_ADD(q, 4);
                    TS2D_ROR_ADD(s, 2); TS2D_ROR_ADD(q, 2); TS2D_ROR_ADD(s, 1); TS2D_ROR_ADD(q, 1);
#undef TS2D_ROR_ADD
                    if (j == 0) stat_wave_put(red, w * BN + 16 * cb + 4 * g + e, s, q, kv[e], 64.f);
                }
            }
            pend = (nimg * tpi + tin) * a.Cout + n0col;      // the cross-wave merge waits for the item's own barrier below (kernels_f16x3_qp.h)
        }
        advance(cur); advance(nx1); advance(nx2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (pend >= 0) {                                     // (uniform)
            if (tid < BN) stat_tile_store(reinterpret_cast<const float*>(smem8 + kQ16Red), 8, BN, tid, a.part + ((size_t)pend + tid) * 4);
            pend = -1;
        }
    }
}

}  // namespace ts2d
