// conv3x3_f16x3_qp16: the persistent pipeline of conv3x3_f16x3_qp (kernels_f16x3_qp.h: one 512-thread workgroup per CU, patch and
// weights double-buffered, weights by LDS-DMA, one barrier per (tile, chunk) item) on v_mfma_f32_16x16x32_f16.
//
// Why (profiles/r03_mfma_shape_probe.txt): on this chip the 16x16x32 shape holds ~1.99 GHz under dense load where
// 32x32x16 holds ~1.72 - equal cycles per FLOP, 1.09-1.14x the FLOP/s (guide: "DVFS give-back" item 7).
//
// How a 16-channel chunk feeds a K = 32 instruction WITHOUT 32-channel chunks (a 32-channel patch + weight block of this tile is
// 153 KB: it fits LDS once, never twice): the K dimension is packed with the two PARTS of the split instead of 32 channels.
//   k-group g = lane >> 4 of the X (patch) operand  = LDS plane g = [x_hi ch 0-7 | x_hi ch 8-15 | x_lo ch 0-7 | x_lo ch 8-15]
//   k-group g of the W operand                      = [w_hi ch 0-7 | w_hi ch 8-15 | w_hi ch 0-7 | w_hi ch 8-15]
// so ONE instruction per tap computes w_hi * x_hi + w_hi * x_lo; the third product w_lo * x_hi packs TWO TAPS into K
// (k-groups 0,1 = tap t, k-groups 2,3 = tap t + 1: a per-lane address offset on both operands), the ninth tap alone with the upper
// k-groups of W reading a zeroed LDS block.  Per chunk and wave: 9 + 5 = 14 k-steps x 16 blocks = 224 MFMAs of 16 cycles (3 584
// cycles; the 32x32x16 form: 108 x 32 = 3 456) and 112 ds_read_b128 (72).  Nothing else changes: same patch planes, same weight image in HBM
// (conv3x3_f16x3_qp's: [chunk][column tile][tap][hi,lo][h][column][8 halves]), same staging, same DMA.
//
// The product is issued TRANSPOSED, D[cout][pixel] = W^T X (as conv3x3_res32): a lane holds 4 consecutive output channels of one
// pixel, the epilogue is 16 stores of 16 bytes per lane and tile instead of 64 of 4 bytes, the tile statistics reduce over the 16
// pixels of a lane row by DPP.  LDS: planes padded to 624 slots (a multiple of 256 bytes: a ds_read_b128 lane group - lanes
// {0-3, 12-15} of one k-group and {4-11} of the next - then covers 16 distinct 16-byte slots whatever the tap offset):
// 2 x 39 936 (patch) + 2 x 36 864 (weights) + 1 024 (zeros) + 8 192 (statistics) = 162 816 bytes.
// Summation order differs from the 32x32x16 kernels (three products interleaved differently): values agree to fp32 rounding,
// not bit for bit.
#pragma once
#include "kernels_f16x3_qp.h"

namespace ts2d {

constexpr int kQ16Slots = 624, kQ16Plane = kQ16Slots * 16, kQ16Patch = 4 * kQ16Plane;
constexpr int kQ16Zero = 2 * kQ16Patch + 2 * kQWts, kQ16Red = kQ16Zero + 1024, kQ16Lds = kQ16Red + 8192;
constexpr f32x4 kZero4 = {0.f, 0.f, 0.f, 0.f};

template <int VAR = 0>      // VAR: experiment switches (TS2D_Q16V) - bit 0: static issue priority for waves 4-7 (guide, "Two waves per SIMD" item 4)
__global__ __launch_bounds__(kQThreads, 1) void conv3x3_f16x3_qp16(const ConvArgs a) {
    constexpr int BN = 64, MAXU = 3, WTAP = 4 * BN * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int j = lane & 15, g = lane >> 4;                  // MFMA lane roles: row / column index, k-group
    const int octi = (lane >> 3) & 1, oct = octi * 8;        // staging roles (as conv3x3_f16x3_qp)

    // ---- this workgroup's tiles: virtual block v = blockIdx.x + k * gridDim.x -> (xcd, column tile) fixed, pixel tile mtile0 + k * mstep
    const int xcd = blockIdx.x & 7, q80 = blockIdx.x >> 3;
    const int ctile = q80 & (a.n_ctiles - 1), n0col = ctile * BN;
    const int mtile0 = (q80 >> a.lg_nct) * 8 + xcd, mstep = ((int)(gridDim.x >> 3) >> a.lg_nct) * 8;
    if (mtile0 >= a.n_mtiles) return;
    const int ntl = (a.n_mtiles - 1 - mtile0) / mstep + 1;                 // tiles of this workgroup
    const int nchunks = (a.C0 + a.C1) / 16;
    const int tpi = a.tiles_x * a.tiles_y;
    const size_t img_px = (size_t)a.Hin * a.Win;
    const float* const src1p = a.src1 ? a.src1 : a.src0;
    const float* const sc1p = a.src1 ? a.sc1 : a.sc0;
    const float* const sh1p = a.src1 ? a.sh1 : a.sh0;

    if (tid < 64) *reinterpret_cast<uint4*>(smem8 + kQ16Zero + tid * 16) = uint4{0u, 0u, 0u, 0u};      // W operand of the missing tenth tap

    // ---- staging units: patch pixel pp = 32 (8 it + w) + (lane & 7) + 8 (lane >> 4) -> (py, px), tile-independent
    //      (register budget: this kernel sits at the 256-register limit of two waves per SIMD - the three (py, px) pairs share ONE
    //       register: px in 6 bits, py in 3 / 4 / 5 bits (unit it covers pixels < 256 (it + 1)) at bits 0 / 9 / 19; py >= 18 = the unit does
    //       not exist; the LDS write address of unit it is lw0 + 4096 it)
    unsigned upk = 0;
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int pp = 32 * (8 * it + w) + (lane & 7) + 8 * (lane >> 4);
        const int py = pp / kPPW, px = pp - py * kPPW;
        upk |= (unsigned)((py << 6) | px) << (it == 0 ? 0 : (it == 1 ? 9 : 19));
    }
    auto unit_py = [&](int it) { return (int)((upk >> (it == 0 ? 6 : (it == 1 ? 15 : 25))) & (it == 0 ? 7u : (it == 1 ? 15u : 31u))); };
    auto unit_px = [&](int it) { return (int)((upk >> (it == 0 ? 0 : (it == 1 ? 9 : 19))) & 63u); };
    const int lw0 = octi * kQ16Plane + (32 * w + (lane & 7) + 8 * (lane >> 4)) * 16;
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

    u32x4 pv[MAXU][2];
    f32x4 nsa, nsb, nta, ntb;
    unsigned real_pf = 0;                                    // bit it: unit it of the prefetched item lies inside the image (read by convert, which
                                                             // runs before the next prefetch overwrites it)
    auto prefetch = [&](const Item& t) {                     // 6 buffer loads + 4 global loads, branch-free
        int nimg, ty0, tx0, tin;
        tile_origin(t.k, nimg, ty0, tx0, tin);
        const int cb0 = t.c * 16;
        const bool first = cb0 < a.C0;
        const int cb = first ? cb0 : cb0 - a.C0, C = first ? a.C0 : a.C1;
        const float* base = (first ? a.src0 : src1p) + (size_t)nimg * img_px * C;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(img_px * C * 4), 0x00020000);
        unsigned m = 0;
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            const int py = unit_py(it), px = unit_px(it);
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            const bool in = (py < kQRows) & ((unsigned)iy < (unsigned)a.Hin) & ((unsigned)ix < (unsigned)a.Win);      // (bitwise: no short-circuit branches)
            const unsigned vo = in ? (unsigned)(((iy * a.Win + ix) * C + oct) * 4) : 0x80000000u;
            pv[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, cb * 4, 0);
            pv[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 16, cb * 4, 0);
            m |= in ? (1u << it) : 0u;
        }
        real_pf = m;
        const float* ps = (first ? a.sc0 : sc1p) + (size_t)nimg * C + cb + oct;
        const float* pt = (first ? a.sh0 : sh1p) + (size_t)nimg * C + cb + oct;
        nsa = *reinterpret_cast<const f32x4*>(ps); nsb = *reinterpret_cast<const f32x4*>(ps + 4);
        nta = *reinterpret_cast<const f32x4*>(pt); ntb = *reinterpret_cast<const f32x4*>(pt + 4);
    };
    // conversion of a staging unit in two halves (each rides in its own 4-MFMA group of the loop): A = InstanceNorm + LeakyReLU,
    // B = hi / lo split + LDS write; branch-free arithmetic (a padding pixel stores zeros)
    f32x4 cva, cvb;
    auto convert_a = [&](int it) {
        f32x4 va = __builtin_bit_cast(f32x4, pv[it][0]), vb = __builtin_bit_cast(f32x4, pv[it][1]);
        va = va * nsa + nta; vb = vb * nsb + ntb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            va[e] = fmaxf(va[e], va[e] * a.slope);
            vb[e] = fmaxf(vb[e], vb[e] * a.slope);
        }
        cva = va; cvb = vb;
    };
    auto convert_b = [&](int it, unsigned char* pb) {
        uint4 hi, lo;
        split_hi_lo_8(cva, cvb, hi, lo);
        const bool real = (real_pf >> it) & 1u;
        hi.x = real ? hi.x : 0u; hi.y = real ? hi.y : 0u; hi.z = real ? hi.z : 0u; hi.w = real ? hi.w : 0u;
        lo.x = real ? lo.x : 0u; lo.y = real ? lo.y : 0u; lo.z = real ? lo.z : 0u; lo.w = real ? lo.w : 0u;
        if (it < 2 || unit_py(2) < kQRows) {
            *reinterpret_cast<uint4*>(pb + lw0 + it * 4096) = hi;
            *reinterpret_cast<uint4*>(pb + lw0 + it * 4096 + 2 * kQ16Plane) = lo;
        }
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
    for (int it = 0; it < MAXU; ++it) { convert_a(it); convert_b(it, smem8); }
    advance(nx1);
    nx2 = nx1;
    prefetch(nx1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    advance(nx2);

    // ---- lane constants of the MFMA phase (byte offsets inside a patch / weight buffer)
    //   X, products 1+2 (tap t):      plane g,           pixel (2 w + (pb >> 1) + dy, 16 (pb & 1) + dx + j)
    //   X, product 3 (taps t, t + 1): plane g & 1 (hi),  tap t + (g >> 1): + 16 bytes for the pairs (0,1) (4,5) (6,7), + 512 for (2,3)
    //   W, products 1+2 (tap t):      [t][hi][h = g & 1][column 16 cb + j]
    //   W, product 3:                 [t + (g >> 1)][lo][h = g & 1][column]; ninth tap: k-groups 2, 3 read the zero block
    //   (two lane constants; everything for k-groups 2, 3 is a select on lane >= 32)
    const int xA = g * kQ16Plane + ((2 * w) * kPPW + j) * 16;
    const int wA = (g & 1) * 1024 + j * 16;
    const bool up = lane >= 32;

    f32x4 acc_t[4][4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) acc_t[cb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (VAR & 1) { if (w >= 4) __builtin_amdgcn_s_setprio(1); }
    const int nitems = ntl * nchunks;
    int pend = -1;                                           // statistics of a finished tile waiting for the item barrier: its entry in a.part
    for (int i = 0; i < nitems; ++i) {
        const int b = i & 1;
        const unsigned char* pA = smem8 + b * kQ16Patch + xA;
        const unsigned char* pH = up ? pA - 2 * kQ16Plane : pA;
        const unsigned char* pH16 = up ? pA - 2 * kQ16Plane + 16 : pA;
        const unsigned char* pH512 = up ? pA - 2 * kQ16Plane + 512 : pA;
        const unsigned char* qA = wbuf0 + b * kQWts + wA;
        const unsigned char* qL = up ? qA + 2048 + WTAP : qA + 2048;
        const unsigned char* qL8 = up ? smem8 + kQ16Zero + (wA & 1023) : qA + 2048 + 8 * WTAP;      // ninth tap alone: zeros for k-groups 2, 3
        unsigned char* pb_next = smem8 + (b ^ 1) * kQ16Patch;
        unsigned char* wb_next = wbuf0 + (b ^ 1) * kQWts;

        f32x4 acc_c[4][4];
        // k-step S = 0 .. 13 in issue order: C0 C1 T01 C2 C3 T23 C4 C5 T45 C6 C7 T67 C8 T8 (C = products 1+2 of a tap, T = product 3 of a
        // pair).  Registers: the X fragments of step S+1 are read during step S (two sets of 4), the W fragments travel through a ring
        // of four, read two 4-MFMA groups (128 cycles) ahead of their use (the slot being filled was last read two groups ago: no
        // read-after-MFMA hazard holds the load back) - 48 fragment registers instead of 64 for two full sets:
        // with 128 accumulator registers and the staging data in flight, two full sets spilled lane constants into the loop.
        // The 14 x 4 groups below are written out (a `#pragma unroll` loop over the k-steps was NOT unrolled by hipcc: its fragment
        // arrays went through runtime-indexed selects and 96 registers spilled).  TS2D_G(ring slot to use, ring slot to fill, W pointer of
        // the group two ahead, X set, column block, first k-step, extra work): 1 W read two groups ahead, 4 MFMAs.
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
        TS2D_XL(1, pA + 0*kPPW*16 + 16)                                   // ---- k-step 0 = C0
        TS2D_G(0, 2, qA + 512, 0, 0, true, convert_a(0);)
        TS2D_G(1, 3, qA + 768, 0, 1, true, convert_b(0, pb_next);)
        TS2D_G(2, 0, qA + WTAP + 0, 0, 2, true, convert_a(1);)
        TS2D_G(3, 1, qA + WTAP + 256, 0, 3, true, convert_b(1, pb_next);)
        TS2D_XL(0, pH16 + 0*kPPW*16 + 0)                                   // ---- k-step 1 = C1
        TS2D_G(0, 2, qA + WTAP + 512, 1, 0, false, convert_a(2);)
        TS2D_G(1, 3, qA + WTAP + 768, 1, 1, false, convert_b(2, pb_next);)
        TS2D_G(2, 0, qL + 0, 1, 2, false, prefetch(nx2);)
        TS2D_G(3, 1, qL + 256, 1, 3, false, weights_dma(nx1.c, wb_next);)
        TS2D_XL(1, pA + 0*kPPW*16 + 32)                                   // ---- k-step 2 = T01
        TS2D_G(0, 2, qL + 512, 0, 0, false, )
        TS2D_G(1, 3, qL + 768, 0, 1, false, )
        TS2D_G(2, 0, qA + 2 * WTAP + 0, 0, 2, false, )
        TS2D_G(3, 1, qA + 2 * WTAP + 256, 0, 3, false, )
        TS2D_XL(0, pA + 1*kPPW*16 + 0)                                   // ---- k-step 3 = C2
        TS2D_G(0, 2, qA + 2 * WTAP + 512, 1, 0, false, )
        TS2D_G(1, 3, qA + 2 * WTAP + 768, 1, 1, false, )
        TS2D_G(2, 0, qA + 3 * WTAP + 0, 1, 2, false, )
        TS2D_G(3, 1, qA + 3 * WTAP + 256, 1, 3, false, )
        TS2D_XL(1, pH512 + 0*kPPW*16 + 32)                                   // ---- k-step 4 = C3
        TS2D_G(0, 2, qA + 3 * WTAP + 512, 0, 0, false, )
        TS2D_G(1, 3, qA + 3 * WTAP + 768, 0, 1, false, )
        TS2D_G(2, 0, qL + 2 * WTAP + 0, 0, 2, false, )
        TS2D_G(3, 1, qL + 2 * WTAP + 256, 0, 3, false, )
        TS2D_XL(0, pA + 1*kPPW*16 + 16)                                   // ---- k-step 5 = T23
        TS2D_G(0, 2, qL + 2 * WTAP + 512, 1, 0, false, )
        TS2D_G(1, 3, qL + 2 * WTAP + 768, 1, 1, false, )
        TS2D_G(2, 0, qA + 4 * WTAP + 0, 1, 2, false, )
        TS2D_G(3, 1, qA + 4 * WTAP + 256, 1, 3, false, )
        TS2D_XL(1, pA + 1*kPPW*16 + 32)                                   // ---- k-step 6 = C4
        TS2D_G(0, 2, qA + 4 * WTAP + 512, 0, 0, false, )
        TS2D_G(1, 3, qA + 4 * WTAP + 768, 0, 1, false, )
        TS2D_G(2, 0, qA + 5 * WTAP + 0, 0, 2, false, )
        TS2D_G(3, 1, qA + 5 * WTAP + 256, 0, 3, false, )
        TS2D_XL(0, pH16 + 1*kPPW*16 + 16)                                   // ---- k-step 7 = C5
        TS2D_G(0, 2, qA + 5 * WTAP + 512, 1, 0, false, )
        TS2D_G(1, 3, qA + 5 * WTAP + 768, 1, 1, false, )
        TS2D_G(2, 0, qL + 4 * WTAP + 0, 1, 2, false, )
        TS2D_G(3, 1, qL + 4 * WTAP + 256, 1, 3, false, )
        TS2D_XL(1, pA + 2*kPPW*16 + 0)                                   // ---- k-step 8 = T45
        TS2D_G(0, 2, qL + 4 * WTAP + 512, 0, 0, false, )
        TS2D_G(1, 3, qL + 4 * WTAP + 768, 0, 1, false, )
        TS2D_G(2, 0, qA + 6 * WTAP + 0, 0, 2, false, )
        TS2D_G(3, 1, qA + 6 * WTAP + 256, 0, 3, false, )
        TS2D_XL(0, pA + 2*kPPW*16 + 16)                                   // ---- k-step 9 = C6
        TS2D_G(0, 2, qA + 6 * WTAP + 512, 1, 0, false, )
        TS2D_G(1, 3, qA + 6 * WTAP + 768, 1, 1, false, )
        TS2D_G(2, 0, qA + 7 * WTAP + 0, 1, 2, false, )
        TS2D_G(3, 1, qA + 7 * WTAP + 256, 1, 3, false, )
        TS2D_XL(1, pH16 + 2*kPPW*16 + 0)                                   // ---- k-step 10 = C7
        TS2D_G(0, 2, qA + 7 * WTAP + 512, 0, 0, false, )
        TS2D_G(1, 3, qA + 7 * WTAP + 768, 0, 1, false, )
        TS2D_G(2, 0, qL + 6 * WTAP + 0, 0, 2, false, )
        TS2D_G(3, 1, qL + 6 * WTAP + 256, 0, 3, false, )
        TS2D_XL(0, pA + 2*kPPW*16 + 32)                                   // ---- k-step 11 = T67
        TS2D_G(0, 2, qL + 6 * WTAP + 512, 1, 0, false, )
        TS2D_G(1, 3, qL + 6 * WTAP + 768, 1, 1, false, )
        TS2D_G(2, 0, qA + 8 * WTAP + 0, 1, 2, false, )
        TS2D_G(3, 1, qA + 8 * WTAP + 256, 1, 3, false, )
        TS2D_XL(1, pH + 2*kPPW*16 + 32)                                   // ---- k-step 12 = C8
        TS2D_G(0, 2, qA + 8 * WTAP + 512, 0, 0, false, )
        TS2D_G(1, 3, qA + 8 * WTAP + 768, 0, 1, false, )
        TS2D_G(2, 0, qL8 + 0, 0, 2, false, )
        TS2D_G(3, 1, qL8 + 256, 0, 3, false, )
        // ---- k-step 13 = T8
        TS2D_G(0, 2, qL8 + 512, 1, 0, false, )
        TS2D_G(1, 3, qL8 + 768, 1, 1, false, )
        TS2D_GL(2, 1, 2, )
        TS2D_GL(3, 1, 3, )
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
            const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(a.dst) + (size_t)nimg * img_el, 0, (int)(img_el * 4), 0x00020000);
            const float oscale = *a.oscale;
            f32x4 bv[4];                                     // (every bias value before the first store)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) bv[cb] = *reinterpret_cast<const f32x4*>(a.bias + n0col + 16 * cb + 4 * g);
            // lane = pixel j of block pb, channels n0col + 16 cb + 4 g .. + 3: one 16-byte store per 16x16 block
            const unsigned vst = (unsigned)((((ty0 + 2 * w) * a.Wt + tx0 + j) * a.Cout + n0col + 4 * g) * 4);
            float* red = reinterpret_cast<float*>(smem8 + kQ16Red);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                f32x4 ov[4];
                // shifted statistics (kernels.h): pivot = the stored value of pixel 0 of the wave's first block, per channel
                f32x4 kv, ss = kZero4, qq = kZero4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = __builtin_fmaf(acc_t[cb][0][e], oscale, bv[cb][e]);
                    kv[e] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v0), 0x150, 0xF, 0xF, true));      // row_newbcast:0
                }
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = __builtin_fmaf(acc_t[cb][pb][e], oscale, bv[cb][e]);
                        const float d = v[e] - kv[e];
                        ss[e] += d; qq[e] = __builtin_fmaf(d, d, qq[e]);
                    }
                    // (gfx950 wide-store hazard, kernels_res32.h: the whole offset rides in the VGPR, the stored vectors stay live below)
                    const unsigned off = vst + (unsigned)((((pb >> 1) * a.Wt + 16 * (pb & 1)) * a.Cout + 16 * cb) * 4);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsd, off, 0, 0);
                    ov[pb] = v;
                    acc_t[cb][pb] = kZero4;
                }
                // sum over the 16 pixels of the lane row (DPP row rotations: every lane of the row ends up with the total)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = ss[e], q = qq[e];
#define TS2D_ROR_ADD(X, N) X += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, X), 0x120 + N, 0xF, 0xF, true))
                    TS2D_ROR_ADD(s, 8); TS2D_ROR_ADD(q, 8); TS2D_ROR_ADD(s, 4); TS2D_ROR_ADD(q, 4);
                    TS2D_ROR_ADD(s, 2); TS2D_ROR_ADD(q, 2); TS2D_ROR_ADD(s, 1); TS2D_ROR_ADD(q, 1);
#undef TS2D_ROR_ADD
                    if (j == 0) stat_wave_put(red, w * BN + 16 * cb + 4 * g + e, s, q, kv[e], 64.f);
                }
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) asm volatile("" :: "v"(ov[pb]));      // store data registers untouched up to here (dozens of VALU behind the stores)
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
