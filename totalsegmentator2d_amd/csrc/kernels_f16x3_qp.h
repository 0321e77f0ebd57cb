// conv3x3_f16x3_qp: the stride-1 3x3 split kernel as ONE persistent 512-thread workgroup per CU with everything double-buffered.
//
// LDS layout ("k-group major planes", round 2): patch plane[part hi,lo][h][pixel] and weights plane[tap][part][h][column] of 16-byte
// slots (8 channels): the 32 lanes of a lane half read 32 CONSECUTIVE slots for a 32x32x16 fragment (conflict-free ds_read_b128 at
// any alignment), 8 consecutive lanes of the staging phase write 8 consecutive slots, every fragment address is lane base +
// immediate, and the weight block of a (chunk, column tile) is stored in HBM in LDS order so that it travels L2 -> LDS by
// global_load_lds (no registers, no ds_write).  Tile = 16 x 32 pixels x 64 channels, 8 waves (wave w = rows 2w, 2w + 1).
// Pipeline: ONE stream of (tile, chunk) items per workgroup - MFMAs of item i, conversion + weight DMA of item i+1, raw loads of
// item i+2 - one raw s_barrier per item; a tile's epilogue overlaps the staging of the next tile's first chunks.  Same tap order,
// chunking and per-chunk accumulators as conv3x3_f16x3_one: conv outputs bit-identical to it.
// (Round-2 predecessors, removed in round 3 - numbers in DESIGN.md section 4: conv3x3_f16x3_p = the plane layout on 8 x 32 tiles with
//  256-thread workgroups, same speed as the 80-byte record layout; conv3x3_f16x3_q = this pipeline filled and drained per tile:
//  512 -> 512 block 0.80 -> 0.73 ms, but slower than _p below 8 chunks per tile.)
//
// A workgroup (grid = 8 * n_ctiles * m workgroups, one per CU) keeps its XCD lane and its column tile and walks the pixel tiles
// v = blockIdx.x + k * gridDim.x of the XCD-aware map of the other kernels.  Three pipeline stages per iteration: MFMAs of item i,
// conversion + weight DMA of item i+1, raw patch / scale / shift loads of item i+2; each stage carries its own (tile, chunk)
// counter; the in-image test of a staging unit is recomputed from the tile origin (two compares per unit) instead of being held
// in registers per tile.  LDS: 152064 bytes + 8192 for the statistics exchange (the patch buffers stay busy).
// Measured (gpurun r2 qp1/qp2): 64 -> 64 at level 1 0.93 (conv3x3_f16x3_p) -> 0.84-0.89 ms, 128 -> 128 0.83 -> 0.78, 256 / 512 channels
// unchanged (0.76 / 0.74): the gain is the per-tile fill and drain, which weighs less the more chunks a tile has.
//
// Round 4: the raw patch loads.  In-kernel stamps (profiles/r04_experiments.txt): wave 0 spent 2.3 k of an item's ~9.5 k cycles at the
// ISSUE of the six patch loads of tap 2.  Each of those wave instructions touched 32 different 128-byte lines (32 pixels x 2 lanes x
// 16 bytes); the address path retires about a line per cycle, and all eight waves queued their loads in one burst.  Now a staging
// unit is ONE 16-byte quarter of a pixel's 64-byte chunk slice - four lanes per pixel, a wave instruction covers 16 pixels = 16
// lines - five units per thread, each converted at its own tap (taps 0-4) and re-requested for the item after next right behind
// its conversion; the weight DMA moves to tap 5 (every use of a loaded register still precedes it).  Worth 2-5 % at levels 1-2
// (64 / 128 channels), nothing at 256 / 512: wave 0's stall at the loads was covered by its SIMD partner.  Measured NOT to help on top
// (profiles/r04_experiments.txt): a third fewer scalar / address instructions per item (request kept across items, padding selects
// on border tiles only) - the loop is not bound by its non-MFMA instruction count either.
#pragma once
#include "kernels_f16x3_one.h"

namespace ts2d {

constexpr int kPPW = 34;                                     // patch row: 32 pixels + halo, one 16-byte slot per pixel and plane
constexpr int kQThreads = 512, kQRows = 18, kQSlots = kQRows * kPPW, kQPlane = kQSlots * 16, kQPatch = 4 * kQPlane;
constexpr int kQWts = 9 * 4 * 64 * 16, kQLds = 2 * kQPatch + 2 * kQWts;      // 2 x 39168 (patch) + 2 x 36864 (weights) = 152064 bytes
constexpr int kQpRed = kQLds, kQpLds = kQLds + 8192;          // + the statistics exchange [8 waves][64 columns] x (S, Q, K, n)

__global__ TS2D_PACKED_F32 __launch_bounds__(kQThreads, 1) void conv3x3_f16x3_qp(const ConvArgs a) {
    constexpr int BN = 64, NT = 2, MAXU = 5, WTAP = 4 * BN * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
    const int r = lane & 31, h = lane >> 5;
    const int sub = tid & 3;                                 // this thread's quarter of a chunk slice: channels 4 sub .. 4 sub + 3

    // ---- this workgroup's tiles: virtual block v = blockIdx.x + k * gridDim.x -> (xcd, column tile) fixed, pixel tile mtile0 + k * mstep
    const int xcd = blockIdx.x & 7, q80 = blockIdx.x >> 3;
    const int qm0 = q80 / a.n_ctiles;                       // (any tile count: round 5; the grid is a multiple of 8 * n_ctiles)
    const int ctile = q80 - qm0 * a.n_ctiles, n0col = ctile * BN;
    const int mtile0 = qm0 * 8 + xcd, mstep = ((int)(gridDim.x >> 3) / a.n_ctiles) * 8;
    if (mtile0 >= a.n_mtiles) return;
    const int ntl = (a.n_mtiles - 1 - mtile0) / mstep + 1;                 // tiles of this workgroup
    const int nchunks = (a.C0 + a.C1) / 16;
    const int tpi = a.tiles_x * a.tiles_y;
    const size_t img_px = (size_t)a.Hin * a.Win;
    const float* const src1p = a.src1 ? a.src1 : a.src0;
    const float* const sc1p = a.src1 ? a.sc1 : a.sc0;
    const float* const sh1p = a.src1 ? a.sh1 : a.sh0;

    // ---- staging units: unit it of thread tid = patch slot (tid >> 2) + 128 it -> (py, px), quarter sub; tile-independent.  (py, px) of the
    //      five units are packed into two registers: 5 + 6 bits per unit (py = 31: the unit does not exist - slots 612 .. 639 of unit 4).
    //      LDS write address of unit it: lw0 + 2048 it (plane h = sub >> 1, bytes 8 (sub & 1) .. of the slot); lo part: + 2 planes.
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
    const int lw0 = (sub >> 1) * kQPlane + (tid >> 2) * 16 + (sub & 1) * 8;
    struct Item { int k, c; };                               // tile number within the workgroup, chunk
    auto advance = [&](Item& t) {                            // next item of the stream; the last item repeats (loaded / staged, never used)
        int c = t.c + 1, k = t.k;
        if (c == nchunks) { c = 0; ++k; }
        if (k < ntl) { t.k = k; t.c = c; }
    };
    auto tile_origin = [&](int k, int& nimg, int& ty0, int& tx0, int& tin) {
        const int mtile = mtile0 + k * mstep;
        nimg = udiv_magic(mtile, a.mg_tpi); tin = mtile - nimg * tpi;
        const int tyi = udiv_magic(tin, a.mg_tx), txi = tin - tyi * a.tiles_x;
        ty0 = tyi << 4; tx0 = txi << 5;
    };

    u32x4 pv[MAXU];
    f32x4 nsa, nta;
    unsigned real_pf = 0;                                    // bit it: unit it of the item in the registers lies inside the image (set by its load,
                                                             // read by its conversion one item later)
    // a request = its wave-uniform part (source, descriptor, tile origin, chunk offset), computed once per item, + one load per unit
    struct Req { __amdgpu_buffer_rsrc_t rs; int ty0, tx0, C, cb; const float* ps; const float* pt; };
    auto request = [&](const Item& t) {
        int nimg, tin;
        Req q;
        tile_origin(t.k, nimg, q.ty0, q.tx0, tin);
        const int cb0 = t.c * 16;
        const bool first = cb0 < a.C0;
        q.cb = first ? cb0 : cb0 - a.C0; q.C = first ? a.C0 : a.C1;
        const float* base = (first ? a.src0 : src1p) + (size_t)nimg * img_px * q.C;
        q.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(img_px * q.C * 4), 0x00020000);
        q.ps = (first ? a.sc0 : sc1p) + (size_t)nimg * q.C + q.cb + 4 * sub;
        q.pt = (first ? a.sh0 : sh1p) + (size_t)nimg * q.C + q.cb + 4 * sub;
        return q;
    };
    auto load_unit = [&](const Req& q, int it) {             // one buffer load, branch-free
        const int py = unit_py(it), px = unit_px(it);
        const int iy = q.ty0 - 1 + py, ix = q.tx0 - 1 + px;
        const bool in = (py < kQRows) & ((unsigned)iy < (unsigned)a.Hin) & ((unsigned)ix < (unsigned)a.Win);      // (bitwise: no short-circuit branches)
        const unsigned vo = in ? (unsigned)(((iy * a.Win + ix) * q.C) * 4 + 16 * sub) : 0x80000000u;
        pv[it] = __builtin_amdgcn_raw_buffer_load_b128(q.rs, vo, q.cb * 4, 0);
        real_pf = (real_pf & ~(1u << it)) | (in ? (1u << it) : 0u);
    };
    auto load_norm = [&](const Req& q) {
        nsa = *reinterpret_cast<const f32x4*>(q.ps); nta = *reinterpret_cast<const f32x4*>(q.pt);
    };
    auto convert = [&](int it, unsigned char* pb) {          // branch-free arithmetic (a padding pixel stores zeros)
        f32x4 va = __builtin_bit_cast(f32x4, pv[it]);
        va = va * nsa + nta;
#pragma unroll
        for (int e = 0; e < 4; ++e) va[e] = fmaxf(va[e], va[e] * a.slope);
        uint2 hi, lo;
        split_hi_lo_4(va, hi, lo);
        const bool real = (real_pf >> it) & 1u;
        hi.x = real ? hi.x : 0u; hi.y = real ? hi.y : 0u;
        lo.x = real ? lo.x : 0u; lo.y = real ? lo.y : 0u;
        if (it < MAXU - 1 || unit_py(MAXU - 1) < kQRows) {
            *reinterpret_cast<uint2*>(pb + lw0 + it * 2048) = hi;
            *reinterpret_cast<uint2*>(pb + lw0 + it * 2048 + 2 * kQPlane) = lo;
        }
    };
    auto weights_dma = [&](int ch, unsigned char* wb) {     // 36 pieces of 1 KiB; every wave issues exactly 5 (pieces 32..35 twice)
        const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * kQWts + lane * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds(wsrc + (w + 8 * j) * 1024, (lds_ptr)(wb + (w + 8 * j) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(wsrc + ((w & 3) + 32) * 1024, (lds_ptr)(wb + ((w & 3) + 32) * 1024), 16, 0, 0);
    };

    unsigned char* const wbuf0 = smem8 + 2 * kQPatch;
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

    const int abase = h * kQPlane + ((2 * w) * kPPW + r) * 16;
    const int bbase = h * BN * 16 + r * 16;

    f32x16 acc_t[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_t[mt][nt][i] = 0.f;

    // static issue priority for the second-dispatched half of the workgroup (guide, "Two waves per SIMD" item 4: the younger wave of a
    // SIMD loses every VALU arbitration to the older one); measured (gpurun r3, scripts/gpu_ab.py): 2.5-5 % per layer, off: TS2D_DBG=512
    if (!(a.dbg & 512)) { if (w >= 4) __builtin_amdgcn_s_setprio(1); }
    const int nitems = ntl * nchunks;
    int pend = -1;                                           // statistics of a finished tile waiting for the item barrier: its entry in a.part
    for (int i = 0; i < nitems; ++i) {
        const int b = i & 1;
        const unsigned char* pa = smem8 + b * kQPatch + abase;
        const unsigned char* pw = wbuf0 + b * kQWts + bbase;
        unsigned char* pb_next = smem8 + (b ^ 1) * kQPatch;
        unsigned char* wb_next = wbuf0 + (b ^ 1) * kQWts;

        f32x16 acc_c[2][NT];
        half8 fa[2][2][2], fb[2][NT][2];                     // [buffer][tile][hi, lo]
#define TS2D_LOAD_FRAGS(BUF, TAP) { \
            constexpr int toff_ = (((TAP) / 3) * kPPW + ((TAP) % 3)) * 16; \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) { \
                fa[BUF][mt][0] = *reinterpret_cast<const half8*>(pa + mt * kPPW * 16 + toff_); \
                fa[BUF][mt][1] = *reinterpret_cast<const half8*>(pa + mt * kPPW * 16 + toff_ + 2 * kQPlane); } \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { \
                fb[BUF][nt][0] = *reinterpret_cast<const half8*>(pw + (TAP) * WTAP + nt * 512); \
                fb[BUF][nt][1] = *reinterpret_cast<const half8*>(pw + (TAP) * WTAP + nt * 512 + 2 * BN * 16); } }
#define TS2D_TAP(TAP, EXTRA) { constexpr int cur_ = (TAP) & 1; \
            if constexpr ((TAP) + 1 < 9) TS2D_LOAD_FRAGS(cur_ ^ 1, (TAP) + 1) \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur_][mt][1], fb[cur_][nt][0], (TAP) == 0 ? kZero16 : acc_c[mt][nt], 0, 0, 0); \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur_][mt][0], fb[cur_][nt][1], acc_c[mt][nt], 0, 0, 0); \
            EXTRA \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
                acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur_][mt][0], fb[cur_][nt][0], acc_c[mt][nt], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); }
        const Req rq = request(nx2);                         // the item after next: each unit re-requested right behind its conversion
        TS2D_LOAD_FRAGS(0, 0)
        // (memory operations of an item: every use of a loaded register first, THEN the weight DMA: hipcc answers ANY use of a loaded
        //  register with s_waitcnt vmcnt(0) while a global_load_lds is in flight, and falls back to vmcnt(0) at control-flow joins)
        TS2D_TAP(0, convert(0, pb_next); load_unit(rq, 0);)
        TS2D_TAP(1, convert(1, pb_next); load_unit(rq, 1);)
        TS2D_TAP(2, convert(2, pb_next); load_unit(rq, 2);)
        TS2D_TAP(3, convert(3, pb_next); load_unit(rq, 3);)
        TS2D_TAP(4, convert(4, pb_next); load_unit(rq, 4); load_norm(rq);)
        TS2D_TAP(5, weights_dma(nx1.c, wb_next);)
        TS2D_TAP(6, ) TS2D_TAP(7, ) TS2D_TAP(8, )
#undef TS2D_TAP
#undef TS2D_LOAD_FRAGS
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc_t[mt][nt] += acc_c[mt][nt];

        if (cur.c == nchunks - 1) {                          // (uniform) the tile is complete: bias, store, statistics; the next items' staging is in flight
            int nimg, ty0, tx0, tin;
            tile_origin(cur.k, nimg, ty0, tx0, tin);
            const size_t img_el = (size_t)a.Ht * a.Wt * a.Cout;
            const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(a.dst) + (size_t)nimg * img_el, 0, (int)(img_el * 4), 0x00020000);
            float st_s[NT], st_q[NT], st_k[NT], bvs[NT];
            const float oscale = *a.oscale;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bvs[nt] = a.bias[n0col + nt * 32 + r];       // (both before the first store)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = n0col + nt * 32 + r;
                const float kv = stat_pivot(__builtin_fmaf(acc_t[0][nt][0], oscale, bvs[nt]));      // shifted statistics (kernels.h)
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int oy = ty0 + 2 * w + mt, ox = tx0 + 4 * h;
                    const unsigned voff = (unsigned)(((oy * a.Wt + ox) * a.Cout + co) * 4);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rowoff = (e & 3) + 8 * (e >> 2);
                        const unsigned soff = (unsigned)(rowoff * a.Cout * 4);
                        const float v = __builtin_fmaf(acc_t[mt][nt][e], oscale, bvs[nt]);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsd, voff, soff, 0);
                        const float d = v - kv;
                        s += d; q = __builtin_fmaf(d, d, q);
                        acc_t[mt][nt][e] = 0.f;
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
            // the cross-wave merge waits for the item's own barrier below instead of one of its own (in-kernel stamps of round 2: the
            // extra barrier + its skew cost ~3 k cycles per tile); `red` is not written again before the next tile's epilogue
            pend = (nimg * tpi + tin) * a.Cout + n0col;
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
