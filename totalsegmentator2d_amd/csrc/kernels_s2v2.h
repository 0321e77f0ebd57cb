// Strided-conv downsample (Conv2d 3x3 stride 2 pad 1, enc{1..4}.c0 of the canonical net), second design.
//
// A stride-2 conv stages FOUR input pixels per output pixel, so per MFMA the InstanceNorm + LeakyReLU + hi/lo split of the patch
// costs four times what it costs the stride-1 kernel; round 1's kernel (256 threads, 64 output columns, 8-channel chunks) spent
// as long in staging as in MFMAs and read every 128-byte line of the input in 32-byte slices (2.8x the algorithmic HBM reads).
// Here:
//   * ONE 512-thread workgroup per CU (8 waves, 4 along the pixels x 2 along the output channels) produces a 256-pixel tile x
//     up to 128 output channels: the converted patch is shared by twice the MFMAs, the staging work is spread over 512 threads.
//   * chunks of 16 input channels (64-byte slices of the pixel records), one tap per MFMA k-step (9 k-steps, none wasted - the
//     8-channel x 2-tap packing issued 10 per 16 channels).
//   * "k-group major" LDS planes, plane[part][h][pixel slot] of 16 bytes (h = the 8-channel half a lane half feeds to the MFMA),
//     unpadded: the 32 lanes of a lane half read 32 consecutive slots - conflict-free ds_read_b128 at any alignment - and every
//     fragment address is lane base + immediate.  Patch columns are stored even columns first, then odd columns (66 slots per
//     row), so that the stride-2 pixel walk of a fragment is a walk over consecutive slots.
//   * the weight block of a (chunk, column tile) is stored in HBM in LDS order and copied linearly.
// Arithmetic as in kernels_f16x3.h (split: hi/lo fp16, 3 products, fresh accumulator per chunk; f16: one product).
#pragma once
#include "kernels_f16x3.h"
#include "kernels_h32.h"

namespace ts2d {

constexpr int kS2Threads = 512;
constexpr int kS2PW = 66, kS2Slots = 17 * kS2PW, kS2Plane = kS2Slots * 16;      // patch: 17 rows x (33 even + 33 odd columns)

// Round 3: PERSISTENT workgroups.  In-kernel stamps of round 2 (profiles/r02_phase_stamps.txt, enc1.c0): of ~27 000 cycles per
// two-chunk workgroup 8 800 were spent waiting for the first chunk's patch - one workgroup per CU (LDS), nothing else resident to
// hide it, 16 384 workgroups per launch.  Now a workgroup (one per CU, fixed XCD lane and column tile as conv3x3_f16x3_qp) walks
// ONE stream of (tile, chunk) items with the NEXT item's raw patch and scale / shift in flight in registers during the current
// item's MFMAs and epilogue - also across tile boundaries.  RESW: every chunk's weight block of the column tile stays RESIDENT in
// LDS for the workgroup's life when it fits beside the patch (enc1.c0: 2 x 36 KB), so the per-chunk weight staging disappears.
// Padding is decided per tile: a unit of patch row 0 / column 0 on a top / left border tile stores zeros (stride 2, pad 1 and even
// extents: the patch never leaves the image at the bottom or the right).
template <int BN, typename ST, int NP, bool RESW>
__global__ __launch_bounds__(kS2Threads, 2) void conv3x3s2_v2(const ConvArgs a) {
    constexpr int NPP = NP == 3 ? 2 : 1;                   // fp16 parts per value
    constexpr int NTW = BN / 64;                           // 32-column MFMA tiles per wave (a wave owns BN / 2 columns)
    constexpr int WTAP = NPP * 2 * BN * 16, WB = 9 * WTAP; // weight bytes per tap / per chunk
    constexpr int MAXU = 5;                                // staging units per thread: 5 x 256 slots >= 1122
    constexpr int NL = sizeof(ST) == 4 ? 2 : 1;            // 16-byte loads per unit (8 channels)
    constexpr int WIT = (WB / 16 + kS2Threads - 1) / kS2Threads;
    constexpr bool PFS = BN == 64;                         // scale / shift prefetched with the patch (BN = 128: no registers left - loaded at the item's start)
    constexpr int DEPTH = BN == 64 ? 2 : 1;                // items kept in flight in registers.  In-kernel stamps (enc1.c0, depth 1): the "MFMA" phases take
                                                           // 6-8 k cycles for 1.7 k of MFMAs - the wave sits at the ISSUE of the next item's loads behind the previous tile's
                                                           // stores: a CU moves ~10 B per cycle to / from HBM (guide), 175 KB per tile = 17 k of the tile's 26 k cycles, and
                                                           // with one item in flight that pipe idles during every conversion / epilogue.  Two items keep it fed.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    // ---- this workgroup's tiles: virtual block v = blockIdx.x + k * gridDim.x -> (xcd, column tile) fixed, pixel tile mtile0 + k * mstep
    const int xcd = blockIdx.x & 7, q80 = blockIdx.x >> 3;
    const int ctile = q80 & (a.n_ctiles - 1), n0col = ctile * BN;
    const int mtile0 = (q80 >> a.lg_nct) * 8 + xcd, mstep = ((int)(gridDim.x >> 3) >> a.lg_nct) * 8;
    if (mtile0 >= a.n_mtiles) return;
    const int ntl = (a.n_mtiles - 1 - mtile0) / mstep + 1;
    const int nchunks = a.C0 / 16;                         // the strided conv never reads a concat
    const int tpi = a.tiles_x * a.tiles_y;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 3, wn = w >> 2;
    const int r = lane & 31, h = lane >> 5;

    unsigned char* sA = smem8;                             // [part][h][slot] x 16 B
    unsigned char* sB = smem8 + NPP * 2 * kS2Plane;        // [chunk (RESW)][tap][part][h][column BN] x 16 B

    // ---- staging plan (tile-independent).  A wave instruction covers 32 slots x 2 channel octets: slot = 32 (8 it + w) + (lane & 7) +
    //      8 (lane >> 4), octet = (lane >> 3) & 1 - 8 consecutive lanes write 8 consecutive slots of one plane.  Slot q = patch row
    //      q / 66, then the 33 even columns, then the 33 odd ones (the 66th slot of a row is unused).
    const int oct = (lane >> 3) & 1;
    unsigned rel[MAXU];                                    // byte offset of the unit's 8 channels from the patch origin, 0x80000000 = no unit
    unsigned emask = 0;                                    // per unit: bit 0 = patch row 0, bit 1 = patch column 0 (the padding candidates)
    const int lw0 = oct * kS2Plane + (32 * w + (lane & 7) + 8 * (lane >> 4)) * 16;      // LDS write address of unit 0; unit it: + 4096 it
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int q = 32 * (8 * it + w) + (lane & 7) + 8 * (lane >> 4);
        const int py = q / kS2PW, rem = q - py * kS2PW;
        const int half = rem >= 33 ? 1 : 0, px = 2 * (rem - 33 * half) + half;
        const bool exists = q < kS2Slots && px <= 64;
        rel[it] = exists ? (unsigned)(((py * a.Win + px) * a.C0 + 8 * oct) * (int)sizeof(ST)) : 0x80000000u;
        if (exists) emask |= ((py == 0 ? 1u : 0u) | (px == 0 ? 2u : 0u)) << (2 * it);
        if (q < kS2Slots && !exists) {                     // the unused 66th slot of a row: zero once (never staged, read by no fragment)
            *reinterpret_cast<uint4*>(sA + lw0 + it * 4096) = uint4{0u, 0u, 0u, 0u};
            if (NPP == 2) *reinterpret_cast<uint4*>(sA + lw0 + it * 4096 + 2 * kS2Plane) = uint4{0u, 0u, 0u, 0u};
        }
    }
    if constexpr (RESW) {                                  // resident weights: chunks x 9 taps of this column tile, one linear copy each
        for (int c = 0; c < nchunks; ++c) {
            const uint4* wsrc = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(a.wph) + ((size_t)c * a.n_ctiles + ctile) * (9 * 2 * 2 * BN * 16));
#pragma unroll
            for (int k = 0; k < WIT; ++k) {
                const int sl = tid + k * kS2Threads;
                if (WB / 16 % kS2Threads == 0 || sl < WB / 16)
                    *reinterpret_cast<uint4*>(sB + c * WB + sl * 16) = wsrc[NPP == 2 ? sl : (sl / (2 * BN)) * (4 * BN) + sl % (2 * BN)];
            }
        }
    }

    struct Item { int k, c; };
    auto advance = [&](Item& t) {                          // next item of the stream; the last item repeats (loaded, never used)
        int c = t.c + 1, k = t.k;
        if (c == nchunks) { c = 0; ++k; }
        if (k < ntl) { t.k = k; t.c = c; }
    };
    auto tile_origin = [&](int k, int& nimg, int& tyi, int& txi, int& tin) {
        const int mtile = mtile0 + k * mstep;
        nimg = mtile >> a.lg_tpi; tin = mtile - nimg * tpi;
        tyi = tin >> a.lg_tx; txi = tin - tyi * a.tiles_x;
    };
    const size_t img_px = (size_t)a.Hin * a.Win;
    struct Stage { u32x4 pv[MAXU][NL]; f32x4 nsa, nsb, nta, ntb; };      // raw patch + scale / shift of one item in flight
    Stage sg0, sg1;
    const bool normed = a.sc0 != nullptr;
    auto load_norm = [&](Stage& S, int nimg, int c) {      // scale / shift of this thread's 8 channels (not normalised: loaded, not used)
        const float* ps = (normed ? a.sc0 : a.bias) + (normed ? (size_t)nimg * a.C0 + c * 16 + 8 * oct : 0);
        const float* pt = (normed ? a.sh0 : a.bias) + (normed ? (size_t)nimg * a.C0 + c * 16 + 8 * oct : 0);
        S.nsa = *reinterpret_cast<const f32x4*>(ps); S.nsb = *reinterpret_cast<const f32x4*>(ps + 4);
        S.nta = *reinterpret_cast<const f32x4*>(pt); S.ntb = *reinterpret_cast<const f32x4*>(pt + 4);
    };
    auto prefetch = [&](Stage& S, const Item& t) {         // 5 (x NL) buffer loads (+ 4 global loads), branch-free
        int nimg, tyi, txi, tin;
        tile_origin(t.k, nimg, tyi, txi, tin);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(reinterpret_cast<const ST*>(a.src0)) + (size_t)nimg * img_px * a.C0, 0,
                                                          (int)(img_px * a.C0 * sizeof(ST)), 0x00020000);
        // patch origin (2 ty0 - 1, 2 tx0 - 1) may lie one row / column outside the image: unsigned wrap-around is fine, the affected
        // units are padding (zeroed at conversion)
        const unsigned org = (unsigned)((((16 * tyi - 1) * a.Win + 64 * txi - 1) * a.C0) * (int)sizeof(ST));
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            const unsigned vo = rel[it] == 0x80000000u ? 0x80000000u : org + rel[it];
#pragma unroll
            for (int l = 0; l < NL; ++l) S.pv[it][l] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 16 * l, t.c * 16 * (int)sizeof(ST), 0);
        }
        if constexpr (PFS) load_norm(S, nimg, t.c);
    };

    Item cur{0, 0}, pf{0, 0};                              // item being computed; last item requested
    prefetch(sg0, cur);
    if constexpr (DEPTH == 2) { advance(pf); prefetch(sg1, pf); }

    // ---- lane constants of the MFMA phase: output pixel (2 wm + mt, r) reads patch row 2 (2 wm + mt) + dy, slot (dx & 1) 33 + r + (dx >> 1)
    const int abase = h * kS2Plane + ((4 * wm) * kS2PW + r) * 16;                  // + mt * 2 * 66 * 16 + part * 2 * Plane + tap offset
    const int bbase = NPP * 2 * kS2Plane + h * BN * 16 + (wn * (BN / 2) + r) * 16;  // + chunk * WB (RESW) + tap * WTAP + part * 2 * BN * 16 + nt * 512
    const _Float16 slope_h = (_Float16)a.slope;
    const unsigned slope2 = (unsigned)__builtin_bit_cast(unsigned short, slope_h) * 0x10001u;
    const f32x4 slope4 = f32x4{a.slope, a.slope, a.slope, a.slope};

    f32x16 acc_t[2][NTW];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_t[mt][nt][i] = 0.f;

    TS2D_PROF_DECL(a.prof);
    const int nitems = ntl * nchunks;
    const bool younger = w >= 4 && !(a.dbg & 512);         // static issue priority for waves 4-7 (kernels_f16x3_qp.h): restored after every MFMA phase
    auto item = [&](Stage& S) {                            // one (tile, chunk) item: conversion from S, next request into S, MFMAs, epilogue
        int nimg0, tyi, txi, tin;
        tile_origin(cur.k, nimg0, tyi, txi, tin);
        const int ty0 = tyi * 8, tx0 = txi * 32;
        lds_barrier();                                     // the previous item's MFMA reads of LDS (and its statistics exchange) are done
        TS2D_STAMP_AT(a.prof, 1)                           // (LDS-only barriers: a __syncthreads() here would wait for the previous tile's output stores)
        if constexpr (!PFS) load_norm(S, nimg0, cur.c);
        const f32x4 nsa = S.nsa, nsb = S.nsb, nta = S.nta, ntb = S.ntb;
        // ---- weights of this chunk (not resident): one linear block; loads issued first, written to LDS behind the patch conversion
        uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8;          // (named registers: an indexed array ends up in scratch)
        if constexpr (!RESW) {
            const uint4* wsrc = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(a.wph) + ((size_t)cur.c * a.n_ctiles + ctile) * (9 * 2 * 2 * BN * 16));
            // (f16 mode: the hi parts only - the first 2 BN slots of every 4 BN)
#define TS2D_WLOAD(K, R) { const int sl = tid + K * kS2Threads; if (K < WIT && (WB / 16 % kS2Threads == 0 || sl < WB / 16)) \
                R = wsrc[NPP == 2 ? sl : (sl / (2 * BN)) * (4 * BN) + sl % (2 * BN)]; }
            TS2D_WLOAD(0, w0) TS2D_WLOAD(1, w1) TS2D_WLOAD(2, w2) TS2D_WLOAD(3, w3) TS2D_WLOAD(4, w4)
            TS2D_WLOAD(5, w5) TS2D_WLOAD(6, w6) TS2D_WLOAD(7, w7) TS2D_WLOAD(8, w8)
#undef TS2D_WLOAD
        }
        // ---- patch: InstanceNorm + LeakyReLU on the fly, split into fp16 hi / lo; padding units store zeros
        const unsigned pad = (tyi == 0 ? 0x155u : 0u) | (txi == 0 ? 0x2AAu : 0u);      // (uniform) which emask bits mean "outside the image" here
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            if (rel[it] != 0x80000000u) {
                unsigned char* d = sA + lw0 + it * 4096;
                const bool real = ((emask & pad) >> (2 * it) & 3u) == 0u;
                if constexpr (sizeof(ST) == 4) {
                    f32x4 va = __builtin_bit_cast(f32x4, S.pv[it][0]), vb = __builtin_bit_cast(f32x4, S.pv[it][NL - 1]);
                    if (normed) {
                        va = va * nsa + nta; vb = vb * nsb + ntb;
                        const f32x4 na = va * slope4, nb2 = vb * slope4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { va[e] = fmaxf(va[e], na[e]); vb[e] = fmaxf(vb[e], nb2[e]); }      // LeakyReLU (0 < slope < 1)
                    }
                    uint4 hi, lo;
                    split_hi_lo_8(va, vb, hi, lo);
                    hi.x = real ? hi.x : 0u; hi.y = real ? hi.y : 0u; hi.z = real ? hi.z : 0u; hi.w = real ? hi.w : 0u;
                    *reinterpret_cast<uint4*>(d) = hi;
                    if (NPP == 2) {
                        lo.x = real ? lo.x : 0u; lo.y = real ? lo.y : 0u; lo.z = real ? lo.z : 0u; lo.w = real ? lo.w : 0u;
                        *reinterpret_cast<uint4*>(d + 2 * kS2Plane) = lo;
                    }
                } else {
                    uint4 x = uint4{S.pv[it][0][0], S.pv[it][0][1], S.pv[it][0][2], S.pv[it][0][3]};
                    if (normed) x = norm_lrelu_8(x, nsa, nsb, nta, ntb, slope2);
                    x.x = real ? x.x : 0u; x.y = real ? x.y : 0u; x.z = real ? x.z : 0u; x.w = real ? x.w : 0u;
                    *reinterpret_cast<uint4*>(d) = x;
                }
            }
        }
        if constexpr (!RESW) {
#define TS2D_WSTORE(K, R) { const int sl = tid + K * kS2Threads; if (K < WIT && (WB / 16 % kS2Threads == 0 || sl < WB / 16)) \
                *reinterpret_cast<uint4*>(sB + sl * 16) = R; }
            TS2D_WSTORE(0, w0) TS2D_WSTORE(1, w1) TS2D_WSTORE(2, w2) TS2D_WSTORE(3, w3) TS2D_WSTORE(4, w4)
            TS2D_WSTORE(5, w5) TS2D_WSTORE(6, w6) TS2D_WSTORE(7, w7) TS2D_WSTORE(8, w8)
#undef TS2D_WSTORE
        }
        lds_barrier();
        TS2D_STAMP_AT(a.prof, 0)
        advance(pf);
        prefetch(S, pf);                                   // the item DEPTH ahead (possibly of another tile) into the registers just converted

        f32x16 acc_c[2][NTW];                              // fresh accumulator per chunk (accuracy)
        const unsigned char* pw = smem8 + bbase + (RESW ? cur.c * WB : 0);
        __builtin_amdgcn_s_setprio(1);
        // fragments of tap t+1 are read while the MFMAs of tap t run (two register sets, one scheduling region per tap: the plain
        // loop waited for every tap's six reads before its MFMAs - in-kernel stamps: the 54-MFMA phase took 6-8 k cycles, twice the
        // matrix-pipe time of the two waves of a SIMD); BN = 128 keeps the plain loop (registers)
        auto load_frags = [&](half8 (&fa)[2][NPP], half8 (&fb)[NTW][NPP], int tap) {
            const int dy = tap / 3, dx = tap - 3 * dy;
            const int toff = (dy * kS2PW + (dx & 1) * 33 + (dx >> 1)) * 16;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int p = 0; p < NPP; ++p) fa[mt][p] = *reinterpret_cast<const half8*>(smem8 + abase + mt * 2 * kS2PW * 16 + p * 2 * kS2Plane + toff);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int p = 0; p < NPP; ++p) fb[nt][p] = *reinterpret_cast<const half8*>(pw + tap * WTAP + p * 2 * BN * 16 + nt * 512);
        };
        auto mma = [&](half8 (&fa)[2][NPP], half8 (&fb)[NTW][NPP], bool first) {
            if constexpr (NP == 3) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][1], fb[nt][0], first ? kZero16 : acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][0], fb[nt][1], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][0], fb[nt][0], acc_c[mt][nt], 0, 0, 0);
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][0], fb[nt][0], first ? kZero16 : acc_c[mt][nt], 0, 0, 0);
            }
        };
        if constexpr (BN == 64) {
            half8 fa0[2][NPP], fb0[NTW][NPP], fa1[2][NPP], fb1[NTW][NPP];
            load_frags(fa0, fb0, 0);
#define TS2D_TAP2(T) { if constexpr ((T) + 1 < 9) { if constexpr ((T) & 1) load_frags(fa0, fb0, (T) + 1); else load_frags(fa1, fb1, (T) + 1); } \
                if constexpr ((T) & 1) mma(fa1, fb1, false); else mma(fa0, fb0, (T) == 0); \
                __builtin_amdgcn_sched_barrier(0); }
            TS2D_TAP2(0) TS2D_TAP2(1) TS2D_TAP2(2) TS2D_TAP2(3) TS2D_TAP2(4) TS2D_TAP2(5) TS2D_TAP2(6) TS2D_TAP2(7) TS2D_TAP2(8)
#undef TS2D_TAP2
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                half8 fa[2][NPP], fb[NTW][NPP];
                load_frags(fa, fb, tap);
                mma(fa, fb, tap == 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (younger) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc_t[mt][nt] += acc_c[mt][nt];

        if (cur.c == nchunks - 1) {                        // (uniform) the tile is complete
            // ---- epilogue: C/D map of the 32x32 MFMA: column = lane & 31 (output channel), row = (i & 3) + 8 (i >> 2) + 4 h (pixel ox)
            const float oscale = *a.oscale;
            const size_t img_el = (size_t)a.Ht * a.Wt * a.Cout;
            const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<ST*>(a.dst) + (size_t)nimg0 * img_el, 0, (int)(img_el * sizeof(ST)), 0x00020000);
            float st_s[NTW], st_q[NTW], st_k[NTW];
            TS2D_STAMP_AT(a.prof, 3)
            float bvs[NTW];       // every bias value before the first store: a load issued between stores waits (in-order vmcnt) for the stores ahead of it
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) bvs[nt] = a.bias[n0col + wn * (BN / 2) + nt * 32 + r];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int co = n0col + wn * (BN / 2) + nt * 32 + r;
                const float bv = bvs[nt];
                const float kv = stat_pivot(round_act<ST>(__builtin_fmaf(acc_t[0][nt][0], oscale, bv)));      // shifted statistics (kernels.h)
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int oy = ty0 + 2 * wm + mt, ox = tx0 + 4 * h;
                    const unsigned voff = (unsigned)(((oy * a.Wt + ox) * a.Cout + co) * (int)sizeof(ST));
#pragma unroll
                    for (int i2 = 0; i2 < 16; ++i2) {
                        const unsigned soff = (unsigned)((((i2 & 3) + 8 * (i2 >> 2)) * a.Cout) * (int)sizeof(ST));      // scalar
                        float v = __builtin_fmaf(acc_t[mt][nt][i2], oscale, bv);
                        buffer_store_act<ST>(v, rsd, voff, soff);
                        const float d = round_act<ST>(v) - kv;                           // statistics of what is stored
                        s += d; q = __builtin_fmaf(d, d, q);
                        acc_t[mt][nt][i2] = 0.f;
                    }
                }
                st_s[nt] = s; st_q[nt] = q; st_k[nt] = kv;
            }
            TS2D_STAMP_AT(a.prof, 4)
            lds_barrier();                                         // every wave is done with the LDS images (the output stores stay in flight)
            float* red = reinterpret_cast<float*>(smem8);          // [wm 4][column BN] x (S, Q, K, n)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                float s = st_s[nt], q = st_q[nt];
                s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
                if (h == 0) stat_wave_put(red, wm * BN + wn * (BN / 2) + nt * 32 + r, s, q, st_k[nt], 64.f);
            }
            lds_barrier();
            if (tid < BN) stat_tile_store(red, 4, BN, tid, a.part + ((size_t)(nimg0 * tpi + tin) * a.Cout + n0col + tid) * 4);
            TS2D_STAMP_AT(a.prof, 5)
        }
        advance(cur);
    };
    for (int i = 0; i < nitems; i += DEPTH) {
        item(sg0);
        if constexpr (DEPTH == 2) { if (i + 1 < nitems) item(sg1); }
    }
    TS2D_PROF_FLUSH(a.prof)
}

}  // namespace ts2d
