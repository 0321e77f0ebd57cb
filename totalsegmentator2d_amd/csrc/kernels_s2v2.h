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
//
// Round 4.  In-kernel stamps (profiles/r04_s2v2_stamps.txt, enc2.c0, per item of 6.9 k cycles of MFMAs): 5.2 k cycles between the
// conversion barrier and the first MFMA - the wave sat at the ISSUE of the next item's ten patch loads.  Each of those wave
// instructions touched 32 different 128-byte lines (32 pixels x 2 lanes x 16 bytes) and the address path retires about one line per
// cycle: 80 instructions x ~64 cycles per item and CU, all eight waves queueing at once in front of their MFMAs.  Now (fp32 storage):
//   * FOUR lanes per pixel: a lane loads one 16-byte quarter of the chunk's 64-byte slice, a wave instruction covers 16 pixels =
//     16 lines (the conversion works on 4 channels per unit, the LDS writes are ds_write_b64);
//   * the loads are SPREAD over the item's taps, one unit per tap, instead of being issued in one burst;
//   * the per-chunk weight block travels L2 -> LDS by global_load_lds instead of through registers and ds_write_b128 (72 wave stores
//     per item less on the LDS store path; 36 registers, which pay for the next point);
//   * BN = 128 reads the next tap's fragments during the current tap's MFMAs, as BN = 64 did;
//   * the zero-selects of the padding units run on border tiles only (wave-uniform branch).
#pragma once
#include "kernels_f16x3.h"
#include "kernels_h32.h"

namespace ts2d {

constexpr int kS2Threads = 512;
constexpr int kS2PW = 66, kS2Slots = 17 * kS2PW;      // patch: 17 rows x (33 even + 33 odd columns)
// plane stride = 64 mod 128 bytes: the four lanes of a pixel write 8 bytes each into two planes (ds_write_b64, fp32 storage) - with the
// planes 16 dwords apart in the bank map a 16-lane group covers 32 distinct banks (at 32 mod 128 bytes, the unpadded stride, two of its
// four pixels collided: 16-19 % LDS bank conflicts in the round-4 counters); the fragment reads are conflict-free at any alignment
constexpr int kS2Plane = (kS2Slots + 2) * 16;
static_assert(kS2Plane % 128 == 64, "plane stride of conv3x3s2_v2");

// Round 3: PERSISTENT workgroups.  In-kernel stamps of round 2 (profiles/r02_phase_stamps.txt, enc1.c0): of ~27 000 cycles per
// two-chunk workgroup 8 800 were spent waiting for the first chunk's patch - one workgroup per CU (LDS), nothing else resident to
// hide it, 16 384 workgroups per launch.  Now a workgroup (one per CU, fixed XCD lane and column tile as conv3x3_f16x3_qp) walks
// ONE stream of (tile, chunk) items with the NEXT item's raw patch and scale / shift in flight in registers during the current
// item's MFMAs and epilogue - also across tile boundaries.  RESW: every chunk's weight block of the column tile stays RESIDENT in
// LDS for the workgroup's life when it fits beside the patch (enc1.c0: 2 x 36 KB), so the per-chunk weight staging disappears.
// Padding is decided per tile: a unit of patch row 0 / column 0 on a top / left border tile stores zeros (stride 2, pad 1 and even
// extents: the patch never leaves the image at the bottom or the right).
// FLEX (round 5): output tile a.TH x a.TW instead of 8 x 32 - any shape with TH | Ht, TW | Wt, TW % 4 == 0, TH * TW <= 256 and a patch of
// (2 TH + 1) rows x (2 TW + 2) slots <= 1122 (the engine picks it: 5 x 48 for a level of 80 x 48, 7 x 36 for 28 x 36 / 56 x 72, 16 x 16 for
// the 16 x 16 level of the canonical net).  Tiles divide the level exactly, so - as with the fixed tile - a patch leaves the image at the top
// and the left only.  Row m of the tile's GEMM is pixel (m / TW, m % TW): lane bases by division (once), tap offsets in registers, rows
// past TH * TW read a dummy slot and are masked in the epilogue (which addresses per run of 4 pixels).
// K32 (round 6, 16-bit mode): chunks of 32 channels.  With fp16 storage a pixel's 32-channel slice is as many bytes as the split mode's hi + lo of 16, so
// the second PART of every LDS image (the split mode's lo planes / lo weight blocks) holds channels 16-31 of the chunk instead: half the items, barriers
// and staging phases per tile for the same MFMAs (two per tap and block pair: part 0 x part 0 + part 1 x part 1), four lanes per pixel in the loads like
// the fp32 path, no per-chunk accumulator (one product, fp32 accumulation: the 16-bit contract).  The 16-channel form at one product spent two thirds of an
// item outside its taps (profiles/r04_s2v2_stamps.txt: conversion 4.7 k + barriers 2.1 k + accumulate 3.9 k cycles beside 5.2 k of taps, single-buffered).
template <int BN, typename ST, int NP, bool RESW, bool FLEX = false, bool K32 = false>
__global__ __launch_bounds__(kS2Threads, 2) void conv3x3s2_v2(const ConvArgs a) {
    static_assert(!K32 || (NP == 1 && sizeof(ST) == 2), "32-channel chunks: fp16 storage, one product");
    constexpr int NPP = NP == 3 ? 2 : 1;                   // fp16 parts per value
    constexpr int PARTS = K32 ? 2 : NPP;                   // LDS planes / weight blocks per (tap, k half): hi, lo - or channels 0-15, 16-31 of a K32 chunk
    constexpr int CK = K32 ? 32 : 16;                      // channels per chunk
    constexpr int NTW = BN / 64;                           // 32-column MFMA tiles per wave (a wave owns BN / 2 columns)
    constexpr int WTAP = PARTS * 2 * BN * 16, WB = 9 * WTAP; // weight bytes per tap / per chunk
    constexpr bool F32 = sizeof(ST) == 4;
    constexpr bool Q4 = F32 || K32;                        // four lanes per pixel (a pixel's slice of the chunk is 64 bytes)
    // staging unit = 16 bytes of a pixel's slice of the chunk.  fp32 storage: a quarter (4 channels), four lanes per pixel, 9 units per
    // thread (9 x 128 slots >= 1122); fp16 storage: a half (8 channels), two lanes per pixel, 5 units per thread (5 x 256 slots) - K32: a quarter
    // (8 of 32 channels), four lanes per pixel
    constexpr int MAXU = Q4 ? 9 : 5;
    constexpr int USTEP = Q4 ? 128 : 256;                  // slots covered by one unit index of the whole workgroup
    constexpr int DEPTH = BN == 64 ? 2 : 1;                // items kept in flight in registers (BN = 64: enc1.c0 runs at the HBM rate; two items keep that pipe fed)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int TH = FLEX ? a.TH : 8, TW = FLEX ? a.TW : 32;
    const int PW2 = FLEX ? 2 * TW + 2 : kS2PW, HO = FLEX ? TW + 1 : 33, PSL = FLEX ? (2 * TH + 1) * PW2 : kS2Slots;      // patch row pitch, odd-column offset, slots

    // ---- this workgroup's tiles: virtual block v = blockIdx.x + k * gridDim.x -> (xcd, column tile) fixed, pixel tile mtile0 + k * mstep
    const int xcd = blockIdx.x & 7, q80 = blockIdx.x >> 3;
    const int qm0 = q80 / a.n_ctiles;                       // (any tile count: round 5; the grid is a multiple of 8 * n_ctiles)
    const int ctile = q80 - qm0 * a.n_ctiles, n0col = ctile * BN;
    const int mtile0 = qm0 * 8 + xcd, mstep = ((int)(gridDim.x >> 3) / a.n_ctiles) * 8;
    if (mtile0 >= a.n_mtiles) return;
    const int ntl = (a.n_mtiles - 1 - mtile0) / mstep + 1;
    const int nchunks = a.C0 / CK;                         // the strided conv never reads a concat
    const int tpi = a.tiles_x * a.tiles_y;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 3, wn = w >> 2;
    const int r = lane & 31, h = lane >> 5;

    unsigned char* sA = smem8;                             // [part][h][slot] x 16 B
    unsigned char* sB = smem8 + PARTS * 2 * kS2Plane;      // [chunk (RESW)][tap][part][h][column BN] x 16 B

    // ---- staging plan (tile-independent).  Slot q = patch row q / 66, then the 33 even columns, then the 33 odd ones (the 66th slot of
    //      a row is unused).  fp32: unit it of thread tid = slot (tid >> 2) + 128 it, quarter sub = tid & 3 (channels 4 sub .. 4 sub + 3 of the
    //      chunk: plane h = sub >> 1, bytes 8 (sub & 1) .. of the slot); fp16: slot 32 (8 it + w) + (lane & 7) + 8 (lane >> 4), octet
    //      sub = (lane >> 3) & 1 (plane h = sub).
    const int sub = Q4 ? (tid & 3) : ((lane >> 3) & 1);    // (K32: channels 8 sub .. 8 sub + 7 of the chunk = part sub >> 1, plane h = sub & 1: plane index = sub)
    const int slot0 = Q4 ? (tid >> 2) : (32 * w + (lane & 7) + 8 * (lane >> 4));
    unsigned rel[MAXU];                                    // byte offset of the unit from the patch origin, 0x80000000 = no unit
    unsigned emask = 0;                                    // per unit: bit 0 = patch row 0, bit 1 = patch column 0 (the padding candidates)
    const int lw0 = F32 ? (sub >> 1) * kS2Plane + slot0 * 16 + (sub & 1) * 8 : sub * kS2Plane + slot0 * 16;      // LDS write address of unit 0; unit it: + 16 USTEP it
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int q = slot0 + USTEP * it;
        const int py = q / PW2, rem = q - py * PW2;
        const int half = rem >= HO ? 1 : 0, px = 2 * (rem - HO * half) + half;
        const bool exists = q < PSL && px <= 2 * TW;
        rel[it] = exists ? (unsigned)(((py * a.Win + px) * a.C0) * (int)sizeof(ST) + 16 * sub) : 0x80000000u;
        if (exists) emask |= ((py == 0 ? 1u : 0u) | (px == 0 ? 2u : 0u)) << (2 * it);
        if (q < PSL && !exists) {                          // the unused last slot of a row: zero once (never staged, read by no fragment)
            if constexpr (F32) {
                *reinterpret_cast<uint2*>(sA + lw0 + it * USTEP * 16) = uint2{0u, 0u};
                if (NPP == 2) *reinterpret_cast<uint2*>(sA + lw0 + it * USTEP * 16 + 2 * kS2Plane) = uint2{0u, 0u};
            } else {
                *reinterpret_cast<uint4*>(sA + lw0 + it * USTEP * 16) = uint4{0u, 0u, 0u, 0u};
            }
        }
    }
    if constexpr (RESW) {                                  // resident weights: chunks x 9 taps of this column tile, one linear copy each
        constexpr int WIT = (WB / 16 + kS2Threads - 1) / kS2Threads;
        for (int c = 0; c < nchunks; ++c) {
            // (the weight image is the split mode's: [16-channel chunk][column tile][tap][hi, lo][h][column]; the 16-bit forms read the hi blocks - K32 those
            //  of the chunks 2 c and 2 c + 1 as its two parts)
            const uint4* wsrc = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(a.wph) + ((size_t)(K32 ? 2 * c : c) * a.n_ctiles + ctile) * (9 * 2 * 2 * BN * 16));
            constexpr int C16 = 9 * 4 * BN;                // uint4s of one 16-channel chunk's block
#pragma unroll
            for (int k = 0; k < WIT; ++k) {
                const int sl = tid + k * kS2Threads;
                if (WB / 16 % kS2Threads == 0 || sl < WB / 16) {
                    int src = sl;
                    if constexpr (K32) src = ((sl / (2 * BN)) & 1) * (C16 * a.n_ctiles) + (sl / (4 * BN)) * (4 * BN) + sl % (2 * BN);
                    else if constexpr (NPP == 1) src = (sl / (2 * BN)) * (4 * BN) + sl % (2 * BN);
                    *reinterpret_cast<uint4*>(sB + c * WB + sl * 16) = wsrc[src];
                }
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
        nimg = udiv_magic(mtile, a.mg_tpi); tin = mtile - nimg * tpi;
        tyi = udiv_magic(tin, a.mg_tx); txi = tin - tyi * a.tiles_x;
    };
    const size_t img_px = (size_t)a.Hin * a.Win;
    // raw patch + scale / shift of one item in flight (fp32: 4 channels per thread -> nsa / nta only; fp16: 8 channels)
    struct Stage { u32x4 pv[MAXU]; f32x4 nsa, nsb, nta, ntb; };
    Stage sg0, sg1;
    const bool normed = a.sc0 != nullptr;
    auto load_norm = [&](Stage& S, int nimg, int c) {      // scale / shift of this thread's channels (not normalised: loaded, not used)
        const int co = F32 ? 4 * sub : 8 * sub;
        const float* ps = (normed ? a.sc0 : a.bias) + (normed ? (size_t)nimg * a.C0 + c * CK + co : 0);
        const float* pt = (normed ? a.sh0 : a.bias) + (normed ? (size_t)nimg * a.C0 + c * CK + co : 0);
        S.nsa = *reinterpret_cast<const f32x4*>(ps); S.nta = *reinterpret_cast<const f32x4*>(pt);
        if constexpr (!F32) { S.nsb = *reinterpret_cast<const f32x4*>(ps + 4); S.ntb = *reinterpret_cast<const f32x4*>(pt + 4); }
    };
    // a request = its scalar part (descriptor, origin, chunk offset) + one load per unit; the units of the NEXT request are issued one
    // per tap inside the MFMA phase (below), the very first ones here in a burst
    struct Req { __amdgpu_buffer_rsrc_t rs; unsigned org; int soff; };
    auto request = [&](const Item& t, int& nimg) {
        int tyi, txi, tin;
        tile_origin(t.k, nimg, tyi, txi, tin);
        Req q;
        q.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(reinterpret_cast<const ST*>(a.src0)) + (size_t)nimg * img_px * a.C0, 0,
                                                 (int)(img_px * a.C0 * sizeof(ST)), 0x00020000);
        // patch origin (2 ty0 - 1, 2 tx0 - 1) may lie one row / column outside the image: unsigned wrap-around is fine, the affected
        // units are padding (zeroed at conversion)
        q.org = FLEX ? (unsigned)((((2 * TH * tyi - 1) * a.Win + 2 * TW * txi - 1) * a.C0) * (int)sizeof(ST))
                     : (unsigned)((((16 * tyi - 1) * a.Win + 64 * txi - 1) * a.C0) * (int)sizeof(ST));
        q.soff = t.c * CK * (int)sizeof(ST);
        return q;
    };
    auto load_unit = [&](Stage& S, const Req& q, int it) {
        const unsigned vo = rel[it] == 0x80000000u ? 0x80000000u : q.org + rel[it];
        S.pv[it] = __builtin_amdgcn_raw_buffer_load_b128(q.rs, vo, q.soff, 0);
    };

    Item cur{0, 0}, pf{0, 0};                              // item being computed; last item requested
    {
        int nimg;
        const Req q = request(cur, nimg);
#pragma unroll
        for (int it = 0; it < MAXU; ++it) load_unit(sg0, q, it);
        load_norm(sg0, nimg, cur.c);
        if constexpr (DEPTH == 2) {
            advance(pf);
            const Req q1 = request(pf, nimg);
#pragma unroll
            for (int it = 0; it < MAXU; ++it) load_unit(sg1, q1, it);
            load_norm(sg1, nimg, pf.c);
        }
    }

    // ---- lane constants of the MFMA phase: output pixel (2 wm + mt, r) reads patch row 2 (2 wm + mt) + dy, slot (dx & 1) 33 + r + (dx >> 1)
    const int abase = h * kS2Plane + ((4 * wm) * kS2PW + r) * 16;                  // + mt * 2 * 66 * 16 + part * 2 * Plane + tap offset
    int fxb[2] = {0, 0};                                                           // FLEX: per M tile (row m = 64 wm + 32 mt + r -> pixel (m / TW, m % TW): patch row 2 ty, slot tx)
    if constexpr (FLEX) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = 64 * wm + 32 * mt + r;
            const int ty = fdiv(m, a.inv_tw), tx = m - ty * TW;
            fxb[mt] = h * kS2Plane + (m < TH * TW ? (2 * ty * PW2 + tx) * 16 : 0);
        }
    }
    const int bbase = PARTS * 2 * kS2Plane + h * BN * 16 + (wn * (BN / 2) + r) * 16;  // + chunk * WB (RESW) + tap * WTAP + part * 2 * BN * 16 + nt * 512
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
        const int ty0 = tyi * TH, tx0 = txi * TW;
        TS2D_STAMP_AT(a.prof, 3)
        lds_barrier();                                     // the previous item's MFMA reads of LDS (and its statistics exchange) are done
        TS2D_STAMP_AT(a.prof, 1)                           // (LDS-only barriers: a __syncthreads() here would wait for the previous tile's output stores)
        const f32x4 nsa = S.nsa, nsb = S.nsb, nta = S.nta, ntb = S.ntb;
        // ---- weights of this chunk (not resident): one linear block of 1-KiB pieces, L2 -> LDS by DMA.  hipcc answers the use of ANY
        //      register with a load pending by vmcnt(0) while a DMA is in flight: every loaded register of this item (raw patch, scale /
        //      shift) is therefore touched - waited for - BEFORE the DMA is issued; the DMA then lands behind the conversion.
        if constexpr (!RESW) {
#pragma unroll
            for (int it = 0; it < MAXU; ++it) asm volatile("" :: "v"(S.pv[it]));
            asm volatile("" :: "v"(nsa), "v"(nta));
            if constexpr (!F32) asm volatile("" :: "v"(nsb), "v"(ntb));
            TS2D_STAMP_AT(a.prof, 2)
            const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wph) + ((size_t)(K32 ? 2 * cur.c : cur.c) * a.n_ctiles + ctile) * (9 * 2 * 2 * BN * 16) + lane * 16;
            constexpr int NPIECE = WB / 1024, RUN = 2 * BN * 16 / 1024;      // f16 mode: the hi parts only - runs of RUN pieces out of every 2 RUN
            const size_t c16 = (size_t)a.n_ctiles * (9 * 2 * 2 * BN * 16);   // (K32) bytes from the block of chunk 2 c to that of chunk 2 c + 1
#pragma unroll
            for (int j = 0; j < (NPIECE + 7) / 8; ++j) {
                const int pc = w + 8 * j;
                if (NPIECE % 8 == 0 || pc < NPIECE) {
                    size_t so;                             // piece pc of the LDS image [tap][part][RUN pieces] <- piece of the source image [tap][hi, lo][RUN]
                    if constexpr (K32) so = (size_t)((pc / RUN) & 1) * c16 + (size_t)((pc / (2 * RUN)) * 2 * RUN + pc % RUN) * 1024;
                    else so = (size_t)(NPP == 2 ? pc : (pc / RUN) * 2 * RUN + pc % RUN) * 1024;
                    __builtin_amdgcn_global_load_lds(wsrc + so, (lds_ptr)(sB + pc * 1024), 16, 0, 0);
                }
            }
        }
        // ---- patch: InstanceNorm + LeakyReLU on the fly, split into fp16 hi / lo; padding units store zeros
        constexpr unsigned kRow0 = Q4 ? 0x15555u : 0x155u, kCol0 = Q4 ? 0x2AAAAu : 0x2AAu;
        const unsigned pad = (tyi == 0 ? kRow0 : 0u) | (txi == 0 ? kCol0 : 0u);      // (uniform) which emask bits mean "outside the image" here
        const bool border = pad != 0u;                     // (uniform) interior tiles skip the zero-selects
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            if (rel[it] != 0x80000000u) {
                unsigned char* d = sA + lw0 + it * USTEP * 16;
                const bool real = ((emask & pad) >> (2 * it) & 3u) == 0u;
                if constexpr (F32) {
                    f32x4 va = __builtin_bit_cast(f32x4, S.pv[it]);
                    if (normed) {
                        va = va * nsa + nta;
                        const f32x4 na = va * slope4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) va[e] = fmaxf(va[e], na[e]);      // LeakyReLU (0 < slope < 1)
                    }
                    uint2 hi, lo;
                    split_hi_lo_4(va, hi, lo);
                    if (border) { hi.x = real ? hi.x : 0u; hi.y = real ? hi.y : 0u; lo.x = real ? lo.x : 0u; lo.y = real ? lo.y : 0u; }
                    *reinterpret_cast<uint2*>(d) = hi;
                    if (NPP == 2) *reinterpret_cast<uint2*>(d + 2 * kS2Plane) = lo;
                } else {
                    uint4 x = uint4{S.pv[it][0], S.pv[it][1], S.pv[it][2], S.pv[it][3]};
                    if (normed) x = norm_lrelu_8(x, nsa, nsb, nta, ntb, slope2);
                    if (border) { x.x = real ? x.x : 0u; x.y = real ? x.y : 0u; x.z = real ? x.z : 0u; x.w = real ? x.w : 0u; }
                    *reinterpret_cast<uint4*>(d) = x;
                }
            }
        }
        TS2D_STAMP_AT(a.prof, 6)
        if constexpr (!RESW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the weight DMA has landed (issued before the conversion)
        lds_barrier();
        TS2D_STAMP_AT(a.prof, 0)
        advance(pf);
        int nimg_pf;
        const Req rq = request(pf, nimg_pf);               // the item DEPTH ahead (possibly of another tile) into the registers just converted:
                                                           // its units are issued one per tap below
        f32x16 acc_c[K32 ? 1 : 2][K32 ? 1 : NTW];          // fresh accumulator per chunk (accuracy; K32: straight into the tile's)
        const unsigned char* pw = smem8 + bbase + (RESW ? cur.c * WB : 0);
        __builtin_amdgcn_s_setprio(1);
        // fragments of tap t+1 are read while the MFMAs of tap t run (two register sets, one scheduling region per tap: the plain
        // loop waited for every tap's reads before its MFMAs)
        auto load_frags = [&](half8 (&fa)[2][PARTS], half8 (&fb)[NTW][PARTS], int tap) {
            const int dy = tap / 3, dx = tap - 3 * dy;
            const int toff = (dy * PW2 + (dx & 1) * HO + (dx >> 1)) * 16;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int p = 0; p < PARTS; ++p)
                    fa[mt][p] = *reinterpret_cast<const half8*>(smem8 + (FLEX ? fxb[mt] : abase + mt * 2 * kS2PW * 16) + p * 2 * kS2Plane + toff);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int p = 0; p < PARTS; ++p) fb[nt][p] = *reinterpret_cast<const half8*>(pw + tap * WTAP + p * 2 * BN * 16 + nt * 512);
        };
        auto mma = [&](half8 (&fa)[2][PARTS], half8 (&fb)[NTW][PARTS], bool first) {
            if constexpr (K32) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt) acc_t[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][p], fb[nt][p], acc_t[mt][nt], 0, 0, 0);
            } else if constexpr (NP == 3) {
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
        {
            half8 fa0[2][PARTS], fb0[NTW][PARTS], fa1[2][PARTS], fb1[NTW][PARTS];
            load_frags(fa0, fb0, 0);
            // one patch load of the next request per tap, behind the tap's MFMAs (units MAXU .. 8 do not exist in the 16-bit mode)
#define TS2D_TAP2(T) { if constexpr ((T) + 1 < 9) { if constexpr ((T) & 1) load_frags(fa0, fb0, (T) + 1); else load_frags(fa1, fb1, (T) + 1); } \
                if constexpr ((T) & 1) mma(fa1, fb1, false); else mma(fa0, fb0, (T) == 0); \
                if constexpr ((T) < MAXU) load_unit(S, rq, (T)); \
                __builtin_amdgcn_sched_barrier(0); }
            TS2D_TAP2(0) TS2D_TAP2(1) TS2D_TAP2(2) TS2D_TAP2(3) TS2D_TAP2(4) TS2D_TAP2(5) TS2D_TAP2(6) TS2D_TAP2(7) TS2D_TAP2(8)
#undef TS2D_TAP2
        }
        load_norm(S, nimg_pf, pf.c);
        __builtin_amdgcn_s_setprio(0);
        if (younger) __builtin_amdgcn_s_setprio(1);
        TS2D_STAMP_AT(a.prof, 5)
        if constexpr (!K32) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc_t[mt][nt] += acc_c[mt][nt];
        }

        if (cur.c == nchunks - 1) {                        // (uniform) the tile is complete
            // ---- epilogue: C/D map of the 32x32 MFMA: column = lane & 31 (output channel), row = (i & 3) + 8 (i >> 2) + 4 h (pixel ox)
            const float oscale = *a.oscale;
            const size_t img_el = (size_t)a.Ht * a.Wt * a.Cout;
            const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<ST*>(a.dst) + (size_t)nimg0 * img_el, 0, (int)(img_el * sizeof(ST)), 0x00020000);
            float st_s[NTW], st_q[NTW], st_k[NTW];
            float nvalid = 64.f;                                   // pixels of this wave inside the tile
            float bvs[NTW];       // every bias value before the first store: a load issued between stores waits (in-order vmcnt) for the stores ahead of it
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) bvs[nt] = a.bias[n0col + wn * (BN / 2) + nt * 32 + r];
            if constexpr (FLEX) {
                // a lane's 16 rows of an M tile = 4 runs of 4 consecutive pixels (TW % 4 == 0: a run stays inside one tile row): one division and
                // one address per run; runs past TH * TW are dropped (out-of-range offset) and kept out of the statistics
                const int thw = TH * TW;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    st_k[nt] = stat_pivot(round_act<ST>(__builtin_fmaf(acc_t[0][nt][0], oscale, bvs[nt])));
                    st_s[nt] = 0.f; st_q[nt] = 0.f;
                }
                int cnt = 0;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int m0 = 64 * wm + 32 * mt + 4 * h + 8 * j;
                        asm volatile("" : "+v"(m0));               // (not hoisted out of the tile loop into 16 live registers)
                        const int ty = fdiv(m0, a.inv_tw), tx = m0 - ty * TW;
                        const bool ok = m0 < thw;
                        const unsigned vrun = ok ? (unsigned)((((ty0 + ty) * a.Wt + tx0 + tx) * a.Cout + n0col + wn * (BN / 2) + r) * (int)sizeof(ST)) : 0x80000000u;
                        cnt += ok ? 4 : 0;
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4)
#pragma unroll
                            for (int nt = 0; nt < NTW; ++nt) {
                                const int i2 = 4 * j + e4;
                                float v = __builtin_fmaf(acc_t[mt][nt][i2], oscale, bvs[nt]);
                                buffer_store_act<ST>(v, rsd, vrun, (unsigned)((e4 * a.Cout + nt * 32) * (int)sizeof(ST)));
                                const float d = ok ? round_act<ST>(v) - st_k[nt] : 0.f;
                                st_s[nt] += d; st_q[nt] = __builtin_fmaf(d, d, st_q[nt]);
                                acc_t[mt][nt][i2] = 0.f;
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                cnt += __shfl_xor(cnt, 32);
                nvalid = (float)cnt;
            } else {
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
            }
            TS2D_STAMP_AT(a.prof, 4)
            lds_barrier();                                         // every wave is done with the LDS images (the output stores stay in flight)
            float* red = reinterpret_cast<float*>(smem8);          // [wm 4][column BN] x (S, Q, K, n)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                float s = st_s[nt], q = st_q[nt];
                s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
                if (h == 0) stat_wave_put(red, wm * BN + wn * (BN / 2) + nt * 32 + r, s, q, st_k[nt], nvalid);
            }
            lds_barrier();
            if (tid < BN) stat_tile_store(red, 4, BN, tid, a.part + ((size_t)(nimg0 * tpi + tin) * a.Cout + n0col + tid) * 4);
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
