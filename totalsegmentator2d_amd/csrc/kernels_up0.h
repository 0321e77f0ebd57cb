// Level-0 decoder entry (SURVEY K5 + K6 at 512 x 512: ConvTranspose2d 64 -> 32, cat((up, skip), 1), Conv2d 3x3 64 -> 32 - dec0.c0 of the
// canonical net, the single slowest op of the step) as a PERSISTENT kernel in the structure of conv3x3_res32:
//   * the composition of kernels_upc.h (the transposed conv folded into the "up" half of the 3x3 conv: a 2x2 convolution over the
//     coarse tensor with parity-specific weights; the upsampled tensor never exists), wave w = output parity class (A, B) =
//     (w >> 1, w & 1): its 64 pixels are the 4 x 16 coarse positions (I, J) of the 8 x 32 tile, output pixel (2I + A, 2J + B);
//   * the skip half's weights (9 x 32 x 32 x (hi + lo) fp16 = 36 KB) RESIDENT in LDS for the workgroup's life (conv3x3_upc staged
//     them per tile and chunk: 36 KB of LDS writes and two barrier pairs per 256 pixels), two workgroups per CU;
//   * the composed weights of a wave's parity (32 KB; the four parities do not fit beside the patches) stream from L2 through a
//     register ring two k-steps deep, in MFMA fragment order (1 KB per load);
//   * v_mfma_f32_16x16x32_f16, transposed product D[cout][pixel] (a lane holds 4 consecutive channels of one pixel: 16-byte stores),
//     one K step = 32 channels: 8 k-steps (2 halves of the 64 coarse channels x 4 taps (dI, dJ)) + 9 (the skip taps) per tile;
//   * per tile FOUR phases on ONE patch region (coarse planes, then skip planes): convert coarse | MFMA composed | convert skip |
//     MFMA skip + epilogue.  The raw patches of the NEXT tile are requested at the start of the last phase - the only phase without
//     global weight loads: vmcnt retires in order, so a weight load issued behind an HBM prefetch cannot be waited for without
//     waiting for the prefetch - and the first two k-steps' weights before the output stores.
// LDS images as in kernels_res32.h (k-group-major planes of 16-byte slots, plane strides multiples of 256 B).  Skip patch columns are
// stored even-first / odd-second (the stride-2 pixel walk of a parity class is then a walk over consecutive slots); the 16 coarse
// planes (6 x 18 pixels) sit two per skip-plane window, clear of the 12 spare slots that hold the statistics scratch.
// Round 6:
//   * channel order of the MFMA rows permuted on the host (row 4 g + i of block cb = channel 8 g + 4 cb + i): a lane holds EIGHT consecutive channels
//     of its pixel - one 16-byte store per pixel in the 16-bit mode (was two 8-byte pieces of a 64-byte record), 32 contiguous bytes in fp32;
//   * the statistics leave the tile loop's serial section: pivot K = the interior bias (no broadcast, no dependency on the tile's data), each WAVE
//     writes its own partial (S, Q, K, 64) per channel straight to global memory (4 partials per tile; finalize_stats combines any number) - the LDS
//     scratch, the barrier-coupled merge by wave 0 and its global store are gone;
//   * 16-bit mode: the coarse planes have their own LDS region (the workgroup's 62 KB still fit twice), so a tile needs TWO barriers instead of four
//     (behind each conversion): a wave that is done with a tile starts converting the next one without waiting for the slowest wave's epilogue.
#pragma once
#include <type_traits>
#include "kernels_res32.h"

namespace ts2d {

struct Up0Args {
    const void* xc; const float* scc; const float* shc;   // coarse tensor NHWC [B, H/2, W/2, 64] (storage type ST) + its per-(n,c) scale / shift
    const void* xs; const float* scs; const float* shs;   // skip tensor NHWC [B, H, W, 32] + scale / shift
    const void* wc0;       // composed weights, fragment order [parity 4][k-step 8 = (half of Cb, dI, dJ)][hi,lo][cb 2][lane 64][8 halves]
    const void* wk0;       // skip half of the 3x3 weights, resident image [tap 9][hi,lo][g 4][cout 32][8 halves] (conv3x3_res32 layout)
    const float* bvar;     // [9 = (ry, rx)][32] bias variants (kernels_upc.h)
    const float* oscale;   // 1 / (common power-of-two pre-scale of both images)
    void* dst; float* part;       // raw NHWC [B, H, W, 32] output; InstanceNorm partials [n][tile (column-major)][32] x (S, Q, K, n)
                                  // (16-bit mode: [n][tile][wave 4][32] - one partial per wave)
    int B, H, W;           // output geometry: H % 8 == 0, W % 32 == 0
    int tiles_x, tiles_y, n_tiles, seg;      // as Res32Args
    float slope;
    unsigned long long* prof;      // diagnostic (TS2D_DBG=256): cycles of wave 0 in [0] convert coarse, [1] barrier, [2] composed MFMAs, [3] barrier,
                                   // [4] convert skip + barrier, [5] skip MFMAs, [6] epilogue + barrier - summed over the workgroup's tiles
};

constexpr int kU0CP = 1792;        // coarse plane: 108 slots (+ 4)
constexpr int kU0NT = 0;            // cache policy of the output stores.  nt (2) measured: stores 2.07 -> 2.12 ms (16-bit mode, 8-byte stores: 1.04 -> 1.28: partial
                                    // lines no longer merge in L2); nt on the patch loads as well: 2.24 / 1.35 ms (the halo re-reads of the neighbour tiles miss)
constexpr bool kU0Rot = false;       // k-step order rotated per tile, so that the workgroups do not all walk the same 4 KB of the weight image at once: measured
                                    // SLOWER (2.08 -> 2.21 ms, 16-bit mode 1.04 -> 1.14): same-address traffic is what the L2 serves best
constexpr bool kU0Burst = false;     // all 20 prefetch loads at the first tap (measured equal: the CU memory path is the bound either way; more spills)

template <typename ST, int NP>
__global__ __launch_bounds__(kBlock, 2) void conv3x3_up0(const Up0Args a) {
    constexpr int NPP = NP == 3 ? 2 : 1;
    constexpr int WB = 9 * NPP * 4 * 512;                  // resident skip weights (bytes)
    constexpr int NL = sizeof(ST) == 4 ? 2 : 1;            // 16-byte loads per staging unit (8 channels)
    constexpr int RING = NP == 3 ? 2 : 8;                  // k-steps of composed weights in flight per wave (16-bit mode: a k-step is 8 MFMAs - all eight requested up front)
    constexpr int NUS = 6, NUC = 4;                        // staging units per thread: skip 6 x 64 >= 340 pixels, coarse 4 x 32 >= 108
    constexpr int WIN = 2 * kResPS;                        // coarse planes of one (part, half) quadruple: two skip-plane windows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), pA = w >> 1, pB = w & 1;
    const int j = lane & 15, g = lane >> 4;

    const int tpi = a.tiles_x * a.tiles_y;
    const int t0 = (int)blockIdx.x * a.seg;
    if (t0 >= a.n_tiles) return;
    const int t1 = t0 + a.seg;
    const int n = t0 / tpi;
    int trem = t0 - n * tpi;
    int txi = trem / a.tiles_y, tyi = trem - txi * a.tiles_y;

    {   // resident skip weights: one linear copy (f16 mode: the hi parts only)
        const uint4* wsrc = reinterpret_cast<const uint4*>(a.wk0);
#pragma unroll
        for (int k = 0; k < (WB / 16 + kBlock - 1) / kBlock; ++k) {
            const int sl = tid + k * kBlock;
            if (sl < WB / 16) *reinterpret_cast<uint4*>(smem8 + sl * 16) = wsrc[NPP == 2 ? sl : (sl >> 7) * 256 + (sl & 127)];
        }
    }
    unsigned char* sP = smem8 + WB;                        // patch region: skip planes [part][g][slot] x 16 B / coarse planes
    constexpr int CO = NPP == 2 ? 0 : 4 * kResPS;          // coarse planes: on the skip planes (split mode: the two workgroups of a CU use all 160 KB) / behind them
    constexpr bool TWO_BARRIERS = NPP == 1;
    constexpr bool DIRECT = NPP == 1;                      // per-wave partials straight to global memory (split mode: measured 3 % SLOWER than the LDS merge - kept there)
    auto scratch = [&](int ww, int gg) -> float* {         // !DIRECT: per wave and k-group row [2 cb][3 = S, Q, K][4 channels] floats in the 12 spare slots of a plane
        return reinterpret_cast<float*>(sP + (2 * ww + (gg >> 1)) * kResPS + kResP * 16 + (gg & 1) * 96);
    };

    // ---- skip staging plan: unit it = patch pixel p = 64 it + 16 w + 2 (lane & 7) + (lane >> 5), channel group sg: a wave instruction
    //      covers 16 whole pixel records; lanes 0-7 hold the even pixels, lanes 32-39 the odd ones: 8 consecutive slots of a half row
    const int sg = (lane >> 3) & 3;
    unsigned rels[NUS]; int lws[NUS];
    unsigned emask = 0;                                    // per unit 4 bits: patch row 0 / row 9 / column 0 / column 33
#pragma unroll
    for (int it = 0; it < NUS; ++it) {
        const int p = 64 * it + 16 * w + 2 * (lane & 7) + (lane >> 5);
        const int py = p / kResPW, px = p - py * kResPW;
        rels[it] = (unsigned)(((py * a.W + px) * 32 + 8 * sg) * (int)sizeof(ST));
        lws[it] = sg * kResPS + (py * kResPW + (px & 1) * 17 + (px >> 1)) * 16;
        if (p < kResP) emask |= ((py == 0 ? 1u : 0u) | (py == 9 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == kResPW - 1 ? 8u : 0u)) << (4 * it);
    }
    const bool last_s = 64 * (NUS - 1) + 16 * w + 2 * (lane & 7) + (lane >> 5) < kResP;
    // ---- coarse staging plan: unit it = patch pixel p = 8 (w + 4 it) + (lane & 7) of the 6 x 18 patch, channel group cg = lane >> 3
    //      (half ks = cg >> 2 of the 64 channels, k-group cg & 3): a wave instruction covers 8 whole 64-channel records
    const int Wc = a.W >> 1, Hc = a.H >> 1;
    const int cg = lane >> 3;
    unsigned relc[NUC];
    unsigned cmask = 0;                                    // per unit 4 bits: patch row 0 / row 5 / column 0 / column 17
#pragma unroll
    for (int it = 0; it < NUC; ++it) {
        const int p = 8 * (w + 4 * it) + (lane & 7);
        const int py = p / 18, px = p - py * 18;
        relc[it] = (unsigned)(((py * Wc + px) * 64 + 8 * cg) * (int)sizeof(ST));
        if (p < 108) cmask |= ((py == 0 ? 1u : 0u) | (py == 5 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == 17 ? 8u : 0u)) << (4 * it);
    }
    const bool last_c = 8 * (w + 4 * (NUC - 1)) + (lane & 7) < 108;
    // coarse plane (part, ks, g) at (part * 2 + ks) * WIN + (g >> 1) * kResPS + (g & 1) * kU0CP
    const int lwc = CO + (cg >> 2) * WIN + ((cg >> 1) & 1) * kResPS + (cg & 1) * kU0CP + (8 * w + (lane & 7)) * 16;      // unit it: + 512 it; lo part: + 2 WIN

    const size_t simg = (size_t)a.H * a.W * 32 * sizeof(ST), cimg = (size_t)Hc * Wc * 64 * sizeof(ST);
    const auto rss = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.xs)) + (size_t)n * simg, 0, (int)simg, 0x00020000);
    const auto rsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.xc)) + (size_t)n * cimg, 0, (int)cimg, 0x00020000);
    u32x4 pvs[NUS][NL], pvc[NUC][NL];
    // prefetch of the units [u0, u1) of a tile (units 0-3 coarse, 4-9 skip); origins may lie one row / column outside the image:
    // unsigned wrap-around is fine, the affected units are padding
    auto prefetch = [&](int ptx, int pty, int u0, int u1) {
        const unsigned orgc = (unsigned)((((pty * 4 - 1) * Wc + (ptx * 16 - 1)) * 64) * (int)sizeof(ST));
        const unsigned orgs = (unsigned)((((pty * 8 - 1) * a.W + (ptx * 32 - 1)) * 32) * (int)sizeof(ST));
#pragma unroll
        for (int u = 0; u < NUC + NUS; ++u) {
            if (u < u0 || u >= u1) continue;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                if (u < NUC) pvc[u < NUC ? u : 0][l] = __builtin_amdgcn_raw_buffer_load_b128(rsc, orgc + relc[u < NUC ? u : 0] + 16 * l, 0, 0);
                else pvs[u >= NUC ? u - NUC : 0][l] = __builtin_amdgcn_raw_buffer_load_b128(rss, orgs + rels[u >= NUC ? u - NUC : 0] + 16 * l, 0, 0);
            }
        }
    };

    // ---- composed weights of this wave's parity: fragment (k-step s, part, cb) at wgl + s * 4096 + (part * 2 + cb) * 1024
    //      (buffer loads: lane offset in one VGPR, the fragment's offset as SGPR / immediate - flat addresses cost a VGPR pair per 4 KB)
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.wc0)) + (size_t)w * 32768, 0, 32768, 0x00020000);
    const unsigned wlane = (unsigned)lane * 16u;
    half8 ring[RING][NPP][2];
    auto wload = [&](int slot, int s) {
#pragma unroll
        for (int pt2 = NPP - 1; pt2 >= 0; --pt2)           // (the lo parts first: their product is issued first)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                ring[slot][pt2][cb] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsw, wlane, s * 4096 + (pt2 * 2 + cb) * 1024, 0));
    };
    prefetch(txi, tyi, 0, NUC + NUS);

    // ---- lane constants of the MFMA phases and the epilogue
    const int wbase = g * 512 + j * 16;                                                     // resident weight fragment
    const int cbase = WB + CO + (g >> 1) * kResPS + (g & 1) * kU0CP + (pA * 18 + j + pB) * 16;    // coarse fragment: + q * WIN + ((I + dI) * 18 + dJ) * 16
    const int sbase = WB + g * kResPS + j * 16;                                             // skip fragment: + tap term (wave-uniform) + part * 4 PS + 2 I * 34 * 16
    const float oscale = *a.oscale;
    float bv[2][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[cb][i] = a.bvar[4 * 32 + 8 * g + 4 * cb + i];      // (row 4 g + i of MFMA block cb = channel 8 g + 4 cb + i: pack_weights)
    const unsigned vst = (unsigned)(((2 * j + pB) * 32 + 8 * g) * (int)sizeof(ST));
    const _Float16 slope_h = (_Float16)a.slope;
    const unsigned slope2 = (unsigned)__builtin_bit_cast(unsigned short, slope_h) * 0x10001u;
    const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.dst) + (size_t)n * simg, 0, (int)simg, 0x00020000);
    const f32x4 slope4 = f32x4{a.slope, a.slope, a.slope, a.slope};
    // scale / shift of the image, one channel per lane (skip: lanes 0-31; coarse: all 64); a thread's 8 staging channels are fetched
    // with ds_bpermute at the start of each convert phase (16 + 16 values held per lane all tile long cost 28 more VGPRs than fit)
    const float nss = a.scs[(size_t)n * 32 + (lane & 31)], nts = a.shs[(size_t)n * 32 + (lane & 31)];
    const float nsc = a.scc[(size_t)n * 64 + lane], ntc = a.shc[(size_t)n * 64 + lane];
    auto gather8 = [&](float v, int first, f32x4& lo4, f32x4& hi4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo4[e] = __shfl(v, first + e); hi4[e] = __shfl(v, first + 4 + e); }
    };

    auto convert = [&](const u32x4 (&pv)[NL], const f32x4& na, const f32x4& nb, const f32x4& ta, const f32x4& tb, unsigned char* d, int lo_off) {
        if constexpr (sizeof(ST) == 4) {
            f32x4 va = __builtin_bit_cast(f32x4, pv[0]), vb = __builtin_bit_cast(f32x4, pv[NL - 1]);
            va = va * na + ta; vb = vb * nb + tb;
            const f32x4 ma = va * slope4, mb = vb * slope4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] = fmaxf(va[e], ma[e]); vb[e] = fmaxf(vb[e], mb[e]); }      // LeakyReLU (0 < slope < 1)
            uint4 hi, lo;
            split_hi_lo_8(va, vb, hi, lo);
            *reinterpret_cast<uint4*>(d) = hi;
            if (NPP == 2) *reinterpret_cast<uint4*>(d + lo_off) = lo;
        } else {
            const uint4 x = uint4{pv[0][0], pv[0][1], pv[0][2], pv[0][3]};
            *reinterpret_cast<uint4*>(d) = norm_lrelu_8(x, na, nb, ta, tb, slope2);
        }
    };

    TS2D_PROF_DECL(a.prof);
#define TS2D_STAMP(I) TS2D_STAMP_AT(a.prof, I)
    for (int t = t0; t < t1; ++t) {
        const unsigned edge = (tyi == 0 ? 1u : 0u) | (tyi == a.tiles_y - 1 ? 2u : 0u) | (txi == 0 ? 4u : 0u) | (txi == a.tiles_x - 1 ? 8u : 0u);
        // ============================================================ A: coarse patch -> LDS (norm + LeakyReLU + split on the fly)
        __builtin_amdgcn_sched_barrier(0);
                // (experiment switch kU0Rot: k-step order rotated by the tile's index inside its image)
        const int rot = kU0Rot ? ((t - n * tpi) & 7) : 0;
#pragma unroll
        for (int s = 0; s < RING; ++s) wload(s, (s + rot) & 7);                        // the first two k-steps' weights (behind the previous tile's stores in the vmcnt order)
        {
            f32x4 csa, csb, cta, ctb;
            gather8(nsc, 8 * cg, csa, csb); gather8(ntc, 8 * cg, cta, ctb);
#pragma unroll
            for (int it = 0; it < NUC; ++it)
                if (it < NUC - 1 || last_c) convert(pvc[it], csa, csb, cta, ctb, sP + lwc + it * 512, 2 * WIN);
        }
        if (edge) {                                        // wave-uniform: out-of-image coarse pixels are zero padding of `up`
            const unsigned hit = cmask & (edge * 0x1111u);
#pragma unroll
            for (int it = 0; it < NUC; ++it)
                if (hit & (0xFu << (4 * it))) {
                    unsigned char* d = sP + lwc + it * 512;
                    *reinterpret_cast<uint4*>(d) = uint4{0u, 0u, 0u, 0u};
                    if (NPP == 2) *reinterpret_cast<uint4*>(d + 2 * WIN) = uint4{0u, 0u, 0u, 0u};
                }
        }
        TS2D_STAMP(0)
        lds_barrier();
        TS2D_STAMP(1)

        // ============================================================ B: composed up half: 8 k-steps x (2 cb x 4 I) x NP products
        f32x4 acc[2][4];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int sr = (s + rot) & 7;                  // the k-step this iteration computes (rot = 0: s)
            const int koff = (sr >> 2) * WIN + (((sr >> 1) & 1) * 18 + (sr & 1)) * 16;      // (scalar) half ks, tap (dI, dJ)
            const unsigned char* xa = smem8 + cbase + koff;
            half8 fx[4][NPP];
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                for (int pt2 = 0; pt2 < NPP; ++pt2)
                    fx[pb][pt2] = *reinterpret_cast<const half8*>(xa + pt2 * 2 * WIN + pb * 18 * 16);
            if constexpr (NP == 3) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[s % RING][1][cb], fx[pb][0], acc[cb][pb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[s % RING][0][cb], fx[pb][1], acc[cb][pb], 0, 0, 0);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[s % RING][0][cb], fx[pb][0], acc[cb][pb], 0, 0, 0);
            if (s + RING < 8) {                                // (fenced: the scheduler otherwise sinks the loads next to their use two k-steps
                __builtin_amdgcn_sched_barrier(0);          //  later - a full L2 round trip per fragment in front of its MFMAs)
                wload(s % RING, (s + RING + rot) & 7);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        TS2D_STAMP(2)
        if constexpr (!TWO_BARRIERS) lds_barrier();        // every wave is done with the coarse planes (16-bit mode: they have their own region)
        TS2D_STAMP(3)

        // ============================================================ C: skip patch -> LDS
        {
            f32x4 ssa, ssb, sta, stb;
            gather8(nss, 8 * sg, ssa, ssb); gather8(nts, 8 * sg, sta, stb);
#pragma unroll
            for (int it = 0; it < NUS; ++it)
                if (it < NUS - 1 || last_s) convert(pvs[it], ssa, ssb, sta, stb, sP + lws[it], 4 * kResPS);
        }
        if (edge) {
            const unsigned hit = emask & (edge * 0x111111u);
#pragma unroll
            for (int it = 0; it < NUS; ++it)
                if (hit & (0xFu << (4 * it))) {
                    unsigned char* d = sP + lws[it];
                    *reinterpret_cast<uint4*>(d) = uint4{0u, 0u, 0u, 0u};
                    if (NPP == 2) *reinterpret_cast<uint4*>(d + 4 * kResPS) = uint4{0u, 0u, 0u, 0u};
                }
        }
        lds_barrier();

        // ---- next tile of the segment: its raw patches are requested during this phase (no global weight load follows until the next tile)
        int ntx = txi, nty = tyi + 1;
        if (nty == a.tiles_y) { nty = 0; ntx = txi + 1; }
        if (t + 1 >= t1) { ntx = txi; nty = tyi; }        // (last tile of the segment: the same patch again, never used - no branch around the loads)
        TS2D_STAMP(4)

        // ============================================================ D: skip half: 9 taps x (2 cb x 4 I) x NP products
        int qA = pA, qB = pB;                              // (opaque per tile: the nine tap addresses are one v_add each, not nine
        asm volatile("" : "+s"(qA), "+s"(qB));             //  loop-invariant VGPRs carried through every phase)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            // patch pixel (2I + A + ky, 2J + B + kx): row 2I + A + ky, half (B + kx) & 1, index J + ((B + kx) >> 1)
            const int toff = ((qA + ky) * kResPW + ((qB + kx) & 1) * 17 + ((qB + kx) >> 1)) * 16;      // wave-uniform
            half8 fw[2][NPP], fx[4][NPP];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pt2 = 0; pt2 < NPP; ++pt2)
                    fw[cb][pt2] = *reinterpret_cast<const half8*>(smem8 + wbase + ((tap * NPP + pt2) * 4) * 512 + cb * 256);
            const unsigned char* xb = smem8 + sbase + toff;
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                for (int pt2 = 0; pt2 < NPP; ++pt2)
                    fx[pb][pt2] = *reinterpret_cast<const half8*>(xb + pt2 * 4 * kResPS + 2 * pb * kResPW * 16);
            if constexpr (NP == 3) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[cb][1], fx[pb][0], acc[cb][pb], 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[cb][0], fx[pb][1], acc[cb][pb], 0, 0, 0);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[cb][0], fx[pb][0], acc[cb][pb], 0, 0, 0);
            // the next tile's raw patches, a unit behind every tap: a burst of 20 loads blocks the wave at their ISSUE for as long as the
            // CU's memory path needs to take them (in-kernel stamps: 12 000 cycles with the MFMA pipe idle)
            if (kU0Burst) { if (tap == 0) prefetch(ntx, nty, 0, 10); }
            else { prefetch(ntx, nty, tap, tap + 1); if (tap == 8) prefetch(ntx, nty, 9, 10); }
        }
        __builtin_amdgcn_s_setprio(0);
        TS2D_STAMP(5)

        // ---- epilogue: lane = pixel (2 pb + A, 2 j + B) of the tile, channels 16 cb + 4 g .. + 3.  TWO copies of the code: a tile on the
        //      image border loads its bias variants (the transposed conv's bias reaches a border pixel through fewer taps); with one copy
        //      the wait for those loads - vmcnt(0): the next tile's patches, the previous stores - sits in front of EVERY store of EVERY tile
        auto epilogue = [&](auto border) {
            constexpr bool BORDER = decltype(border)::value;
            float ss[2][4], qq[2][4], kv[2][4];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ss[cb][i] = 0.f; qq[cb][i] = 0.f;
                    // pivot of the shifted statistics (kernels.h): any finite value near the data.  DIRECT: the interior bias; else a value of the tile
                    if constexpr (DIRECT) kv[cb][i] = bv[cb][i];
                    else kv[cb][i] = __shfl(__builtin_fmaf(acc[cb][0][i], oscale, bv[cb][i]), lane & 48);
                }
            const unsigned tile_off = (unsigned)((((tyi * 8 + pA) * a.W + txi * 32) * 32) * (int)sizeof(ST));      // scalar
            f32x4 b4[4][2];
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    b4[pb][cb] = f32x4{bv[cb][0], bv[cb][1], bv[cb][2], bv[cb][3]};
                    if constexpr (BORDER) {
                        const int X = txi * 32 + 2 * j + pB, Y = tyi * 8 + 2 * pb + pA;
                        const int rx = X == 0 ? 0 : (X == a.W - 1 ? 2 : 1), ry = Y == 0 ? 0 : (Y == a.H - 1 ? 2 : 1);
                        b4[pb][cb] = *reinterpret_cast<const f32x4*>(a.bvar + (ry * 3 + rx) * 32 + 8 * g + 4 * cb);
                    }
                }
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const unsigned soff = tile_off + (unsigned)(((2 * pb * a.W) * 32) * (int)sizeof(ST));
                f32x4 v[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[cb][i] = __builtin_fmaf(acc[cb][pb][i], oscale, b4[pb][cb][i]);
                // wide-store hazard of gfx950 (found with the round-2 form of conv3x3_res32): a VALU write to the data registers of a 128-bit
                // buffer store two instructions behind it reaches the stored data (last dword, lanes 12-15 of each lane row) - hipcc's one
                // wait state is not enough, and with an SGPR soffset it inserts none.  The offset rides in the VGPR and wait states follow the store
                if constexpr (sizeof(ST) == 4) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[0]), rsd, vst + soff, 0, kU0NT);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[1]), rsd, vst + soff + 16, 0, kU0NT);
                    asm volatile("s_nop 3" :: "v"(v[0]), "v"(v[1]) : "memory");      // (v as operands: their registers stay allocated up to the wait states)
                } else {
                    half8 hv;
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int i = 0; i < 4; ++i) { hv[4 * cb + i] = (_Float16)v[cb][i]; v[cb][i] = (float)hv[4 * cb + i]; }
                    const u32x4 hq = __builtin_bit_cast(u32x4, hv);
                    __builtin_amdgcn_raw_buffer_store_b128(hq, rsd, vst + soff, 0, kU0NT);      // the lane's 8 channels of this pixel: 16 bytes
                    asm volatile("s_nop 3" :: "v"(hq) : "memory");
                }
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float d = v[cb][i] - kv[cb][i]; ss[cb][i] += d; qq[cb][i] = __builtin_fmaf(d, d, qq[cb][i]); }
            }
            // the wave's partial per channel: 16 lanes (j) -> lane j = 0 of each k-group row, then straight to global memory
            float* pw = a.part + ((((size_t)n * tpi + (size_t)(t - n * tpi)) * 4 + w) * 32 + 8 * g) * 4;      // (DIRECT)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float s = ss[cb][i], q = qq[cb][i];
#define TS2D_ROR_ADD(X_, N) X_ += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, X_), 0x120 + N, 0xF, 0xF, true))
                    TS2D_ROR_ADD(s, 8); TS2D_ROR_ADD(q, 8); TS2D_ROR_ADD(s, 4); TS2D_ROR_ADD(q, 4);
                    TS2D_ROR_ADD(s, 2); TS2D_ROR_ADD(q, 2); TS2D_ROR_ADD(s, 1); TS2D_ROR_ADD(q, 1);
#undef TS2D_ROR_ADD
                    ss[cb][i] = s; qq[cb][i] = q;
                }
            }
            if (j == 0) {
                if constexpr (DIRECT) {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 pq = f32x4{ss[cb][i], qq[cb][i], kv[cb][i], 64.f};
                            *reinterpret_cast<f32x4*>(pw + (4 * cb + i) * 4) = pq;
                            asm volatile("s_nop 3" :: "v"(pq) : "memory");
                        }
                } else {
                    float* sc4 = scratch(w, g);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        *reinterpret_cast<f32x4*>(sc4 + cb * 12) = f32x4{ss[cb][0], ss[cb][1], ss[cb][2], ss[cb][3]};
                        *reinterpret_cast<f32x4*>(sc4 + cb * 12 + 4) = f32x4{qq[cb][0], qq[cb][1], qq[cb][2], qq[cb][3]};
                        *reinterpret_cast<f32x4*>(sc4 + cb * 12 + 8) = f32x4{kv[cb][0], kv[cb][1], kv[cb][2], kv[cb][3]};
                    }
                }
            }
        };
        if (edge) epilogue(std::true_type{}); else epilogue(std::false_type{});
        if constexpr (!TWO_BARRIERS) lds_barrier();        // every wave is done with the skip planes (16-bit mode: the next conversion writes the coarse region)
        if constexpr (!DIRECT) {
            if (tid < 32) {                                // tile partial: fixed order over the 4 waves, rebased onto wave 0's pivot; channel co = 8 gg + 4 cb + i
                const int co = tid, gg = co >> 3, cb = (co >> 2) & 1, i = co & 3;
                const float* s0 = scratch(0, gg) + cb * 12 + i;
                f32x4 acc4 = f32x4{s0[0], s0[4], s0[8], 64.f};
#pragma unroll
                for (int ww = 1; ww < 4; ++ww) {
                    const float* sw_ = scratch(ww, gg) + cb * 12 + i;
                    const float d = sw_[8] - acc4[2];
                    acc4[1] += sw_[4] + d * (2.f * sw_[0] + 64.f * d);
                    acc4[0] += sw_[0] + 64.f * d;
                    acc4[3] += 64.f;
                }
                *reinterpret_cast<f32x4*>(a.part + (((size_t)n * tpi + (t - n * tpi)) * 32 + co) * 4) = acc4;
            }
        }
        txi = ntx; tyi = nty;
        TS2D_STAMP(6)
    }
    TS2D_PROF_FLUSH(a.prof)
#undef TS2D_STAMP
}

}  // namespace ts2d
