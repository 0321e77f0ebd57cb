// Level-0 block: Conv2d 3x3 stride 1, 32 -> 32 channels (enc0.c1 / dec0.c1 of the canonical net, 512x512 pixels per
// slice - the layers that sit between the HBM and the MFMA roof), as a PERSISTENT kernel with the whole layer's weights RESIDENT
// in LDS (32 x 32 x 9 x (hi + lo) fp16 = 36 KB):
//   * v_mfma_f32_32x32x16_f16, D[pixel][cout]: a wave owns 2 rows of 32 pixels x the 32 output channels (two k-steps per tap, 108
//     MFMAs per tile in split mode), epilogue as conv3x3_f16x3_qp: lane = channel, 4-byte stores (32 lanes = one pixel's whole
//     128-byte record), statistics summed in registers with one cross-half add.
//     (Rounds 2-3 issued v_mfma_f32_16x16x32_f16 TRANSPOSED, D[cout][pixel], for 16-byte stores: that shape holds the SIMD's vector
//      issue for 8 of its 16 cycles - the partner workgroup's conversion / epilogue ran at 8.4 cycles per VALU instruction - and its
//      epilogue needed 64 DPP adds per tile for the 16-lane statistics reduction.  Same-process A/B, round 4
//      (profiles/r04_experiments.txt): 1.16 -> 1.07 ms per launch in split mode, whole step -0.15 ms; equal in the 16-bit mode.)
//   * one staging phase and two barriers per 256-pixel tile (the generic kernel: two 16-channel chunks, each with its own weight
//     staging and barrier pair); no weight traffic after the first tile of a workgroup.
//   * the next tile's raw patch is prefetched into registers across the tile boundary (behind the MFMA phase and the epilogue).
//   * a workgroup owns a SEGMENT of consecutive tiles of one image (column by column: consecutive tiles share their halo
//     rows through L2).
//   * LDS images are "k-group major": plane[g][pixel] of 16-byte slots (g = the 8-channel group (k-step, lane half) a lane feeds to
//     the MFMA), plane stride a multiple of 256 B: the 32 lanes of a lane half read 32 consecutive slots - conflict-free
//     ds_read_b128 at any alignment - and every fragment address is lane base + immediate.
// Arithmetic: split mode (NP = 3): x = hi + lo, w = whi + wlo, products xlo*whi + xhi*wlo + xhi*whi, fp32 accumulation over the
// 9 taps x 32 channels of the layer (288-term chains; the generic kernel sums two 144-term chunks).  f16 mode (NP = 1): one product.
#pragma once
#include "kernels_f16x3.h"
#include "kernels_h32.h"

namespace ts2d {

struct Res32Args {
    const void* src; const float* sc; const float* sh;    // NHWC [B,H,W,32] activations of storage type ST; per-(n,c) scale / shift
    const void* wres;      // resident weight image [tap 9][part hi,lo][g 4][cout 32][8 halves] (fp16; the f16 mode reads the hi parts)
    const float* bias;     // [32]
    const float* oscale;   // 1 / (power-of-two weight pre-scale)
    void* dst;             // raw NHWC [B,H,W,32] output (ST)
    float* part;           // InstanceNorm partials [n][tile (column-major index inside the image)][32] x (S, Q, K, n) - kernels.h
    int B, H, W;           // H % 8 == 0, W % 32 == 0
    int tiles_x, tiles_y;  // W / 32, H / 8
    int n_tiles;           // B * tiles_x * tiles_y
    int seg;               // tiles per workgroup segment; divides tiles_x * tiles_y (a segment never leaves its image)
                           // (the statistics stay per TILE: a slice computes bit-identically alone or in a batch)
    float slope;
    unsigned long long* prof;   // diagnostic (TS2D_DBG=256): cycles of wave 0 in [0] patch conversion, [1] barrier, [2] MFMAs, [3] epilogue, [4] barrier + tile partial
    // FUSE variant (round 4): `src` is never read - the producing block (the network's FIRST block: C0 <= 2 input channels -> 32, exact
    // fp32 MFMA as conv3x3_first computes it) is recomputed per tile from the NCHW network input; sc / sh are ITS scale / shift, known
    // from a statistics-only pass of conv3x3_first.
    const float* x0; int C0;           // network input [B, C0, H, W] fp32
    const float* w0; const float* b0;  // first block: weights [32][C0][3][3] (PyTorch layout), bias [32]
};

constexpr int kResPW = 34, kResP = 340, kResPS = 352 * 16;      // patch 10 x 34 pixels; plane stride (bytes)

// FUSE (round 4): the block in front (enc0.c0: 2 -> 32 channels at full resolution) is the one tensor of the net that is
// written once (33.5 MB per slice) only to be read back by this kernel.  With FUSE the kernel reads the 2-channel network input
// instead (12 x 36 input pixels per tile, straight from L1 / L2 into the B operand of v_mfma_f32_32x32x2_f32: 27 dword loads per lane,
// prefetched across the tile boundary like the patch before), recomputes enc0.c0 for the 340 patch pixels as 11 M tiles of 32 pixels
// (TRANSPOSED product: rows = channels, so that a lane holds 4 consecutive channels of one pixel = half a 16-byte LDS slot),
// normalises with the scale / shift of a statistics-only pass, applies LeakyReLU, zeroes what lies outside the image (the padding of
// THIS conv), splits hi / lo and writes the same k-group-major planes the MFMA phase reads.  Cost per tile and wave: 27 fp32 MFMAs of
// 64 cycles beside the 108 (x 32 cycles) of the block itself; saved: 4.3 GB of HBM traffic per 64-slice step.
template <typename ST, int NP, bool FUSE = false>
__global__ __launch_bounds__(kBlock, 2) void conv3x3_res32(const Res32Args a) {
    constexpr int NPP = NP == 3 ? 2 : 1;                   // fp16 parts per value (hi, lo)
    constexpr int WB = 9 * NPP * 4 * 512;                  // resident weights (bytes)
    constexpr int NL = sizeof(ST) == 4 ? 2 : 1;            // 16-byte loads per staging unit (8 channels)
    constexpr int NU = 6;                                  // staging units per thread: 6 x 64 pixels >= 340
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;                // MFMA lane roles: row / column index, k half

    // ---- work list: segment s = tiles [s * seg, (s + 1) * seg), tiles numbered column by column inside an image
    //      (t = (n * tiles_x + txi) * tiles_y + tyi); seg divides tiles_x * tiles_y, so a segment lies in one image.
    const int tpi = a.tiles_x * a.tiles_y;
    const int t0 = (int)blockIdx.x * a.seg;
    if (t0 >= a.n_tiles) return;
    const int t1 = t0 + a.seg;
    const int n = t0 / tpi;                                // the segment's image
    int trem = t0 - n * tpi;
    int txi = trem / a.tiles_y, tyi = trem - txi * a.tiles_y;

    // ---- resident weights: one linear copy of the pre-arranged image (once per workgroup)
    {
        const uint4* wsrc = reinterpret_cast<const uint4*>(a.wres);
#pragma unroll
        for (int k = 0; k < (WB / 16 + kBlock - 1) / kBlock; ++k) {       // (f16 mode: the hi parts only - every other 2-KB block of the image)
            const int sl = tid + k * kBlock;
            if (sl < WB / 16) *reinterpret_cast<uint4*>(smem8 + sl * 16) = wsrc[NPP == 2 ? sl : (sl >> 7) * 256 + (sl & 127)];
        }
    }
    unsigned char* sP = smem8 + WB;                        // patch planes [part][g][pixel] x 16 B
    // ---- staging plan.  Unit (it): pixel p = 64 it + 16 w + (lane & 7) + 8 (lane >> 5), channel group sg = (lane >> 3) & 3:
    //      one wave instruction covers 16 whole pixels (every byte of their 128-byte records), 8 consecutive lanes write 8
    //      consecutive 16-byte slots of one plane (conflict-free ds_write_b128).
    const int sg = (lane >> 3) & 3, pl = (lane & 7) + 8 * (lane >> 5);
    // cross-wave statistics scratch: per wave and 16-channel half [3 = S, Q, K][16 channels] floats = 192 bytes = the 12 unused slots at the
    // end of ONE patch plane (split mode: the two workgroups of a CU use all 160 KiB; wave ww: planes 2 ww, 2 ww + 1), or behind the planes (f16 mode)
    auto scratch32 = [&](int ww, int half) -> float* {
        if (NPP == 2) return reinterpret_cast<float*>(sP + (2 * ww + half) * kResPS + kResP * 16);
        return reinterpret_cast<float*>(sP + NPP * 4 * kResPS + (ww * 2 + half) * 192);
    };
    unsigned rel[FUSE ? 1 : NU];          // byte offset of the unit's 8 channels relative to the patch origin (ty0 - 1, tx0 - 1)
    unsigned emask = 0;        // per unit 4 bits: patch row 0 / row 9 / column 0 / column 33 (the padding candidates)
    bool last_unit = false;
    const int swr = sg * kResPS + (16 * w + pl) * 16;      // LDS write address of unit 0 (hi part); unit it: + 1024 it
    const size_t img_bytes = (size_t)a.H * a.W * 32 * sizeof(ST);
    u32x4 pv[FUSE ? 1 : NU][NL];
    // FUSE: M tile k of this wave = patch pixels p = 32 (w + 4 k) + r2, k = 0..2 (11 M tiles: wave 3 has two; pixels >= 340 are not written)
    const int r2 = lane & 31, h2 = lane >> 5;              // lane roles in the 32x32x2 fp32 MFMA: column (pixel) / k (input channel)
    constexpr int NK = 3;
    unsigned xrel[FUSE ? NK : 1], pq[FUSE ? NK : 1];       // input byte offset of (py, px) of channel h2 relative to the input-patch origin; (py << 8) | px
    float xv[FUSE ? NK : 1][FUSE ? 9 : 1];                 // the 27 prefetched input values
    float wr0[FUSE ? 9 : 1], scq[FUSE ? 16 : 1], shq[FUSE ? 16 : 1];
    float bq[(FUSE && sizeof(ST) != 4) ? 16 : 1];
    // Round 6: the recompute on the fp16 matrix path (as conv3x3_first_split computes the statistics): lane half h2 = input channel, its 8 K entries of
    // the four v_mfma_f32_32x32x16_f16 are  m0: hi, lo of taps 0-3 | m1: hi, lo of taps 4-7 | m2: hi, lo, hi of tap 8 | m3: hi of taps 0-7  against the
    // weights  whi whi | whi whi | whi whi wlo | wlo  (x_hi w_hi + x_lo w_hi + x_hi w_lo; w scaled by the power of two S that puts max|w| into [2^11, 2^12)).
    // 128 instead of 576 matrix-pipe cycles per M tile and ~ 20 VALU to split the lane's nine values.  A wave whose patch holds |x| >= 65 504 (fp16's
    // range; no pre-scale here) takes the exact fp32 MFMAs with the SAME scaled weights (a power of two: exact), so both paths share scq.
    typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
    u32x4w wsp[FUSE ? 4 : 1];
    float inv_s0 = 1.f;
    const bool have3 = w != 3;                             // (wave-uniform)
    __amdgpu_buffer_rsrc_t rs, rsx;
    if constexpr (!FUSE) {
#pragma unroll
        for (int it = 0; it < NU; ++it) {
            const int p = 64 * it + 16 * w + pl;
            const int py = p / kResPW, px = p - py * kResPW;
            rel[it] = (unsigned)(((py * a.W + px) * 32 + 8 * sg) * (int)sizeof(ST));
            if (p < kResP) emask |= ((py == 0 ? 1u : 0u) | (py == 9 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == kResPW - 1 ? 8u : 0u)) << (4 * it);
        }
        last_unit = 64 * (NU - 1) + 16 * w + pl < kResP;          // unit NU-1 exists for this thread
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.src)) + (size_t)n * img_bytes,
                                               0, (int)img_bytes, 0x00020000);
    } else {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int p = 32 * (w + 4 * k) + r2;
            const int py = p / kResPW, px = p - py * kResPW;
            xrel[k] = (unsigned)(((h2 * a.H + py) * a.W + px) * 4);
            pq[k] = (unsigned)((py << 8) | px);
            if (p < kResP) emask |= ((py == 0 ? 1u : 0u) | (py == 9 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == kResPW - 1 ? 8u : 0u)) << (4 * k);
        }
        const size_t in_bytes = (size_t)a.C0 * a.H * a.W * 4;
        rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x0) + (size_t)n * a.C0 * a.H * a.W, 0, (int)in_bytes, 0x00020000);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) wr0[tap] = h2 < a.C0 ? a.w0[((size_t)r2 * a.C0 + h2) * 9 + tap] : 0.f;      // A[m = channel r2][k = input channel h2]
        {
            float wmax = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wmax = fmaxf(wmax, fabsf(wr0[tap]));
#pragma unroll
            for (int d = 1; d < 64; d *= 2) wmax = fmaxf(wmax, __shfl_xor(wmax, d));
            int e2 = 0;
            (void)frexpf(wmax, &e2);
            const bool sane = wmax > 0.f && wmax < 3.0e38f;
            const float S0 = sane ? ldexpf(1.f, 12 - e2) : 1.f;
            inv_s0 = sane ? ldexpf(1.f, e2 - 12) : 1.f;
            unsigned short wh[9], wl[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                wr0[tap] *= S0;                                                       // (exact: a power of two)
                const _Float16 hh = (_Float16)wr0[tap], ll = (_Float16)(wr0[tap] - (float)hh);
                wh[tap] = __builtin_bit_cast(unsigned short, hh); wl[tap] = __builtin_bit_cast(unsigned short, ll);
            }
            auto pk = [](unsigned short lo_, unsigned short hi_) { return (unsigned)lo_ | ((unsigned)hi_ << 16); };
            wsp[0] = u32x4w{pk(wh[0], wh[1]), pk(wh[2], wh[3]), pk(wh[0], wh[1]), pk(wh[2], wh[3])};
            wsp[1] = u32x4w{pk(wh[4], wh[5]), pk(wh[6], wh[7]), pk(wh[4], wh[5]), pk(wh[6], wh[7])};
            wsp[2] = u32x4w{pk(wh[8], wh[8]), pk(wl[8], 0), 0u, 0u};
            wsp[3] = u32x4w{pk(wl[0], wl[1]), pk(wl[2], wl[3]), pk(wl[4], wl[5]), pk(wl[6], wl[7])};
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {                     // rows of the 32x32 product held by this lane: channel (i & 3) + 8 (i >> 2) + 4 h2
            const int c = (i & 3) + 8 * (i >> 2) + 4 * h2;
            const float s_ = a.sc[(size_t)n * 32 + c], t_ = a.sh[(size_t)n * 32 + c], b_ = a.b0[c];
            if constexpr (sizeof(ST) == 4) { scq[i] = s_ * inv_s0; shq[i] = __builtin_fmaf(b_, s_, t_); }      // (acc / S + b) s + t = acc (s / S) + (b s + t)
            else { scq[i] = s_; shq[i] = t_; bq[i] = b_; }                          // 16-bit mode: the raw value is rounded to fp16 first, as if stored
        }
    }
    auto prefetch = [&](int ptx, int pty) {
        if constexpr (!FUSE) {
            // origin may lie one row / column outside the image: unsigned wrap-around is fine, the affected units are padding
            const unsigned org = (unsigned)((((pty * 8 - 1) * a.W + (ptx * 32 - 1)) * 32) * (int)sizeof(ST));
#pragma unroll
            for (int it = 0; it < NU; ++it)
#pragma unroll
                for (int l = 0; l < NL; ++l)
                    pv[it][l] = __builtin_amdgcn_raw_buffer_load_b128(rs, org + rel[it] + 16 * l, 0, 0);
        } else {
            // input patch origin (ty0 - 2, tx0 - 2).  Interior tiles: every address lies inside the image (a lane with h2 >= C0 is beyond
            // the buffer's range and reads 0).  Border tiles: the first conv's own zero padding, per load.
            const int oy0 = pty * 8 - 2, ox0 = ptx * 32 - 2;
            const bool pe = (pty == 0) | (pty == a.tiles_y - 1) | (ptx == 0) | (ptx == a.tiles_x - 1);      // wave-uniform
            const unsigned org = (unsigned)((oy0 * a.W + ox0) * 4);
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                if (k < NK - 1 || have3) {
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int dy = tap / 3, dx = tap - 3 * dy;
                        // (the buffer's range check looks at the VGPR offset alone: on a border tile the origin can be "negative", so the
                        //  tap offset goes into the VGPR there; interior tiles keep it in the scalar offset)
                        unsigned vo = org + xrel[k];
                        unsigned so = (unsigned)((dy * a.W + dx) * 4);
                        if (pe) {
                            const int iy = oy0 + (int)(pq[k] >> 8) + dy, ix = ox0 + (int)(pq[k] & 255u) + dx;
                            const bool ok = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W) & (h2 < a.C0);
                            vo = ok ? vo + so : 0x80000000u;
                            so = 0;
                        }
                        xv[k][tap] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, vo, so, 0));
                    }
                }
            }
        }
    };
    prefetch(txi, tyi);

    // ---- lane constants of the MFMA phase and the epilogue
    const int wb32 = h * 512 + r * 16;                                     // weight fragment: + ((tap * NPP + part) * 4 + 2 ks) * 512
    const int pb32 = WB + h * kResPS + (2 * w * kResPW + r) * 16;          // patch fragment: + (part * 4 + 2 ks) * PS + ((row + dy) * 34 + dx) * 16
    const float bv32 = a.bias[r];
    const float oscale = *a.oscale;
    const _Float16 slope_h = (_Float16)a.slope;
    const unsigned slope2 = (unsigned)__builtin_bit_cast(unsigned short, slope_h) * 0x10001u;
    const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.dst) + (size_t)n * img_bytes, 0, (int)img_bytes, 0x00020000);
    // scale / shift of this thread's 8 staging channels (image n)
    const float* ps = a.sc + (size_t)n * 32 + 8 * sg; const float* pt = a.sh + (size_t)n * 32 + 8 * sg;
    f32x4 nsa, nsb, nta, ntb;
    if constexpr (!FUSE) {
        nsa = *reinterpret_cast<const f32x4*>(ps); nsb = *reinterpret_cast<const f32x4*>(ps + 4);
        nta = *reinterpret_cast<const f32x4*>(pt); ntb = *reinterpret_cast<const f32x4*>(pt + 4);
    }
    const f32x4 slope4 = f32x4{a.slope, a.slope, a.slope, a.slope};

    TS2D_PROF_DECL(a.prof);
    for (int t = t0; t < t1; ++t) {
        // tiles on the image border: the patch rows / columns outside the image are zero padding (AFTER norm + activation)
        const unsigned edge = (tyi == 0 ? 1u : 0u) | (tyi == a.tiles_y - 1 ? 2u : 0u) | (txi == 0 ? 4u : 0u) | (txi == a.tiles_x - 1 ? 8u : 0u);
        if constexpr (!FUSE) {
        // ---- patch: InstanceNorm + LeakyReLU on the fly, split into fp16 hi / lo, written k-group major
#pragma unroll
        for (int it = 0; it < NU; ++it) {
            if (it < NU - 1 || last_unit) {
                unsigned char* d = sP + swr + it * 1024;
                if constexpr (sizeof(ST) == 4) {
                    f32x4 va = __builtin_bit_cast(f32x4, pv[it][0]), vb = __builtin_bit_cast(f32x4, pv[it][NL - 1]);
                    va = va * nsa + nta; vb = vb * nsb + ntb;
                    const f32x4 na = va * slope4, nb2 = vb * slope4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { va[e] = fmaxf(va[e], na[e]); vb[e] = fmaxf(vb[e], nb2[e]); }      // LeakyReLU (0 < slope < 1)
                    uint4 hi, lo;
                    split_hi_lo_8(va, vb, hi, lo);
                    *reinterpret_cast<uint4*>(d) = hi;
                    if (NPP == 2) *reinterpret_cast<uint4*>(d + 4 * kResPS) = lo;
                } else {
                    uint4 x = uint4{pv[it][0][0], pv[it][0][1], pv[it][0][2], pv[it][0][3]};
                    *reinterpret_cast<uint4*>(d) = norm_lrelu_8(x, nsa, nsb, nta, ntb, slope2);
                }
            }
        }
        if (edge) {                                        // wave-uniform; 15 % of the tiles of a 512 x 512 image
            const unsigned hit = emask & (edge * 0x111111u);
#pragma unroll
            for (int it = 0; it < NU; ++it)
                if (hit & (0xFu << (4 * it))) {
                    unsigned char* d = sP + swr + it * 1024;
                    *reinterpret_cast<uint4*>(d) = uint4{0u, 0u, 0u, 0u};
                    if (NPP == 2) *reinterpret_cast<uint4*>(d + 4 * kResPS) = uint4{0u, 0u, 0u, 0u};
                }
        }
        } else {
        // ---- FUSE: recompute the first block on the patch pixels of this wave's M tiles, normalise, activate, split, write the planes
        const unsigned hit = edge ? (emask & (edge * 0x111u)) : 0u;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (k < NK - 1 || have3) {
                f32x16 acc0 = kZero16;
                float amax = 0.f;
#pragma unroll
                for (int tap = 0; tap < 9; tap += 3) amax = fmaxf(amax, fmaxf(fabsf(xv[k][tap]), fmaxf(fabsf(xv[k][tap + 1]), fabsf(xv[k][tap + 2]))));
                if (__builtin_amdgcn_ballot_w64(!(amax < 65504.f)) == 0) {          // (wave-uniform) every value splits into two finite fp16 parts
                    uint4 xh, xl;
                    split_hi_lo_8(f32x4{xv[k][0], xv[k][1], xv[k][2], xv[k][3]}, f32x4{xv[k][4], xv[k][5], xv[k][6], xv[k][7]}, xh, xl);
                    const _Float16 h8 = (_Float16)xv[k][8], l8 = (_Float16)(xv[k][8] - (float)h8);
                    const unsigned uh8 = __builtin_bit_cast(unsigned short, h8), ul8 = __builtin_bit_cast(unsigned short, l8);
                    const u32x4w xb0 = u32x4w{xh.x, xh.y, xl.x, xl.y}, xb1 = u32x4w{xh.z, xh.w, xl.z, xl.w};
                    const u32x4w xb2 = u32x4w{uh8 | (ul8 << 16), uh8, 0u, 0u}, xb3 = u32x4w{xh.x, xh.y, xh.z, xh.w};
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, wsp[0]), __builtin_bit_cast(half8, xb0), acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, wsp[1]), __builtin_bit_cast(half8, xb1), acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, wsp[2]), __builtin_bit_cast(half8, xb2), acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, wsp[3]), __builtin_bit_cast(half8, xb3), acc0, 0, 0, 0);
                } else {
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wr0[tap], xv[k][tap], acc0, 0, 0, 0);
                }
                const bool zero = (hit >> (4 * k)) & 0xFu;                        // this lane's pixel lies outside the image
                const int p = 32 * (w + 4 * k) + r2;
                unsigned char* d = sP + p * 16 + 8 * h2;                          // plane g = quad index: + g * kResPS; lo part: + 4 * kResPS
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                if constexpr (sizeof(ST) == 4) {
                    float v[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const float y = __builtin_fmaf(acc0[i], scq[i], shq[i]); v[i] = fmaxf(y, y * a.slope); }
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        uint4 hi, lo;
                        split_hi_lo_8(f32x4{v[8 * qq], v[8 * qq + 1], v[8 * qq + 2], v[8 * qq + 3]},
                                      f32x4{v[8 * qq + 4], v[8 * qq + 5], v[8 * qq + 6], v[8 * qq + 7]}, hi, lo);
                        if (zero) { hi = uint4{0u, 0u, 0u, 0u}; lo = hi; }
                        if (k < NK - 1 || p < kResP) {
                            *reinterpret_cast<u32x2*>(d + (2 * qq) * kResPS) = u32x2{hi.x, hi.y};
                            *reinterpret_cast<u32x2*>(d + (2 * qq + 1) * kResPS) = u32x2{hi.z, hi.w};
                            *reinterpret_cast<u32x2*>(d + (2 * qq + 4) * kResPS) = u32x2{lo.x, lo.y};
                            *reinterpret_cast<u32x2*>(d + (2 * qq + 5) * kResPS) = u32x2{lo.z, lo.w};
                        }
                    }
                } else {
                    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        half4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = 4 * q + e;
                            const float raw = (float)(_Float16)__builtin_fmaf(acc0[i], inv_s0, bq[i]);    // as stored by conv3x3_first in this mode
                            const _Float16 y = (_Float16)__builtin_fmaf(raw, scq[i], shq[i]);         // norm_lrelu_8: fp32 FMA, one rounding,
                            const _Float16 ng = y * slope_h;                                          // LeakyReLU in fp16
                            o[e] = zero ? (_Float16)0.f : (y > ng ? y : ng);
                        }
                        if (k < NK - 1 || p < kResP) *reinterpret_cast<half4*>(d + q * kResPS) = o;
                    }
                }
            }
        }
        }
        TS2D_STAMP_AT(a.prof, 0)
        lds_barrier();                                     // patch (and, first tile, the weights) visible to every wave
        TS2D_STAMP_AT(a.prof, 1)

        // ---- next tile of the segment (one row down, or the top of the next column): prefetch its raw patch behind the MFMAs
        int ntx = txi, nty = tyi + 1;
        if (nty == a.tiles_y) { nty = 0; ntx = txi + 1; }
        if (t + 1 < t1) prefetch(ntx, nty);

        // ---- 9 taps x 2 k-steps x 2 rows x NP products (32 pixels x 32 channels x 16)
        f32x16 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
        __builtin_amdgcn_s_setprio(1);
        // the fragments of tap t + 1 are read while the MFMAs of tap t run: two register sets, one scheduling region per tap (left to
        // itself hipcc put every ds_read_b128 pair directly in front of the MFMA that consumes it)
        half8 fw[2][2][NPP], fx[2][2][2][NPP];                                // [set][k-step][part], [set][row][k-step][part]
#define TS2D_R32_LOAD(SET, TAP) { constexpr int dy_ = (TAP) / 3, dx_ = (TAP) - 3 * dy_; \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int pt2 = 0; pt2 < NPP; ++pt2) { \
                fw[SET][ks][pt2] = *reinterpret_cast<const half8*>(smem8 + wb32 + (((TAP) * NPP + pt2) * 4 + 2 * ks) * 512); \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) \
                    fx[SET][mt][ks][pt2] = *reinterpret_cast<const half8*>(smem8 + pb32 + (pt2 * 4 + 2 * ks) * kResPS + ((mt + dy_) * kResPW + dx_) * 16); } }
#define TS2D_R32_TAP(TAP) { constexpr int c_ = (TAP) & 1; \
            if constexpr ((TAP) + 1 < 9) TS2D_R32_LOAD(c_ ^ 1, (TAP) + 1) \
            if constexpr (NP == 3) { \
                _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) \
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fx[c_][mt][ks][NPP - 1], fw[c_][ks][0], acc[mt], 0, 0, 0); \
                _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) \
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fx[c_][mt][ks][0], fw[c_][ks][NPP - 1], acc[mt], 0, 0, 0); } \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) \
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fx[c_][mt][ks][0], fw[c_][ks][0], acc[mt], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); }
        TS2D_R32_LOAD(0, 0)
        TS2D_R32_TAP(0) TS2D_R32_TAP(1) TS2D_R32_TAP(2) TS2D_R32_TAP(3) TS2D_R32_TAP(4) TS2D_R32_TAP(5) TS2D_R32_TAP(6) TS2D_R32_TAP(7) TS2D_R32_TAP(8)
#undef TS2D_R32_TAP
#undef TS2D_R32_LOAD
        __builtin_amdgcn_s_setprio(0);
        TS2D_STAMP_AT(a.prof, 2)

        // ---- epilogue: C/D map of the 32x32 MFMA: column = lane & 31 (channel), row = (i & 3) + 8 (i >> 2) + 4 h (pixel of the row)
        {
            const float kv = stat_pivot(round_act<ST>(__builtin_fmaf(acc[0][0], oscale, bv32)));      // shifted statistics (kernels.h)
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const unsigned voff = (unsigned)(((((tyi * 8 + 2 * w + mt) * a.W + txi * 32 + 4 * h) * 32) + r) * (int)sizeof(ST));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = __builtin_fmaf(acc[mt][i], oscale, bv32);
                    buffer_store_act<ST>(v, rsd, voff + (unsigned)((((i & 3) + 8 * (i >> 2)) * 32) * (int)sizeof(ST)), 0);
                    const float d = round_act<ST>(v) - kv;                          // statistics of what is stored
                    s += d; q = __builtin_fmaf(d, d, q);
                }
            }
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
            if (h == 0) { float* sc = scratch32(w, r >> 4) + (r & 15); sc[0] = s; sc[16] = q; sc[32] = kv; }
        }
        TS2D_STAMP_AT(a.prof, 3)
        lds_barrier();                                     // every wave is done with the patch; the scratch is complete (stores in flight)
        // tile partial (fixed order over the 4 waves, rebased onto wave 0's pivot).  FUSE: by wave 3, which has one M tile less to
        // recompute in the next tile's first phase (the other waves would wait for wave 0 at the next barrier)
        if (FUSE ? (tid >= 192 && tid < 224) : tid < 32) {
            const int co = tid & 31, c = co & 15;
            const float* s0 = scratch32(0, co >> 4);
            f32x4 acc4 = f32x4{s0[c], s0[16 + c], s0[32 + c], 64.f};
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) {
                const float* sw_ = scratch32(ww, co >> 4);
                const float d = sw_[32 + c] - acc4[2];
                acc4[1] += sw_[16 + c] + d * (2.f * sw_[c] + 64.f * d);
                acc4[0] += sw_[c] + 64.f * d;
                acc4[3] += 64.f;
            }
            *reinterpret_cast<f32x4*>(a.part + (((size_t)n * tpi + (t - n * tpi)) * 32 + co) * 4) = acc4;
        }
        txi = ntx; tyi = nty;
        TS2D_STAMP_AT(a.prof, 4)
    }
    TS2D_PROF_FLUSH(a.prof)
}

}  // namespace ts2d
