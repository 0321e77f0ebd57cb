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

template <int BN, typename ST, int NP>
__global__ __launch_bounds__(kS2Threads, 2) void conv3x3s2_v2(const ConvArgs a) {
    constexpr int NPP = NP == 3 ? 2 : 1;                   // fp16 parts per value
    constexpr int NTW = BN / 64;                           // 32-column MFMA tiles per wave (a wave owns BN / 2 columns)
    constexpr int WTAP = NPP * 2 * BN * 16, WB = 9 * WTAP; // weight bytes per tap / per chunk
    constexpr int MAXU = 5;                                // staging units per thread: 5 x 256 slots >= 1122
    constexpr int NL = sizeof(ST) == 4 ? 2 : 1;            // 16-byte loads per unit (8 channels)
    constexpr int WIT = (WB / 16 + kS2Threads - 1) / kS2Threads;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = bid >> 3;
    const int qm = q8 >> a.lg_nct;                         // (power-of-two tilings only: the engine checks)
    const int mtile = qm * 8 + xcd;
    const int ctile = q8 - qm * a.n_ctiles;
    if (mtile >= a.n_mtiles) return;
    const int n0col = ctile * BN;
    const int tpi = a.tiles_x * a.tiles_y;
    const int nimg0 = mtile >> a.lg_tpi, tin = mtile - nimg0 * tpi;
    const int tyi = tin >> a.lg_tx, txi = tin - tyi * a.tiles_x;
    const int ty0 = tyi * 8, tx0 = txi * 32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 3, wn = w >> 2;
    const int r = lane & 31, h = lane >> 5;

    unsigned char* sA = smem8;                             // [part][h][slot] x 16 B
    unsigned char* sB = smem8 + NPP * 2 * kS2Plane;        // [tap][part][h][column BN] x 16 B

    // ---- staging plan.  A wave instruction covers 32 slots x 2 channel octets: slot = 32 (8 it + w) + (lane & 7) + 8 (lane >> 4),
    //      octet = (lane >> 3) & 1 - 8 consecutive lanes write 8 consecutive slots of one plane.  Slot q = patch row q / 66, then the
    //      33 even columns, then the 33 odd ones (the 66th slot of a row is unused).  Padding pixels are zeroed once and never staged.
    const int oct = (lane >> 3) & 1;
    unsigned vo[MAXU];                                     // byte offset of the unit's 8 channels in the image, 0x80000000 = no unit
    int lw[MAXU];
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int q = 32 * (8 * it + w) + (lane & 7) + 8 * (lane >> 4);
        const int py = q / kS2PW, rem = q - py * kS2PW;
        const int half = rem >= 33 ? 1 : 0, px = 2 * (rem - 33 * half) + half;
        const int iy = 2 * ty0 - 1 + py, ix = 2 * tx0 - 1 + px;
        unsigned v = 0x80000000u;
        lw[it] = oct * kS2Plane + q * 16;
        if (q < kS2Slots) {
            if (px <= 64 && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) v = (unsigned)(((iy * a.Win + ix) * a.C0 + 8 * oct) * (int)sizeof(ST));
            else { *reinterpret_cast<uint4*>(sA + lw[it]) = uint4{0u, 0u, 0u, 0u};
                   if (NPP == 2) *reinterpret_cast<uint4*>(sA + lw[it] + 2 * kS2Plane) = uint4{0u, 0u, 0u, 0u}; }
        }
        vo[it] = v;
    }

    const size_t img_px = (size_t)a.Hin * a.Win;
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(reinterpret_cast<const ST*>(a.src0)) + (size_t)nimg0 * img_px * a.C0, 0,
                                                       (int)(img_px * a.C0 * sizeof(ST)), 0x00020000);
    u32x4 pv[MAXU][NL];
    auto prefetch = [&](int ch) {
#pragma unroll
        for (int it = 0; it < MAXU; ++it)
#pragma unroll
            for (int l = 0; l < NL; ++l) pv[it][l] = __builtin_amdgcn_raw_buffer_load_b128(rs0, vo[it] + 16 * l, ch * 16 * (int)sizeof(ST), 0);
    };
    const int nchunks = a.C0 / 16;                         // the strided conv never reads a concat
    prefetch(0);

    // ---- lane constants of the MFMA phase: output pixel (2 wm + mt, r) reads patch row 2 (2 wm + mt) + dy, slot (dx & 1) 33 + r + (dx >> 1)
    const int abase = h * kS2Plane + ((4 * wm) * kS2PW + r) * 16;                  // + mt * 2 * 66 * 16 + part * 2 * Plane + tap offset
    const int bbase = NPP * 2 * kS2Plane + h * BN * 16 + (wn * (BN / 2) + r) * 16;  // + tap * WTAP + part * 2 * BN * 16 + nt * 512
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
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();                                   // the previous chunk's MFMA reads of LDS are done
        TS2D_STAMP_AT(a.prof, 1)
        // scale / shift of this thread's 8 channels
        f32x4 nsa = f32x4{1.f, 1.f, 1.f, 1.f}, nsb = nsa, nta = f32x4{0.f, 0.f, 0.f, 0.f}, ntb = nta;
        const bool normed = a.sc0 != nullptr;
        if (normed) {
            const float* ps = a.sc0 + (size_t)nimg0 * a.C0 + ch * 16 + 8 * oct; const float* pt = a.sh0 + (size_t)nimg0 * a.C0 + ch * 16 + 8 * oct;
            nsa = *reinterpret_cast<const f32x4*>(ps); nsb = *reinterpret_cast<const f32x4*>(ps + 4);
            nta = *reinterpret_cast<const f32x4*>(pt); ntb = *reinterpret_cast<const f32x4*>(pt + 4);
        }
        // ---- weights of this chunk: one linear block; loads issued first, written to LDS behind the patch conversion
        const uint4* wsrc = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * (9 * 2 * 2 * BN * 16));
        uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8;          // (named registers: an indexed array ends up in scratch)
        // (f16 mode: the hi parts only - the first 2 BN slots of every 4 BN)
#define TS2D_WLOAD(K, R) { const int sl = tid + K * kS2Threads; if (K < WIT && (WB / 16 % kS2Threads == 0 || sl < WB / 16)) \
            R = wsrc[NPP == 2 ? sl : (sl / (2 * BN)) * (4 * BN) + sl % (2 * BN)]; }
        TS2D_WLOAD(0, w0) TS2D_WLOAD(1, w1) TS2D_WLOAD(2, w2) TS2D_WLOAD(3, w3) TS2D_WLOAD(4, w4)
        TS2D_WLOAD(5, w5) TS2D_WLOAD(6, w6) TS2D_WLOAD(7, w7) TS2D_WLOAD(8, w8)
#undef TS2D_WLOAD
        // ---- patch: InstanceNorm + LeakyReLU on the fly, split into fp16 hi / lo
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            if (vo[it] != 0x80000000u && !(a.dbg & 2)) {
                unsigned char* d = sA + lw[it];
                if constexpr (sizeof(ST) == 4) {
                    f32x4 va = __builtin_bit_cast(f32x4, pv[it][0]), vb = __builtin_bit_cast(f32x4, pv[it][NL - 1]);
                    if (normed) {
                        va = va * nsa + nta; vb = vb * nsb + ntb;
                        const f32x4 na = va * slope4, nb2 = vb * slope4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { va[e] = fmaxf(va[e], na[e]); vb[e] = fmaxf(vb[e], nb2[e]); }      // LeakyReLU (0 < slope < 1)
                    }
                    uint4 hi, lo;
                    split_hi_lo_8(va, vb, hi, lo);
                    *reinterpret_cast<uint4*>(d) = hi;
                    if (NPP == 2) *reinterpret_cast<uint4*>(d + 2 * kS2Plane) = lo;
                } else {
                    uint4 x = uint4{pv[it][0][0], pv[it][0][1], pv[it][0][2], pv[it][0][3]};
                    if (normed) x = norm_lrelu_8(x, nsa, nsb, nta, ntb, slope2);
                    *reinterpret_cast<uint4*>(d) = x;
                }
            }
        }
#define TS2D_WSTORE(K, R) { const int sl = tid + K * kS2Threads; if (K < WIT && (WB / 16 % kS2Threads == 0 || sl < WB / 16)) \
            *reinterpret_cast<uint4*>(sB + sl * 16) = R; }
        if (!(a.dbg & 4)) {
        TS2D_WSTORE(0, w0) TS2D_WSTORE(1, w1) TS2D_WSTORE(2, w2) TS2D_WSTORE(3, w3) TS2D_WSTORE(4, w4)
        TS2D_WSTORE(5, w5) TS2D_WSTORE(6, w6) TS2D_WSTORE(7, w7) TS2D_WSTORE(8, w8)
        }
#undef TS2D_WSTORE
        __syncthreads();
        TS2D_STAMP_AT(a.prof, 0)
        if (ch + 1 < nchunks) prefetch(ch + 1);            // HBM latency hides behind the MFMA phase

        f32x16 acc_c[2][NTW];                              // fresh accumulator per chunk (accuracy)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc_c[mt][nt][i] = 0.f;
        __builtin_amdgcn_s_setprio(1);
        if (!(a.dbg & 1)) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - 3 * dy;
            const int toff = (dy * kS2PW + (dx & 1) * 33 + (dx >> 1)) * 16;
            half8 fa[2][NPP], fb[NTW][NPP];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int p = 0; p < NPP; ++p) fa[mt][p] = *reinterpret_cast<const half8*>(smem8 + abase + mt * 2 * kS2PW * 16 + p * 2 * kS2Plane + toff);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int p = 0; p < NPP; ++p) fb[nt][p] = *reinterpret_cast<const half8*>(smem8 + bbase + tap * WTAP + p * 2 * BN * 16 + nt * 512);
            if constexpr (NP == 3) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][1], fb[nt][0], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][0], fb[nt][1], acc_c[mt][nt], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][0], fb[nt][0], acc_c[mt][nt], 0, 0, 0);
        }
        }
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc_t[mt][nt] += acc_c[mt][nt];
    }

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
            for (int i = 0; i < 16; ++i) {
                const unsigned soff = (unsigned)((((i & 3) + 8 * (i >> 2)) * a.Cout) * (int)sizeof(ST));      // scalar
                float v = __builtin_fmaf(acc_t[mt][nt][i], oscale, bv);
                buffer_store_act<ST>(v, rsd, voff, soff);
                const float d = round_act<ST>(v) - kv;                           // statistics of what is stored
                s += d; q = __builtin_fmaf(d, d, q);
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
    TS2D_PROF_FLUSH(a.prof)
}

}  // namespace ts2d
