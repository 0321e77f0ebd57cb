// First conv block (Conv2d C->f0, C = 1..4 input channels, 3x3, stride 1) on the exact fp32 MFMA.
// K = 9*C is tiny, so the layer is HBM-bound on its OUTPUT (33.5 MB per 512x512 slice).  The generic kernel padded C to 8
// (K = 72) and needed a separate NCHW->NHWC pass; here each tap is ONE v_mfma_f32_32x32x2_f32 per channel pair
// (k = the two channels), the patch is read straight from the NCHW boundary tensor, and the weights live in registers
// (one float per lane per (tap, pair, column tile)).  The network input is not normalised, so there is no prologue math.
#pragma once
#include "kernels_f16x3.h"

namespace ts2d {

struct FirstArgs {
    const float* x;       // NCHW [B, C, H, W] (the boundary input)
    const float* w;       // PyTorch layout [Cout][C][3][3]
    const float* bias;    // [Cout]
    float* dst;           // raw NHWC [B, H, W, Cout]
    float* part;          // partial statistics [n][tile][Cout][2] or nullptr (tile spans several images)
    int B, C, H, W, Cout;
    int lgTH, lgTW, lgNIMG, tiles_x, tiles_y, n_mtiles, PH, PW;
    int tr_off;           // byte offset of the per-wave transpose regions in LDS (fp16 STORE variant on complete tiles: kFirstTr bytes per wave), or 0
};

constexpr int kFirstTrPitch = 80, kFirstTr = 32 * kFirstTrPitch;      // one M tile as [32 pixels][32 channels] halves, pixel pitch 80 B (bank-disjoint lane halves)

// FULL: every tile is a complete 256-pixel tile inside one image (the engine checks) - the workgroups are persistent (grid < tiles)
// and carry only the fast epilogue; !FULL: one tile per workgroup, both epilogues (ragged / multi-image test geometries).
// STORE = false (round 4): the statistics-only pass in front of the fused second block (conv3x3_res32<.., FUSE>, kernels_res32.h) - the same
// values, the same per-tile shifted partials, nothing written but them: the layer's 33.5 MB of output per slice never exist.
template <int NT, int KP, typename ST = float, bool FULL = false, bool STORE = true>      // NT = Cout / 32 column tiles, KP = ceil(C / 2) channel pairs, ST = output storage
__global__ TS2D_PACKED_F32 __launch_bounds__(kBlock, FULL ? (NT == 1 ? 4 : 2) : 1) void conv3x3_first(const FirstArgs a) {
    constexpr int CP = 2 * KP + 1;               // floats per patch pixel (+1 pad: conflict-free ds_read_b32)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TH = 1 << a.lgTH, TW = 1 << a.lgTW, NIMG = 1 << a.lgNIMG;
    const int tpi = a.tiles_x * a.tiles_y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int PHW = a.PH * a.PW, P = PHW << a.lgNIMG;

    // weights -> registers: B[k = h][j = r] of (tap, pair kp, column tile nt) = w[co = 32 nt + r][c = 2 kp + h][tap]
    float wr[9][KP][NT];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c = 2 * kp + h;
                wr[tap][kp][nt] = c < a.C ? a.w[((size_t)(nt * 32 + r) * a.C + c) * 9 + tap] : 0.f;
            }

    // Persistent workgroups (round 2): a workgroup's life used to be ONE serial chain - weight loads, patch loads, LDS, barrier,
    // 18 MFMAs, stores - so the layer ran at the latency of that chain times tiles / resident workgroups (0.58 ms for 2.1 GB of
    // output = 3.6 TB/s; the write-only rate of this pool is 5.3-5.6).  Now the weights are loaded once per workgroup and the NEXT
    // tile's patch values are in flight (registers) while the current tile computes and stores: 0.58 -> 0.49 ms (FULL variant only:
    // with both epilogues inside the tile loop hipcc needs > 200 registers).
    const float inv_phw = 1.0f / (float)PHW, inv_pw = 1.0f / (float)a.PW;
    constexpr int MAXI = 4;                                   // prefetch registers per thread (P * 2 KP <= 4 * 256: every one-image tile)
    const bool pf = FULL && P * 2 * KP <= MAXI * kBlock;     // uniform
    // element idx = tid + k * 256 of a tile's patch (plane-major: consecutive threads walk a row): everything that does not depend
    // on the tile is computed once - offset relative to the tile origin, LDS slot, packed (il, py, px)
    int eoff[MAXI], elds[MAXI], epk[MAXI];
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
        const int idx = tid + k * kBlock;
        const int c = idx / P, pp = idx - c * P;
        const int il = (int)(((float)pp + 0.5f) * inv_phw), rem = pp - il * PHW;
        const int py = (int)(((float)rem + 0.5f) * inv_pw), px = rem - py * a.PW;
        const bool ok = idx < P * 2 * KP && c < a.C;
        eoff[k] = ((il * a.C + c) * a.H + (py - 1)) * a.W + (px - 1);
        elds[k] = idx < P * 2 * KP ? pp * CP + c : -1;
        epk[k] = ok ? (il << 20) | (py << 10) | px : -1;
    }
    auto patch_value = [&](int k, int mt_) -> float {
        const int grp_ = mt_ / tpi, tin_ = mt_ - grp_ * tpi, tyi_ = tin_ / a.tiles_x, txi_ = tin_ - tyi_ * a.tiles_x;
        const int n0_ = grp_ << a.lgNIMG, y0_ = tyi_ << a.lgTH, x0_ = txi_ << a.lgTW;
        const int il = epk[k] >> 20, py = (epk[k] >> 10) & 1023, px = epk[k] & 1023;
        const int iy = y0_ - 1 + py, ix = x0_ - 1 + px;
        float v = 0.f;
        if (epk[k] >= 0 && n0_ + il < a.B && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
            v = a.x[(size_t)((long long)(n0_ * a.C * a.H + y0_) * a.W + x0_ + eoff[k])];
        return v;
    };
    float pvn[MAXI];
    if (pf) {
#pragma unroll
        for (int k = 0; k < MAXI; ++k) pvn[k] = patch_value(k, (int)blockIdx.x);
    }
    for (int mtile = blockIdx.x; mtile < a.n_mtiles; mtile += gridDim.x) {
    const int grp = mtile / tpi, tin = mtile - grp * tpi;
    const int tyi = tin / a.tiles_x, txi = tin - tyi * a.tiles_x;
    const int nimg0 = grp << a.lgNIMG;
    const int ty0 = tyi << a.lgTH, tx0 = txi << a.lgTW;
    if (mtile != (int)blockIdx.x) lds_barrier();             // the previous tile's LDS reads (fragments, statistics) are done
    if (pf) {
#pragma unroll
        for (int k = 0; k < MAXI; ++k) if (elds[k] >= 0) smem[elds[k]] = pvn[k];
    } else {                                                  // (multi-image tiles of tiny test images: loaded in place)
        for (int idx = tid; idx < P * 2 * KP; idx += kBlock) {
            const int c = idx / P, pp = idx - c * P;
            const int il = (int)(((float)pp + 0.5f) * inv_phw), rem = pp - il * PHW;
            const int py = (int)(((float)rem + 0.5f) * inv_pw), px = rem - py * a.PW;
            const int n = nimg0 + il, iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            float v = 0.f;
            if (c < a.C && n < a.B && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = a.x[((size_t)(n * a.C + c) * a.H + iy) * a.W + ix];
            smem[pp * CP + c] = v;
        }
    }
    lds_barrier();
    if (pf && mtile + (int)gridDim.x < a.n_mtiles) {         // the next tile's patch: in flight during this tile's MFMAs and stores
#pragma unroll
        for (int k = 0; k < MAXI; ++k) pvn[k] = patch_value(k, mtile + (int)gridDim.x);
    }

    int abase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = 64 * w + 32 * mt + r;
        const int il = m >> (a.lgTH + a.lgTW), ty = (m >> a.lgTW) & (TH - 1), tx = m & (TW - 1);
        abase[mt] = (il < NIMG ? (il * PHW + ty * a.PW + tx) * CP : 0) + h;
    }
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int toff = ((tap / 3) * a.PW + (tap % 3)) * CP;
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            float av[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) av[mt] = smem[abase[mt] + toff + 2 * kp];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], wr[tap][kp][nt], acc[mt][nt], 0, 0, 0);
        }
    }

    float st_s[NT], st_q[NT], st_k[NT], st_n[NT];
    // complete one-image tile: one lane offset per 32x32 block + wave-uniform row offsets (see split_epilogue_one)
    const bool full = FULL || (a.lgNIMG == 0 && a.lgTH + a.lgTW == 8 && ty0 + TH <= a.H && tx0 + TW <= a.W && a.lgTW >= 4 && nimg0 < a.B);
    const size_t img_el = (size_t)a.H * a.W * a.Cout;
    const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<ST*>(a.dst) + (size_t)nimg0 * img_el, 0,
                                                       (int)(full ? img_el * sizeof(ST) : 0), 0x00020000);
    float bvs[NT];       // every bias value before the first store: a load issued between stores waits (in-order vmcnt) for the stores ahead of it
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bvs[nt] = a.bias[nt * 32 + r];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 32 + r;
        const float bv = bvs[nt];
        const float kv = stat_pivot(round_act<ST>(acc[0][nt][0] + bv));      // shifted statistics (kernels.h)
        float ss = 0.f, qq = 0.f, nn = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if constexpr (FULL && STORE && sizeof(ST) == 2 && NT == 1) {
                // 16-bit storage, 32 channels (round 5; VERDICT r4 #4a: 452 us to write 1.07 GB with 2-byte stores): the M tile = 32
                // consecutive pixels of a tile row = 2 KB of contiguous NHWC output.  Each wave transposes it through a private LDS region
                // ([pixel][channel] halves; same-wave LDS operations execute in order: no barrier) and stores it as 2 x 16 bytes per lane.
                unsigned char* tw = reinterpret_cast<unsigned char*>(smem) + a.tr_off + w * kFirstTr;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int px = 4 * h + (i & 3) + 8 * (i >> 2);
                    const float v = acc[mt][nt][i] + bv;
                    const _Float16 hv = (_Float16)v;
                    *reinterpret_cast<_Float16*>(tw + px * kFirstTrPitch + r * 2) = hv;
                    const float d = (float)hv - kv;                                   // statistics of what is stored
                    ss += d; qq = __builtin_fmaf(d, d, qq);
                }
                nn += 16.f;
                typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
                const int m0 = 64 * w + 32 * mt;                                        // first GEMM row of the M tile: tile pixel (m >> lgTW, m & (TW - 1))
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int piece = k * 64 + lane;                                    // 16-byte piece of the 2 KB: pixel piece / 4, part piece % 4
                    const int m = m0 + (piece >> 2);                                    // (TW = 32: one tile row = 2 KB of contiguous output; TW = 16: two rows)
                    const unsigned off = (unsigned)((((ty0 + (m >> a.lgTW)) * a.W + tx0 + (m & (TW - 1))) * 32) * 2 + (piece & 3) * 16);
                    const u32x4s q = *reinterpret_cast<const u32x4s*>(tw + (piece >> 2) * kFirstTrPitch + (piece & 3) * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(q, rsd, off, 0, 0);
                    asm volatile("s_nop 3" :: "v"(q) : "memory");                      // gfx950 wide-store hazard (kernels_up0.h)
                }
            } else if (FULL || full) {
                const int m0 = 64 * w + 32 * mt + 4 * h;
                const int oy = ty0 + (m0 >> a.lgTW), ox = tx0 + (m0 & (TW - 1));
                const unsigned voff = (unsigned)(((oy * a.W + ox) * a.Cout + co) * (int)sizeof(ST));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rowoff = (i & 3) + 8 * (i >> 2);
                    const unsigned soff = (unsigned)((((rowoff & (TW - 1)) + (rowoff >> a.lgTW) * a.W) * a.Cout) * (int)sizeof(ST));
                    float v = acc[mt][nt][i] + bv;
                    if constexpr (STORE) buffer_store_act<ST>(v, rsd, voff, soff);
                    const float d = round_act<ST>(v) - kv;
                    ss += d; qq = __builtin_fmaf(d, d, qq);
                }
                nn += 16.f;
            } else if constexpr (!FULL) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                    const int m = 64 * w + 32 * mt + row;
                    const int il = m >> (a.lgTH + a.lgTW), ty = (m >> a.lgTW) & (TH - 1), tx = m & (TW - 1);
                    const int n = nimg0 + il, oy = ty0 + ty, ox = tx0 + tx;
                    if (il < NIMG && n < a.B && oy < a.H && ox < a.W) {
                        float v = acc[mt][nt][i] + bv;
                        store_act<ST>(a.dst, ((size_t)(n * a.H + oy) * a.W + ox) * a.Cout + co, v);
                        const float d = round_act<ST>(v) - kv;
                        ss += d; qq = __builtin_fmaf(d, d, qq); nn += 1.f;
                    }
                }
            }
        }
        st_s[nt] = ss; st_q[nt] = qq; st_k[nt] = kv; st_n[nt] = nn;
    }
    if (a.part != nullptr) {
        lds_barrier();
        float* red = smem;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float s = st_s[nt], q = st_q[nt], n = st_n[nt];
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32); n += __shfl_xor(n, 32);
            if (h == 0) stat_wave_put(red, w * NT * 32 + nt * 32 + r, s, q, st_k[nt], n);
        }
        lds_barrier();
        if (tid < NT * 32) stat_tile_store(red, 4, NT * 32, tid, a.part + ((size_t)(nimg0 * tpi + tin) * a.Cout + tid) * 4);
    }
    }   // tiles of this workgroup
}

}  // namespace ts2d
