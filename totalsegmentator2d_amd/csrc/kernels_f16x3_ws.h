// Warp-specialised, persistent split-fp16 3x3 convolution (stride 1, tile inside one image).
//
// One 512-thread workgroup per CU: waves 0-3 are CONSUMERS (one per SIMD: LDS fragment reads + MFMA only), waves 4-7
// are PRODUCERS (global loads, InstanceNorm + LeakyReLU, fp16 hi/lo split, LDS writes, weight staging).  LDS holds
// two stages (patch + weights of one 16-channel chunk each); while the consumers run the MFMAs of step p from stage
// p & 1 the producers fill stage (p + 1) & 1 and already have the raw patch of step p + 2 in flight in registers.
// A "step" is one (tile, Cin-chunk) pair; the step sequence runs across tiles, so the producers stage the first chunk
// of the next tile while the consumers store the previous tile's output: no prologue/epilogue bubble per tile.
// One workgroup barrier per step.  Same arithmetic as conv3x3_f16x3 (bit-identical results).
#pragma once
#include "kernels_f16x3.h"

namespace ts2d {

constexpr int kWsThreads = 512;

template <int BN>
__global__ __launch_bounds__(kWsThreads, 2) void conv3x3_f16x3_ws(const ConvArgs a, const int n_virtual) {
    constexpr int NT = BN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];

    const int tid = threadIdx.x;
    const bool producer = tid >= 256;
    const int ptid = tid & 255, lane = tid & 63, w = (tid >> 6) & 3;
    const int r = lane & 31, h = lane >> 5;

    const int TH = 1 << a.lgTH, TW = 1 << a.lgTW;
    const int tpi = a.tiles_x * a.tiles_y;
    const int P = a.PH * a.PW;                        // lgNIMG == 0 (host guarantees)
    const int stage_bytes = P * kRec + 9 * BN * kRec;
    unsigned char* const stage0 = smem8;
    float* const red = reinterpret_cast<float*>(smem8 + 2 * stage_bytes);     // [4 waves][BN][2]
    const int nchunks = (a.C0 + a.C1) / 16;
    const int G = gridDim.x;

    // virtual block id -> (mtile, ctile); XCD-aware like the non-persistent kernels; returns false for padding ids
    auto decode = [&](int vb, int& mtile, int& ctile) -> bool {
        const int xcd = vb & 7, q8 = vb >> 3;
        mtile = (q8 / a.n_ctiles) * 8 + xcd;
        ctile = q8 % a.n_ctiles;
        return mtile < a.n_mtiles;
    };
    auto next_valid = [&](int vb) -> int {
        int m, c;
        while (vb < n_virtual && !decode(vb, m, c)) vb += G;
        return vb < n_virtual ? vb : -1;
    };
    auto item_coords = [&](int vb, int& n, int& ty0, int& tx0, int& tin, int& n0col) {
        int mtile, ctile;
        decode(vb, mtile, ctile);
        n = mtile / tpi; tin = mtile - n * tpi;
        const int tyi = tin / a.tiles_x, txi = tin - tyi * a.tiles_x;
        ty0 = tyi << a.lgTH; tx0 = txi << a.lgTW; n0col = ctile * BN;
    };

    if (producer) {
        // =============================================================== PRODUCER
        const int total = P * 2;
        const float inv_pw = 1.0f / (float)a.PW;
        const int oct = (ptid & 1) * 8;
        int goff[3];
        f32x4 pv[3][2];
        int wst0 = -1, wst1 = -1;         // column tile whose weights currently sit in each stage (residency for <= 2 chunks)
        // weights of the HELD step, prefetched one full phase ahead like the patch (named registers: arrays captured by the
        // lambdas below were demoted to scratch by the compiler)
        constexpr int WU = 9 * BN * 4, WIT = (WU + 255) / 256;
        uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8;
        bool w_need = false;
#define TS2D_WS_WLOAD(K, R) { const int idx = ptid + K * 256; if (K < WIT && idx < WU) R = wsrc_[idx]; }
#define TS2D_WS_WSTORE(K, R) { const int idx = ptid + K * 256; if (K < WIT && idx < WU) *reinterpret_cast<uint4*>(sB_ + (idx >> 2) * kRec + (idx & 3) * 16) = R; }

        auto plan = [&](int vb) {
            int n, ty0, tx0, tin, n0col;
            item_coords(vb, n, ty0, tx0, tin, n0col);
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int u = ptid + it * 256;
                int g = -1;
                if (u < total) {
                    const int pp = u >> 1;
                    const int py = (int)(((float)pp + 0.5f) * inv_pw), px = pp - py * a.PW;
                    const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
                    if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) g = (n * a.Hin + iy) * a.Win + ix;
                }
                goff[it] = g;
            }
        };
        auto issue_loads = [&](int ch) {
            int cb = ch * 16;
            const float* src = a.src0; int C = a.C0;
            if (cb >= a.C0) { cb -= a.C0; src = a.src1; C = a.C1; }
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                pv[it][0] = f32x4{0.f, 0.f, 0.f, 0.f}; pv[it][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (goff[it] >= 0) {
                    const float* p = src + (size_t)goff[it] * C + cb + oct;
                    pv[it][0] = *reinterpret_cast<const f32x4*>(p);
                    pv[it][1] = *reinterpret_cast<const f32x4*>(p + 4);
                }
            }
        };
        // write the step held in registers (item image n_w, chunk ch_w, validity vmask) into `stage`
        auto write_step = [&](unsigned char* stage, int n_w, int ch_w, int vmask, bool stage_w) {
            unsigned char* sA = stage;
            unsigned char* sB_ = stage + P * kRec;
            int cb = ch_w * 16;
            const float* sc = a.sc0; const float* sh = a.sh0; int C = a.C0;
            if (cb >= a.C0) { cb -= a.C0; sc = a.sc1; sh = a.sh1; C = a.C1; }
            f32x4 s1a = f32x4{1.f, 1.f, 1.f, 1.f}, s1b = s1a, s2a = f32x4{0.f, 0.f, 0.f, 0.f}, s2b = s2a;
            if (sc != nullptr) {
                const size_t o = (size_t)n_w * C + cb + oct;
                s1a = *reinterpret_cast<const f32x4*>(sc + o); s1b = *reinterpret_cast<const f32x4*>(sc + o + 4);
                s2a = *reinterpret_cast<const f32x4*>(sh + o); s2b = *reinterpret_cast<const f32x4*>(sh + o + 4);
            }
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int u = ptid + it * 256;
                if (u < total) {
                    f32x4 va = pv[it][0], vb = pv[it][1];
                    if (sc != nullptr && ((vmask >> it) & 1)) {
                        va = va * s1a + s2a; vb = vb * s1b + s2b;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            va[e] = fmaxf(va[e], va[e] * a.slope);
                            vb[e] = fmaxf(vb[e], vb[e] * a.slope);
                        }
                    }
                    half8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const _Float16 ha = (_Float16)va[e], hb = (_Float16)vb[e];
                        hi[e] = ha; hi[e + 4] = hb;
                        lo[e] = (_Float16)(va[e] - (float)ha); lo[e + 4] = (_Float16)(vb[e] - (float)hb);
                    }
                    unsigned char* d = sA + (u >> 1) * kRec + (u & 1) * 16;
                    *reinterpret_cast<half8*>(d) = hi;
                    *reinterpret_cast<half8*>(d + 32) = lo;
                }
            }
            if (stage_w) {                  // weight registers (loaded a phase ago) -> LDS records
                TS2D_WS_WSTORE(0, w0) TS2D_WS_WSTORE(1, w1) TS2D_WS_WSTORE(2, w2) TS2D_WS_WSTORE(3, w3) TS2D_WS_WSTORE(4, w4)
                TS2D_WS_WSTORE(5, w5) TS2D_WS_WSTORE(6, w6) TS2D_WS_WSTORE(7, w7) TS2D_WS_WSTORE(8, w8)
            }
        };
        auto flush_red = [&](int vb) {      // 4 consumer waves' partial (sum, sum of squares) -> global partial buffer
            if (a.part != nullptr && ptid < BN) {
                int n, ty0, tx0, tin, n0col;
                item_coords(vb, n, ty0, tx0, tin, n0col);
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) { s += red[(ww * BN + ptid) * 2]; q += red[(ww * BN + ptid) * 2 + 1]; }
                float* p = a.part + ((size_t)(n * tpi + tin) * a.Cout + n0col + ptid) * 2;
                p[0] = s; p[1] = q;
            }
        };

        // cursors: (l_vb, l_ch) next step to LOAD; registers currently hold step (w_vb, w_ch)
        int l_vb = next_valid(blockIdx.x), l_ch = 0, l_p = 0;          // l_p = step index of the load cursor (stage = l_p & 1)
        int w_vb = -1, w_ch = 0, w_mask = 0;
        auto advance_load = [&]() {          // registers <- step (l_vb, l_ch); move the load cursor on
            w_vb = l_vb; w_ch = l_ch;
            if (l_vb >= 0) {
                if (l_ch == 0) plan(l_vb);
                w_mask = (goff[0] >= 0 ? 1 : 0) | (goff[1] >= 0 ? 2 : 0) | (goff[2] >= 0 ? 4 : 0);
                issue_loads(l_ch);
                {   // weights of that step, unless the stage it will go to still holds them (<= 2 chunks, same column tile)
                    int mt_, ct_;
                    decode(l_vb, mt_, ct_);
                    w_need = !(nchunks <= 2 && ((l_p & 1) ? wst1 : wst0) == ct_);
                    if (w_need) {
                        const uint4* wsrc_ = reinterpret_cast<const uint4*>(a.wph) + ((size_t)l_ch * a.n_ctiles + ct_) * (9 * BN * 4);
                        TS2D_WS_WLOAD(0, w0) TS2D_WS_WLOAD(1, w1) TS2D_WS_WLOAD(2, w2) TS2D_WS_WLOAD(3, w3) TS2D_WS_WLOAD(4, w4)
                        TS2D_WS_WLOAD(5, w5) TS2D_WS_WLOAD(6, w6) TS2D_WS_WLOAD(7, w7) TS2D_WS_WLOAD(8, w8)
                    }
                }
                ++l_p;
                if (++l_ch == nchunks) { l_ch = 0; l_vb = next_valid(l_vb + G); }
            }
        };
        auto write_held = [&](int p) {       // registers (step p) -> stage p & 1
            int n, ty0, tx0, tin, n0col;
            item_coords(w_vb, n, ty0, tx0, tin, n0col);
            const int ct = n0col / BN;
            write_step(stage0 + (p & 1) * stage_bytes, n, w_ch, w_mask, w_need);
            if (p & 1) wst1 = (nchunks <= 2) ? ct : -1; else wst0 = (nchunks <= 2) ? ct : -1;
        };

        advance_load();                       // step 0 -> registers
        if (w_vb >= 0) { write_held(0); advance_load(); }     // step 0 -> stage 0 ; step 1 -> registers
        __syncthreads();
        int c_vb = next_valid(blockIdx.x), prev_end_vb = -1;
        for (int p = 0; c_vb >= 0;) {
            for (int ch = 0; ch < nchunks; ++ch, ++p) {
                if (prev_end_vb >= 0) { flush_red(prev_end_vb); prev_end_vb = -1; }
                if (w_vb >= 0) { write_held(p + 1); advance_load(); }
                __syncthreads();
            }
            prev_end_vb = c_vb;
            c_vb = next_valid(c_vb + G);
        }
        if (prev_end_vb >= 0) flush_red(prev_end_vb);
#undef TS2D_WS_WLOAD
#undef TS2D_WS_WSTORE
    } else {
        // =============================================================== CONSUMER
        __builtin_amdgcn_s_setprio(1);
        int abase[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = 64 * w + 32 * mt + r;
            const int ty = (m >> a.lgTW) & (TH - 1), tx = m & (TW - 1);
            abase[mt] = (ty * a.PW + tx) * kRec + 16 * h;
        }
        const int bbase = P * kRec + r * kRec + 16 * h;
        const float oscale = *a.oscale;
        __syncthreads();
        int c_vb = next_valid(blockIdx.x);
        for (int p = 0; c_vb >= 0;) {
            f32x16 acc_t[2][NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc_t[mt][nt][i] = 0.f;
            for (int ch = 0; ch < nchunks; ++ch, ++p) {
                const unsigned char* st = stage0 + (p & 1) * stage_bytes;
                f32x16 acc_c[2][NT];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc_c[mt][nt][i] = 0.f;
                // software-pipelined over the 9 taps: fragments of tap t+1 are read while the 12 MFMAs of tap t run
                half8 fa[2][2][2], fb[2][NT][2];            // [buffer][tile][hi/lo]
                auto load_frags = [&](int buf, int tap) {
                    const int toff = ((tap / 3) * a.PW + (tap % 3)) * kRec;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        fa[buf][mt][0] = *reinterpret_cast<const half8*>(st + abase[mt] + toff);
                        fa[buf][mt][1] = *reinterpret_cast<const half8*>(st + abase[mt] + toff + 32);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        fb[buf][nt][0] = *reinterpret_cast<const half8*>(st + (tap * BN + nt * 32) * kRec + bbase);
                        fb[buf][nt][1] = *reinterpret_cast<const half8*>(st + (tap * BN + nt * 32) * kRec + bbase + 32);
                    }
                };
                load_frags(0, 0);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int cur = tap & 1;
                    if (tap + 1 < 9) load_frags(cur ^ 1, tap + 1);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][1], fb[cur][nt][0], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][1], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][mt][0], fb[cur][nt][0], acc_c[mt][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);      // keep the read-ahead distance at one tap (register budget)
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc_t[mt][nt] += acc_c[mt][nt];
                if (ch == nchunks - 1) {
                    // ---- tile epilogue (producers are already staging the next tile)
                    int n, ty0, tx0, tin, n0col;
                    item_coords(c_vb, n, ty0, tx0, tin, n0col);
                    // opaque copies: keep the 32 per-row store addresses from being hoisted out of the persistent loop
                    // (they would stay live across the MFMA phase and cost 64 VGPRs)
                    int rr = r, hh = h, ww = w;
                    asm volatile("" : "+v"(rr), "+v"(hh), "+v"(ww));
                    const bool full = (ty0 + TH <= a.Ht) && (tx0 + TW <= a.Wt);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int co = n0col + nt * 32 + rr;
                        const float bv = a.bias[co];
                        float ss = 0.f, qq = 0.f;
                        float* const obase = a.dst + ((size_t)(n * a.Ht + ty0) * a.Wt + tx0) * a.Cout + co;
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
                                const int m = 64 * ww + 32 * mt + row;
                                const int ty = (m >> a.lgTW) & (TH - 1), tx = m & (TW - 1);
                                if (full || (ty0 + ty < a.Ht && tx0 + tx < a.Wt)) {
                                    const float v = acc_t[mt][nt][i] * oscale + bv;
                                    obase[(size_t)(ty * a.Wt + tx) * a.Cout] = v;
                                    ss += v; qq += v * v;
                                }
                            }
                        }
                        ss += __shfl_xor(ss, 32); qq += __shfl_xor(qq, 32);
                        if (hh == 0) { red[(ww * BN + nt * 32 + rr) * 2] = ss; red[(ww * BN + nt * 32 + rr) * 2 + 1] = qq; }
                    }
                }
                __syncthreads();
            }
            c_vb = next_valid(c_vb + G);
        }
    }
}

}  // namespace ts2d
