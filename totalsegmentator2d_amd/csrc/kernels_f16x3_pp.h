// Ping-pong split-fp16 3x3 convolution (stride 1).
//
// Ablation of conv3x3_f16x3 showed time = MFMA + staging: two independent 256-thread workgroups on one CU do not
// interleave their phases by themselves.  Here ONE 512-thread workgroup owns two pixel tiles (group X = waves 0-3,
// group Y = waves 4-7; one wave of each group per SIMD), each with its own LDS stage, and workgroup barriers force the
// anti-phase: in time slot t group g runs   k = t - g;  k even -> STAGE chunk k/2,  k odd -> MFMA chunk (k-1)/2.
// So while X issues MFMAs (matrix pipe) Y converts/writes its next chunk (VALU, LDS, memory pipes) on the same SIMDs,
// and the slot length is max(MFMA, staging) instead of their sum.  Same arithmetic and summation order as
// conv3x3_f16x3 (bit-identical results).
#pragma once
#include "kernels_f16x3.h"

namespace ts2d {

template <int BN, int MAXU>
__global__ __launch_bounds__(512, 2) void conv3x3_f16x3_pp(const ConvArgs a) {
    constexpr int NT = BN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];

    // block -> (tile pair, column tile), XCD-aware like the other kernels; group g takes pixel tile 2 * pair + g
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = bid >> 3;
    const int n_pairs = (a.n_mtiles + 1) >> 1;
    const int pair = (q8 / a.n_ctiles) * 8 + xcd;
    const int ctile = q8 % a.n_ctiles;
    if (pair >= n_pairs) return;
    const int n0col = ctile * BN;

    const int g = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int mtile = 2 * pair + g;
    const bool active = mtile < a.n_mtiles;

    const int TH = 1 << a.lgTH, TW = 1 << a.lgTW, NIMG = 1 << a.lgNIMG;
    const int tpi = a.tiles_x * a.tiles_y;
    const int grp = mtile / tpi, tin = mtile - grp * tpi;
    const int tyi = tin / a.tiles_x, txi = tin - tyi * a.tiles_x;
    const int nimg0 = grp << a.lgNIMG;
    const int ty0 = tyi << a.lgTH, tx0 = txi << a.lgTW;

    const int PHW = a.PH * a.PW;
    const int P = PHW << a.lgNIMG;
    const int stage_bytes = P * kRec + 9 * BN * kRec;
    unsigned char* sA = smem8 + g * stage_bytes;
    unsigned char* sB = sA + P * kRec;
    float* red = reinterpret_cast<float*>(smem8 + 2 * stage_bytes) + g * (4 * BN * 2);

    int goff[MAXU];
    unsigned long long imgbits = 0;
    const int total = P * 2;
    const float inv_phw = 1.0f / (float)PHW, inv_pw = 1.0f / (float)a.PW;
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int u = tid + it * kBlock;
        int gg = -1;
        if (active && u < total) {
            const int pp = u >> 1;
            const int il = (int)(((float)pp + 0.5f) * inv_phw), rem = pp - il * PHW;
            const int py = (int)(((float)rem + 0.5f) * inv_pw), px = rem - py * a.PW;
            const int n = nimg0 + il, iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            if (n < a.B && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) gg = (n * a.Hin + iy) * a.Win + ix;
            imgbits |= (unsigned long long)il << (4 * it);
        }
        goff[it] = gg;
    }
    const int oct = (tid & 1) * 8;

    int abase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = 64 * w + 32 * mt + r;
        const int il = m >> (a.lgTH + a.lgTW), ty = (m >> a.lgTW) & (TH - 1), tx = m & (TW - 1);
        abase[mt] = (il < NIMG ? (il * PHW + ty * a.PW + tx) * kRec : 0) + 16 * h;
    }
    const int bbase = r * kRec + 16 * h;

    f32x16 acc_t[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_t[mt][nt][i] = 0.f;

    const int nchunks = (a.C0 + a.C1) / 16;
    f32x4 pv[MAXU][2];
    auto chunk_src = [&](int ch, const float*& src, const float*& sc, const float*& sh, int& C, int& cb) {
        cb = ch * 16;
        if (cb < a.C0) { src = a.src0; sc = a.sc0; sh = a.sh0; C = a.C0; }
        else { cb -= a.C0; src = a.src1; sc = a.sc1; sh = a.sh1; C = a.C1; }
    };
    auto prefetch = [&](int ch) {
        const float* src; const float* sc; const float* sh; int C, cb;
        chunk_src(ch, src, sc, sh, C, cb);
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            pv[it][0] = f32x4{0.f, 0.f, 0.f, 0.f}; pv[it][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (goff[it] >= 0) {
                const float* p = src + (size_t)goff[it] * C + cb + oct;
                pv[it][0] = *reinterpret_cast<const f32x4*>(p);
                pv[it][1] = *reinterpret_cast<const f32x4*>(p + 4);
            }
        }
    };

    if (active) prefetch(0);
    const int nslots = 2 * nchunks + 2;          // X: stage/MFMA in slots 0..2n-1, epilogue in 2n; Y one slot later
    for (int t = 0; t < nslots; ++t) {
        const int k = t - g;
        if (active && k >= 0 && k < 2 * nchunks) {
            const int ch = k >> 1;
            if ((k & 1) == 0) {
                // ================================ STAGE chunk ch: registers -> (norm, LeakyReLU, hi/lo split) -> LDS; weights
                const float* src; const float* sc; const float* sh; int C, cb;
                chunk_src(ch, src, sc, sh, C, cb);
                f32x4 s1a = f32x4{1.f, 1.f, 1.f, 1.f}, s1b = s1a, s2a = f32x4{0.f, 0.f, 0.f, 0.f}, s2b = s2a;
                if (sc != nullptr && a.lgNIMG == 0 && nimg0 < a.B) {
                    const size_t o = (size_t)nimg0 * C + cb + oct;
                    s1a = *reinterpret_cast<const f32x4*>(sc + o); s1b = *reinterpret_cast<const f32x4*>(sc + o + 4);
                    s2a = *reinterpret_cast<const f32x4*>(sh + o); s2b = *reinterpret_cast<const f32x4*>(sh + o + 4);
                }
#pragma unroll
                for (int it = 0; it < MAXU; ++it) {
                    const int u = tid + it * kBlock;
                    if (u < total) {
                        f32x4 va = pv[it][0], vb = pv[it][1];
                        if (sc != nullptr && goff[it] >= 0) {
                            if (a.lgNIMG != 0) {
                                const size_t o = (size_t)(nimg0 + (int)((imgbits >> (4 * it)) & 15)) * C + cb + oct;
                                s1a = *reinterpret_cast<const f32x4*>(sc + o); s1b = *reinterpret_cast<const f32x4*>(sc + o + 4);
                                s2a = *reinterpret_cast<const f32x4*>(sh + o); s2b = *reinterpret_cast<const f32x4*>(sh + o + 4);
                            }
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
                constexpr int WU = 9 * BN * 4;
                const uint4* wsrc = reinterpret_cast<const uint4*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * (9 * BN * 4);
#pragma unroll
                for (int it = 0; it < (WU + kBlock - 1) / kBlock; ++it) {
                    const int idx = tid + it * kBlock;
                    if (idx < WU) {
                        *reinterpret_cast<uint4*>(sB + (idx >> 2) * kRec + (idx & 3) * 16) = wsrc[idx];
                    }
                }
            } else {
                // ================================ MFMA chunk ch (the other group is staging meanwhile)
                if (ch + 1 < nchunks) prefetch(ch + 1);
                f32x16 acc_c[2][NT];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc_c[mt][nt][i] = 0.f;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int toff = ((tap / 3) * a.PW + (tap % 3)) * kRec;
                    half8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        ah[mt] = *reinterpret_cast<const half8*>(sA + abase[mt] + toff);
                        al[mt] = *reinterpret_cast<const half8*>(sA + abase[mt] + toff + 32);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        bh[nt] = *reinterpret_cast<const half8*>(sB + (tap * BN + nt * 32) * kRec + bbase);
                        bl[nt] = *reinterpret_cast<const half8*>(sB + (tap * BN + nt * 32) * kRec + bbase + 32);
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc_c[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc_c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc_c[mt][nt], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc_t[mt][nt] += acc_c[mt][nt];
            }
        } else if (active && k == 2 * nchunks) {
            // ================================ tile epilogue (X's overlaps Y's last MFMA slot)
            const float oscale = *a.oscale;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = n0col + nt * 32 + r;
                const float bv = a.bias[co];
                float ss = 0.f, qq = 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                        const int m = 64 * w + 32 * mt + row;
                        const int il = m >> (a.lgTH + a.lgTW), ty = (m >> a.lgTW) & (TH - 1), tx = m & (TW - 1);
                        const int n = nimg0 + il, oy = ty0 + ty, ox = tx0 + tx;
                        if (il < NIMG && n < a.B && oy < a.Ht && ox < a.Wt) {
                            const float v = acc_t[mt][nt][i] * oscale + bv;
                            a.dst[((size_t)(n * a.Ht + oy) * a.Wt + ox) * a.Cout + co] = v;
                            ss += v; qq += v * v;
                        }
                    }
                }
                ss += __shfl_xor(ss, 32); qq += __shfl_xor(qq, 32);
                if (h == 0) { red[(w * BN + nt * 32 + r) * 2] = ss; red[(w * BN + nt * 32 + r) * 2 + 1] = qq; }
            }
        }
        __syncthreads();
    }
    if (active && a.part != nullptr && tid < BN) {       // host guarantees NIMG == 1 when part != nullptr
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { s += red[(ww * BN + tid) * 2]; q += red[(ww * BN + tid) * 2 + 1]; }
        float* p = a.part + ((size_t)(nimg0 * tpi + tin) * a.Cout + n0col + tid) * 2;
        p[0] = s; p[1] = q;
    }
}

}  // namespace ts2d
