// Split-fp16 ConvTranspose2d 2x2 stride 2: a GEMM with M = input pixels, N = 4 * Cout ((a,b) tap-major),
// K = Cin, on v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi like kernels_f16x3.h).  No halo; chunks of 32 input
// channels; the epilogue scatters column n = (2a+b) * Cout + co of input pixel (y, x) to output pixel (2y+a, 2x+b).
// HBM-bound at high resolution (writes 4x the pixels it reads); the point of the split MFMA here is to take the
// arithmetic (8x64-cycle fp32 MFMAs per 32x32x16 block) off the critical path.
#pragma once
#include "kernels_f16x3.h"
#include "kernels_h32.h"

namespace ts2d {

template <int BN, typename ST = float, int NP = 3>
__global__ __launch_bounds__(kBlock, 2) void convT2x2_f16x3(const ConvArgs a) {
    constexpr int NT = BN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];

    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = bid >> 3;
    const int mtile = (q8 / a.n_ctiles) * 8 + xcd;
    const int ctile = q8 % a.n_ctiles;
    if (mtile >= a.n_mtiles) return;
    const int n0col = ctile * BN;

    const int NIMG = a.NIMG;
    const int tpi = a.tiles_x * a.tiles_y;
    const int grp = mtile / tpi, tin = mtile - grp * tpi;
    const int tyi = tin / a.tiles_x, txi = tin - tyi * a.tiles_x;
    const int nimg0 = grp * NIMG;
    const int ty0 = tyi * a.TH, tx0 = txi * a.TW;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int P = a.TH * a.TW * NIMG;                 // <= 256 tile pixels, no halo
    unsigned char* sA = smem8;                        // [kk 2][P][80 B]
    unsigned char* sB = smem8 + 2 * P * kRec;         // [kk 2][BN][80 B]

    // staging unit u = (pixel u >> 2, channel octet u & 3): 4 units per thread, pixel = tile row m
    int goff[4], nimg[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int u = tid + it * kBlock;
        const int m = u >> 2;
        int g = -1, n = 0;
        if (m < P) {
            int il, ty, tx;
            tile_row(a, m, il, ty, tx);
            n = nimg0 + il;
            const int iy = ty0 + ty, ix = tx0 + tx;
            if (il < NIMG && n < a.B && iy < a.Hin && ix < a.Win) g = (n * a.Hin + iy) * a.Win + ix;
        }
        goff[it] = g; nimg[it] = n;
    }
    const int oct = (tid & 3) * 8;                    // kBlock % 4 == 0: the octet is the same for all 4 units

    int abase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) abase[mt] = (64 * w + 32 * mt + r) * kRec + 16 * h;      // record = tile row m
    const int bbase = r * kRec + 16 * h;

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const int nchunks = a.C0 / 32;
    Raw8<ST> pv[4];
    auto prefetch = [&](int ch) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            pv[it].zero();
            if (goff[it] >= 0) pv[it].load(a.src0, (size_t)goff[it] * a.C0 + ch * 32 + oct);
        }
    };

    prefetch(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int u = tid + it * kBlock;
            const int m = u >> 2;
            if (m < P) {
                f32x4 va, vb;
                pv[it].get(va, vb);
                if (a.sc0 != nullptr && goff[it] >= 0) {
                    const size_t o = (size_t)nimg[it] * a.C0 + ch * 32 + oct;
                    const f32x4 s1a = *reinterpret_cast<const f32x4*>(a.sc0 + o), s1b = *reinterpret_cast<const f32x4*>(a.sc0 + o + 4);
                    const f32x4 s2a = *reinterpret_cast<const f32x4*>(a.sh0 + o), s2b = *reinterpret_cast<const f32x4*>(a.sh0 + o + 4);
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
                    if (NP == 3) { lo[e] = (_Float16)(va[e] - (float)ha); lo[e + 4] = (_Float16)(vb[e] - (float)hb); }
                }
                // octet o8 = u & 3: k-step kk = o8 >> 1, half of the 16-channel record = o8 & 1
                unsigned char* d = sA + (((u & 3) >> 1) * P + m) * kRec + (u & 1) * 16;
                *reinterpret_cast<half8*>(d) = hi;
                if (NP == 3) *reinterpret_cast<half8*>(d + 32) = lo;
            }
        }
        {
            constexpr int WU = 2 * BN * 4;
            // one contiguous [k-step][column][16 hi | 16 lo] block per (chunk, column tile)
            const uint4* wsrc = reinterpret_cast<const uint4*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * WU;
#pragma unroll
            for (int it = 0; it < (WU + kBlock - 1) / kBlock; ++it) {
                const int idx = tid + it * kBlock;
                if ((WU % kBlock == 0 || idx < WU) && (NP == 3 || (idx & 2) == 0))
                    *reinterpret_cast<uint4*>(sB + (idx >> 2) * kRec + (idx & 3) * 16) = wsrc[idx];
            }
        }
        __syncthreads();
        if (ch + 1 < nchunks) prefetch(ch + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                ah[mt] = *reinterpret_cast<const half8*>(sA + kk * P * kRec + abase[mt]);
                if (NP == 3) al[mt] = *reinterpret_cast<const half8*>(sA + kk * P * kRec + abase[mt] + 32);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bh[nt] = *reinterpret_cast<const half8*>(sB + (kk * BN + nt * 32) * kRec + bbase);
                if (NP == 3) bl[nt] = *reinterpret_cast<const half8*>(sB + (kk * BN + nt * 32) * kRec + bbase + 32);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    { if (NP == 3) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0); }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    { if (NP == 3) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0); }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        }
    }

    const float oscale = *a.oscale;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ab = (n0col + nt * 32) / a.Cout;
        const int co = n0col + nt * 32 - ab * a.Cout + r;
        const int oa = ab >> 1, ob = ab & 1;
        const float bv = a.bias[co];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                const int m = 64 * w + 32 * mt + row;
                int il, ty, tx;
                tile_row(a, m, il, ty, tx);
                const int n = nimg0 + il, oy = ty0 + ty, ox = tx0 + tx;
                if (il < NIMG && n < a.B && oy < a.Ht && ox < a.Wt)
                    store_act<ST>(a.dst, ((size_t)(n * 2 * a.Ht + 2 * oy + oa) * (2 * a.Wt) + 2 * ox + ob) * a.Cout + co,
                                  acc[mt][nt][i] * oscale + bv);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// One-image-tile variant (every level with >= 16x16 input pixels): buffer loads relative to the image base, scale/shift loaded
// once per chunk (the staging octet is per-thread constant) and prefetched with the patch, weight loads issued ahead of the
// conversion, and a scatter epilogue with one lane offset per 32x32 block + scalar row offsets when the tile is complete.
// 64 output columns per workgroup; same arithmetic and summation order as convT2x2_f16x3.
// ------------------------------------------------------------------------------------------------------------
template <typename ST, int NP>
__global__ __launch_bounds__(kBlock, 2) void convT2x2_f16x3_one(const ConvArgs a) {
    constexpr int BN = 64, NT = 2, MAXU = 4, NL = sizeof(ST) == 4 ? 2 : 1;      // NL: 16-byte loads per staging unit (8 channels)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];

    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = bid >> 3;
    const int qm = q8 / a.n_ctiles;                 // (any tile count, any tile shape: round 5)
    const int mtile = qm * 8 + xcd;
    const int ctile = q8 - qm * a.n_ctiles;
    if (mtile >= a.n_mtiles) return;
    const int n0col = ctile * BN;

    const int TH = a.TH, TW = a.TW;
    const int tpi = a.tiles_x * a.tiles_y;
    const int nimg0 = mtile / tpi, tin = mtile - nimg0 * tpi;
    const int tyi = tin / a.tiles_x, txi = tin - tyi * a.tiles_x;
    const int ty0 = tyi * TH, tx0 = txi * TW;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int P = TH * TW;                            // <= 256 tile pixels, no halo
    unsigned char* sA = smem8;                        // [kk 2][P][80 B]
    unsigned char* sB = smem8 + 2 * P * kRec;         // [kk 2][BN][80 B]

    // staging unit u = (pixel u >> 2, channel octet u & 3): 4 units per thread, the octet is per-thread constant
    const int oct = (tid & 3) * 8;
    unsigned vo[MAXU];                                // byte offset inside the image, or out of range (pixel outside the image)
    unsigned inside = 0;
#pragma unroll
    for (int it = 0; it < MAXU; ++it) {
        const int u = tid + it * kBlock, m = u >> 2;
        int il, tyy, txx;
        tile_row(a, m, il, tyy, txx);
        const int iy = ty0 + tyy, ix = tx0 + txx;
        const bool in = il == 0 && iy < a.Hin && ix < a.Win;
        vo[it] = in ? (unsigned)(((iy * a.Win + ix) * a.C0 + oct) * (int)sizeof(ST)) : 0x80000000u;
        inside |= (in ? 1u : 0u) << it;
        if (!in && m < P) {                           // never staged: zero once
            unsigned char* d = sA + (((u & 3) >> 1) * P + m) * kRec + (u & 1) * 16;
            *reinterpret_cast<uint4*>(d) = uint4{0u, 0u, 0u, 0u};
            if (NP == 3) *reinterpret_cast<uint4*>(d + 32) = uint4{0u, 0u, 0u, 0u};
        }
    }
    int abase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) abase[mt] = (64 * w + 32 * mt + r) * kRec + 16 * h;      // record = tile row m
    const int bbase = r * kRec + 16 * h;

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const int nchunks = a.C0 / 32;
    const size_t img_el = (size_t)a.Hin * a.Win * a.C0;
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(reinterpret_cast<const ST*>(a.src0)) + (size_t)nimg0 * img_el, 0,
                                                       (int)(img_el * sizeof(ST)), 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pv[MAXU][NL];
    f32x4 nsa = f32x4{1.f, 1.f, 1.f, 1.f}, nsb = nsa, nta = f32x4{0.f, 0.f, 0.f, 0.f}, ntb = nta;
    const bool normed = a.sc0 != nullptr;
    const _Float16 slope_h = (_Float16)a.slope;
    const unsigned slope2 = (unsigned)__builtin_bit_cast(unsigned short, slope_h) * 0x10001u;
    auto prefetch = [&](int ch) {
#pragma unroll
        for (int it = 0; it < MAXU; ++it)
#pragma unroll
            for (int l = 0; l < NL; ++l) pv[it][l] = __builtin_amdgcn_raw_buffer_load_b128(rs0, vo[it] + 16 * l, ch * 32 * (int)sizeof(ST), 0);
        if (normed) {
            const float* ps = a.sc0 + (size_t)nimg0 * a.C0 + ch * 32 + oct; const float* pt = a.sh0 + (size_t)nimg0 * a.C0 + ch * 32 + oct;
            nsa = *reinterpret_cast<const f32x4*>(ps); nsb = *reinterpret_cast<const f32x4*>(ps + 4);
            nta = *reinterpret_cast<const f32x4*>(pt); ntb = *reinterpret_cast<const f32x4*>(pt + 4);
        }
    };

    prefetch(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
        constexpr int WU = 2 * BN * 4;                // 512 uint4 = 2 per thread
        const uint4* wsrc = reinterpret_cast<const uint4*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * WU;
        uint4 w0, w1;
        if (NP == 3 || (tid & 2) == 0) { w0 = wsrc[tid]; w1 = wsrc[tid + kBlock]; }
#pragma unroll
        for (int it = 0; it < MAXU; ++it) {
            if ((inside >> it) & 1u) {
                const int u = tid + it * kBlock, m = u >> 2;
                unsigned char* d = sA + (((u & 3) >> 1) * P + m) * kRec + (u & 1) * 16;
                if constexpr (sizeof(ST) == 4) {
                    f32x4 va = __builtin_bit_cast(f32x4, pv[it][0]), vb = __builtin_bit_cast(f32x4, pv[it][NL - 1]);
                    if (normed) {
                        va = va * nsa + nta; vb = vb * nsb + ntb;
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
                        if (NP == 3) { lo[e] = (_Float16)(va[e] - (float)ha); lo[e + 4] = (_Float16)(vb[e] - (float)hb); }
                    }
                    *reinterpret_cast<half8*>(d) = hi;
                    if (NP == 3) *reinterpret_cast<half8*>(d + 32) = lo;
                } else {
                    uint4 x = uint4{pv[it][0][0], pv[it][0][1], pv[it][0][2], pv[it][0][3]};
                    if (normed) x = norm_lrelu_8(x, nsa, nsb, nta, ntb, slope2);
                    *reinterpret_cast<uint4*>(d) = x;
                }
            }
        }
        if (NP == 3 || (tid & 2) == 0) {
            *reinterpret_cast<uint4*>(sB + (tid >> 2) * kRec + (tid & 3) * 16) = w0;
            *reinterpret_cast<uint4*>(sB + ((tid + kBlock) >> 2) * kRec + (tid & 3) * 16) = w1;
        }
        __syncthreads();
        if (ch + 1 < nchunks) prefetch(ch + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                ah[mt] = *reinterpret_cast<const half8*>(sA + kk * P * kRec + abase[mt]);
                if (NP == 3) al[mt] = *reinterpret_cast<const half8*>(sA + kk * P * kRec + abase[mt] + 32);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bh[nt] = *reinterpret_cast<const half8*>(sB + (kk * BN + nt * 32) * kRec + bbase);
                if (NP == 3) bl[nt] = *reinterpret_cast<const half8*>(sB + (kk * BN + nt * 32) * kRec + bbase + 32);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    { if (NP == 3) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0); }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    { if (NP == 3) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0); }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        }
    }

    const float oscale = *a.oscale;
    const bool full = a.lgTW >= 4 && a.lgTH + a.lgTW == 8 && ty0 + TH <= a.Ht && tx0 + TW <= a.Wt;     // wave-uniform (power-of-two tiles only)
    const size_t out_el = (size_t)4 * a.Ht * a.Wt * a.Cout;
    const auto rsd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<ST*>(a.dst) + (size_t)nimg0 * out_el, 0, (int)(out_el * sizeof(ST)), 0x00020000);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ab = (n0col + nt * 32) / a.Cout;
        const int co = n0col + nt * 32 - ab * a.Cout + r;
        const int oa = ab >> 1, ob = ab & 1;
        const float bv = a.bias[co];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (full) {
                const int m0 = 64 * w + 32 * mt + 4 * h;
                const int oy = ty0 + (m0 >> a.lgTW), ox = tx0 + (m0 & (TW - 1));
                const unsigned voff = (unsigned)((((2 * oy + oa) * (2 * a.Wt) + 2 * ox + ob) * a.Cout + co) * (int)sizeof(ST));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rowoff = (i & 3) + 8 * (i >> 2);
                    const unsigned soff = (unsigned)(((2 * (rowoff & (TW - 1)) + (rowoff >> a.lgTW) * 4 * a.Wt) * a.Cout) * (int)sizeof(ST));   // scalar
                    buffer_store_act<ST>(acc[mt][nt][i] * oscale + bv, rsd, voff, soff);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                    const int m = 64 * w + 32 * mt + row;
                    int il, tyy, txx;
                    tile_row(a, m, il, tyy, txx);
                    const int oy = ty0 + tyy, ox = tx0 + txx;
                    if (il == 0 && oy < a.Ht && ox < a.Wt)
                        store_act<ST>(a.dst, ((size_t)(nimg0 * 2 * a.Ht + 2 * oy + oa) * (2 * a.Wt) + 2 * ox + ob) * a.Cout + co,
                                      acc[mt][nt][i] * oscale + bv);
                }
            }
        }
    }
}

}  // namespace ts2d
