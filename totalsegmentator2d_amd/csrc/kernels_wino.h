// Stride-1 3x3 convolution as Winograd F(2x2, 3x3) on the split-fp16 matrix path (the layers that hold 83 % of
// the FLOPs).  Y = A^T [ (G g G^T) . (B^T d B) ] A: per 2x2 output tile 16 multiplies per (ci, co) instead of 36 - 2.25x fewer
// MFMAs than the direct implicit GEMM.  Accuracy (scripts/winograd_accuracy_experiment.py, canonical net, CPU emulation of this
// arithmetic): 4.9e-5 from the fp64-accumulating truth, 6.0e-5 from the torch fp32 oracle (direct split conv: 4.6e-5 / 5.7e-5;
// ATen itself is 4.8e-5 from the truth) - inside the 1e-4 budget because
//   * the filter transform U = G g G^T is done on the HOST in fp64 and split into fp16 hi + lo (22 bits) like every weight,
//   * the input transform V = B^T d B (additions only) runs in fp32 on the normalised, activated patch and V is split hi/lo,
//   * the 16 per-position products accumulate in fp32 on the MFMA (3 products: Ulo*Vhi + Uhi*Vlo + Uhi*Vhi),
//   * the output transform (additions only) runs in fp32.
// One 512-thread workgroup per CU owns a 256-pixel tile (8 x 32 = 64 Winograd tiles) x 64 output channels; wave w accumulates
// positions 2w and 2w+1 for ALL 64 tiles x 64 channels (128 accumulator registers), so the [position][tile][channel] products
// never leave the register file until the single output transform at the end.  Per 16-channel chunk:
//   stage 1  raw fp32 patch (prefetched one chunk ahead) -> InstanceNorm + LeakyReLU -> LDS image R [pixel][16 ch] fp32
//   stage 2  thread (tile, channel pair): 4x4 window of R -> B^T d B -> hi/lo -> k-group-major planes V[position][part][h][tile]
//            (16-byte slots, conflict-free for the 32x32x16 fragment reads); the chunk's pre-transformed weights U (one linear
//            64-KB block per (chunk, column tile), stored in HBM in LDS order) are copied behind it
//   MFMA     per position 2 x 2 tiles of 32x32x16, 3 products: 24 MFMAs per wave and chunk (the direct kernel: 108 for half the pixels)
// LDS: R 27 KB + V 68 KB + U 64 KB = 159 KB.
#pragma once
#include "kernels_f16x3.h"

namespace ts2d {

constexpr int kWThreads = 512;
constexpr int kWPW = 34, kWP = 340;                  // patch 10 x 34 pixels
constexpr int kWRpx = 80;                            // bytes per patch pixel in R (16 fp32 + 16 B pad: spreads the b64 window reads)
constexpr int kWR = kWP * kWRpx;                     // 27,200
constexpr int kWVplane = 64 * 16 + 64;               // 64 tiles x 16 B, planes skewed by 64 B (conflict-free ds_write_b32 of stage 2)
constexpr int kWV = 64 * kWVplane;                   // planes p = (position * 2 + part) * 2 + h
constexpr int kWU = 16 * 2 * 2 * 64 * 16;            // [position][part][h][column 64] x 16 B
constexpr int kWLds = kWR + kWV + kWU;               // 162,368

__global__ __launch_bounds__(kWThreads, 2) void conv3x3_wino(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));

    // ---- block -> (pixel tile, column tile).  Blocks b and b + 8 share an XCD and its L2.  The per-chunk operand that every
    //      workgroup re-reads is the 64-KB weight block U(chunk, column tile) - larger than the 22-KB activation patch - so the
    //      blocks of one XCD work on the SAME column tile (different pixel tiles): an XCD streams 1/8 of the layer's U through
    //      its L2 instead of all of it.  (a.dbg & 16: the pixel-tile-major map of the direct kernels, for comparison.)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = bid >> 3;
    int mtile, ctile;
    if (a.dbg & 16) { const int qm = q8 >> a.lg_nct; mtile = qm * 8 + xcd; ctile = q8 - qm * a.n_ctiles; }
    else if (a.n_ctiles >= 8) { const int per = a.n_ctiles >> 3; ctile = xcd + 8 * (q8 & (per - 1)); mtile = q8 >> (a.lg_nct - 3); }
    else { const int rep = 8 >> a.lg_nct; ctile = xcd & (a.n_ctiles - 1); mtile = q8 * rep + (xcd >> a.lg_nct); }
    if (mtile >= a.n_mtiles) return;
    const int n0col = ctile * 64;
    const int tpi = a.tiles_x * a.tiles_y;
    const int nimg0 = mtile >> a.lg_tpi, tin = mtile - nimg0 * tpi;
    const int tyi = tin >> a.lg_tx, txi = tin - tyi * a.tiles_x;
    const int ty0 = tyi * 8, tx0 = txi * 32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    unsigned char* sR = smem8;
    unsigned char* sU = smem8 + kWR + kWV;

    // ---- stage-1 plan: unit = (patch pixel, channel quad); padding pixels are zeroed once and never staged
    constexpr int NU1 = 3;                                 // 3 x 512 units >= 340 x 4
    unsigned vo0[NU1], vo1[NU1];                           // byte offset of the unit's 4 channels in either source image
    const int quad = tid & 3;
#pragma unroll
    for (int it = 0; it < NU1; ++it) {
        const int u = tid + it * kWThreads, pp = u >> 2;
        unsigned v0 = 0x80000000u, v1 = 0x80000000u;
        if (pp < kWP) {
            const int py = pp / kWPW, px = pp - py * kWPW;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) {
                v0 = (unsigned)(((iy * a.Win + ix) * a.C0 + 4 * quad) * 4);
                v1 = (unsigned)(((iy * a.Win + ix) * a.C1 + 4 * quad) * 4);
            } else *reinterpret_cast<uint4*>(sR + pp * kWRpx + quad * 16) = uint4{0u, 0u, 0u, 0u};
        }
        vo0[it] = v0; vo1[it] = v1;
    }
    const size_t img_px = (size_t)a.Hin * a.Win;
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0) + (size_t)nimg0 * img_px * a.C0, 0, (int)(img_px * a.C0 * 4), 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1 ? a.src1 : a.src0) + (size_t)nimg0 * img_px * a.C1, 0,
                                                       (int)(a.src1 ? img_px * a.C1 * 4 : 0), 0x00020000);
    const int nchunks = (a.C0 + a.C1) / 16;
    u32x4 pv[NU1];
    auto prefetch = [&](int ch) {
        int cb = ch * 16;
        if (cb < a.C0) {
#pragma unroll
            for (int it = 0; it < NU1; ++it) pv[it] = __builtin_amdgcn_raw_buffer_load_b128(rs0, vo0[it], cb * 4, 0);
        } else {
            cb -= a.C0;
#pragma unroll
            for (int it = 0; it < NU1; ++it) pv[it] = __builtin_amdgcn_raw_buffer_load_b128(rs1, vo1[it], cb * 4, 0);
        }
    };
    prefetch(0);

    // ---- stage-2 plan: thread = (Winograd tile, channel pair)
    const int tile = tid >> 3, cp = tid & 7;
    const int rbase = ((2 * (tile >> 4)) * kWPW + 2 * (tile & 15)) * kWRpx + cp * 8;            // + (i * 34 + j) * 80
    const int vbase = kWR + (cp >> 2) * kWVplane + tile * 16 + (cp & 3) * 4;                    // + (position * 2 + part) * 2 * plane
    // ---- MFMA plan: wave w owns positions 2w, 2w + 1
    const int abase = kWR + ((4 * w) * 2 + h) * kWVplane + r * 16;     // + (pos * 2 + part) * 2 * plane + mt * 512
    const int bbase = kWR + kWV + (((4 * w) * 2 + h) * 64 + r) * 16;   // + (pos * 2 + part) * 2 * 1024 + nt * 512

    f32x16 acc[2][2][2];                                   // [position][pixel-tile block][column block]
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[p][mt][nt][i] = 0.f;

    for (int ch = 0; ch < nchunks; ++ch) {
        // ---- weights U of this chunk: loads issued first (named registers), written behind stage 2
        const uint4* wsrc = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(a.wph) + ((size_t)ch * a.n_ctiles + ctile) * kWU);
        const uint4 w0 = wsrc[tid], w1 = wsrc[tid + 512], w2 = wsrc[tid + 1024], w3 = wsrc[tid + 1536],
                    w4 = wsrc[tid + 2048], w5 = wsrc[tid + 2560], w6 = wsrc[tid + 3072], w7 = wsrc[tid + 3584];
        // ---- stage 1: InstanceNorm + LeakyReLU of the raw patch -> R (fp32).  R is not read by the MFMA phase, so a wave that
        //      finishes its MFMAs early converts while the others still multiply.
        {
            int cb = ch * 16;
            f32x4 ns = f32x4{1.f, 1.f, 1.f, 1.f}, nt4 = f32x4{0.f, 0.f, 0.f, 0.f};
            bool normed;
            if (cb < a.C0) {
                normed = a.sc0 != nullptr;
                if (normed) { ns = *reinterpret_cast<const f32x4*>(a.sc0 + (size_t)nimg0 * a.C0 + cb + 4 * quad);
                              nt4 = *reinterpret_cast<const f32x4*>(a.sh0 + (size_t)nimg0 * a.C0 + cb + 4 * quad); }
            } else {
                cb -= a.C0;
                normed = a.sc1 != nullptr;
                if (normed) { ns = *reinterpret_cast<const f32x4*>(a.sc1 + (size_t)nimg0 * a.C1 + cb + 4 * quad);
                              nt4 = *reinterpret_cast<const f32x4*>(a.sh1 + (size_t)nimg0 * a.C1 + cb + 4 * quad); }
            }
#pragma unroll
            for (int it = 0; it < NU1; ++it) {
                if (vo0[it] != 0x80000000u && !(a.dbg & 8)) {
                    f32x4 v = __builtin_bit_cast(f32x4, pv[it]);
                    if (normed) {
                        v = v * ns + nt4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * a.slope);      // LeakyReLU (0 < slope < 1)
                    }
                    *reinterpret_cast<f32x4*>(sR + ((tid + it * kWThreads) >> 2) * kWRpx + quad * 16) = v;
                }
            }
        }
        __syncthreads();                                   // R complete; every wave is done with the previous chunk's V and U
        // ---- stage 2: V = B^T d B of this thread's tile and channel pair, split hi / lo, written k-group major
        if (!(a.dbg & 2)) {
            f32x2 d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f32x2*>(sR + rbase + (i * kWPW + j) * kWRpx);
#pragma unroll
            for (int j = 0; j < 4; ++j) {                  // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
                const f32x2 t0 = d[0][j] - d[2][j], t1 = d[1][j] + d[2][j], t2 = d[2][j] - d[1][j], t3 = d[1][j] - d[3][j];
                d[0][j] = t0; d[1][j] = t1; d[2][j] = t2; d[3][j] = t3;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {                  // (.) B: columns, same pattern
                const f32x2 t0 = d[i][0] - d[i][2], t1 = d[i][1] + d[i][2], t2 = d[i][2] - d[i][1], t3 = d[i][1] - d[i][3];
                d[i][0] = t0; d[i][1] = t1; d[i][2] = t2; d[i][3] = t3;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned hi, lo;
                    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
                                 "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                                 "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                                 : "=&v"(hi), "=&v"(lo) : "v"(d[i][j][0]), "v"(d[i][j][1]));
                    const int xi = i * 4 + j;
                    *reinterpret_cast<unsigned*>(smem8 + vbase + (xi * 2 + 0) * 2 * kWVplane) = hi;
                    *reinterpret_cast<unsigned*>(smem8 + vbase + (xi * 2 + 1) * 2 * kWVplane) = lo;
                }
        }
        if (!(a.dbg & 4)) {
        *reinterpret_cast<uint4*>(sU + (tid) * 16) = w0;          *reinterpret_cast<uint4*>(sU + (tid + 512) * 16) = w1;
        *reinterpret_cast<uint4*>(sU + (tid + 1024) * 16) = w2;   *reinterpret_cast<uint4*>(sU + (tid + 1536) * 16) = w3;
        *reinterpret_cast<uint4*>(sU + (tid + 2048) * 16) = w4;   *reinterpret_cast<uint4*>(sU + (tid + 2560) * 16) = w5;
        *reinterpret_cast<uint4*>(sU + (tid + 3072) * 16) = w6;   *reinterpret_cast<uint4*>(sU + (tid + 3584) * 16) = w7;
        }
        __syncthreads();                                   // V and U complete
        if (ch + 1 < nchunks) prefetch(ch + 1);            // HBM latency hides behind the MFMA phase

        // ---- 2 positions x (2 x 2 tiles of 32x32x16) x 3 products
        __builtin_amdgcn_s_setprio(1);
        if (!(a.dbg & 1)) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            half8 fa[2][2], fb[2][2];                      // [block][hi, lo]
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q = 0; q < 2; ++q) fa[mt][q] = *reinterpret_cast<const half8*>(smem8 + abase + (p * 2 + q) * 2 * kWVplane + mt * 512);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 2; ++q) fb[nt][q] = *reinterpret_cast<const half8*>(smem8 + bbase + (p * 2 + q) * 2 * 1024 + nt * 512);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[p][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][1], fb[nt][0], acc[p][mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[p][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][0], fb[nt][1], acc[p][mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[p][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][0], fb[nt][0], acc[p][mt][nt], 0, 0, 0);
        }
        }
        __builtin_amdgcn_s_setprio(0);
    }

    // ---- output transform Y = A^T M A through LDS, 32 output channels per pass: M[position][tile][32] fp32 = 128 KB
    const float oscale = *a.oscale;
    const size_t img_el = (size_t)a.Ht * a.Wt * a.Cout;
    const auto rsd = __builtin_amdgcn_make_buffer_rsrc(a.dst + (size_t)nimg0 * img_el, 0, (int)(img_el * 4), 0x00020000);
    float* sM = reinterpret_cast<float*>(smem8);
    float* sS = reinterpret_cast<float*>(smem8 + 16 * 64 * 32 * 4);        // statistics scratch [pass 2][wave 8][32 ch][2]
    const int cq = tid & 7;                                // this thread's channel quad in the pass: channels 4 cq .. 4 cq + 3
    const int oty = tile >> 4, otx = tile & 15;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        __syncthreads();                                   // LDS free (last MFMA reads / previous pass reads done)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                    sM[(((2 * w + p) * 64) + 32 * mt + row) * 32 + r] = acc[p][mt][nt][i];
                }
        __syncthreads();
        f32x4 m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const f32x4*>(sM + ((i * 4 + j) * 64 + tile) * 32 + 4 * cq);
        f32x4 t[2][4];                                     // A^T M: rows (m0 + m1 + m2, m1 - m2 - m3)
#pragma unroll
        for (int j = 0; j < 4; ++j) { t[0][j] = m[0][j] + m[1][j] + m[2][j]; t[1][j] = m[1][j] - m[2][j] - m[3][j]; }
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n0col + 32 * nt + 4 * cq);
        f32x4 y[2][2];
        f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f}, q4 = s4;
#pragma unroll
        for (int aa = 0; aa < 2; ++aa) {
            const f32x4 y0 = t[aa][0] + t[aa][1] + t[aa][2], y1 = t[aa][1] - t[aa][2] - t[aa][3];
#pragma unroll
            for (int e = 0; e < 4; ++e) { y[aa][0][e] = __builtin_fmaf(y0[e], oscale, bv[e]); y[aa][1][e] = __builtin_fmaf(y1[e], oscale, bv[e]); }
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int oy = ty0 + 2 * oty + aa, ox = tx0 + 2 * otx + bb;
                // (offset in the VGPR, soffset = 0: see the wide-store hazard note in kernels_res32.h)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y[aa][bb]), rsd,
                                                       (unsigned)(((oy * a.Wt + ox) * a.Cout + n0col + 32 * nt + 4 * cq) * 4), 0, 0);
                s4 += y[aa][bb]; q4 += y[aa][bb] * y[aa][bb];
            }
        }
        // sum over the 8 tiles of this wave that share the channel quad (lanes differing in bits 3..5), then across waves via LDS
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = s4[e], q = q4[e];
            s += __shfl_xor(s, 8); q += __shfl_xor(q, 8); s += __shfl_xor(s, 16); q += __shfl_xor(q, 16);
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
            if (lane < 8) { sS[((nt * 8 + w) * 32 + 4 * cq + e) * 2] = s; sS[((nt * 8 + w) * 32 + 4 * cq + e) * 2 + 1] = q; }
        }
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) asm volatile("" :: "v"(y[aa][bb]));      // store data registers untouched up to here
    }
    __syncthreads();
    if (tid < 128) {                                       // (channel 0..63, sum | sum of squares), fixed order over the 8 waves
        const int c = tid >> 1, which = tid & 1;
        float tot = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) tot += sS[(((c >> 5) * 8 + ww) * 32 + (c & 31)) * 2 + which];
        a.part[((size_t)(nimg0 * tpi + tin) * a.Cout + n0col + c) * 2 + which] = tot;
    }
}

}  // namespace ts2d
