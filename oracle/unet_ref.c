/* ORACLE (test infrastructure only - never linked into or called by the product path).
 *
 * Plain-C CPU restatement of the reference's hot path: the forward pass of the nnU-Net 2-D PlainConvUNet that
 * `predictor.predict_logits_from_preprocessed_data` runs (reference call site
 * ts2d/core/inference/prediction_worker.py:209; network built at ts2d/core/inference/nnu.py:164-165 by the
 * third-party nnunetv2ml==2.6.2 / dynamic_network_architectures wheels, pyproject.toml:25, which are NOT in
 * /root/reference - their published algorithm is restated here).
 *
 * PARITY PINNING: the reference holds no golden vectors for this path.  This file is pinned against
 * fixtures produced by oracle/torch_oracle.py, which runs the same ATen CPU kernels the reference dispatches to
 * (tests/test_oracle.py).  Layout: NCHW fp32 like torch.  acc64 != 0 accumulates every dot product and every
 * InstanceNorm statistic in double ("truth" mode for error budgeting); acc64 == 0 is straight fp32 accumulation.
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int input_channels, num_classes, n_stages;
    int features[16];
    int n_conv_enc[16];
    int n_conv_dec[16];
    float eps, slope;
    int strides[16][2];      /* per stage (sy, sx), 1 or 2 each; stage 0 = (1, 1) */
} ts2d_ref_arch;

/* K1/K2/K3/K6: Conv2d 3x3, padding 1, stride (sy, s), bias.  x [Cin,Hi,Wi] -> y [Cout,Ho,Wo]; w [Cout,Cin,3,3]. */
static void conv3x3(const float* x, int cin, int hi, int wi, const float* w, const float* b, int cout, int sy, int s,
                    float* y, int acc64) {
    const int ho = hi / sy, wo = wi / s;
#pragma omp parallel for schedule(dynamic, 1)
    for (int co = 0; co < cout; ++co) {
        double* accd = acc64 ? (double*)malloc(sizeof(double) * ho * wo) : NULL;
        float* out = y + (size_t)co * ho * wo;
        for (int i = 0; i < ho * wo; ++i) { if (acc64) accd[i] = b[co]; else out[i] = b[co]; }
        for (int ci = 0; ci < cin; ++ci) {
            const float* xin = x + (size_t)ci * hi * wi;
            for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
                const float wv = w[(((size_t)co * cin + ci) * 3 + ky) * 3 + kx];
                for (int oy = 0; oy < ho; ++oy) {
                    const int iy = oy * sy + ky - 1;
                    if (iy < 0 || iy >= hi) continue;
                    /* ox range with 0 <= ox*s + kx - 1 < wi */
                    int ox0 = (kx == 0) ? 1 : 0;
                    int ox1 = wo;
                    while (ox1 > ox0 && (ox1 - 1) * s + kx - 1 >= wi) --ox1;
                    const float* row = xin + (size_t)iy * wi + kx - 1;
                    if (acc64) {
                        double* o = accd + (size_t)oy * wo;
                        for (int ox = ox0; ox < ox1; ++ox) o[ox] += (double)wv * (double)row[ox * s];
                    } else {
                        float* o = out + (size_t)oy * wo;
                        if (s == 1) for (int ox = ox0; ox < ox1; ++ox) o[ox] += wv * row[ox];
                        else        for (int ox = ox0; ox < ox1; ++ox) o[ox] += wv * row[ox * s];
                    }
                }
            }
        }
        if (acc64) { for (int i = 0; i < ho * wo; ++i) out[i] = (float)accd[i]; free(accd); }
    }
}

/* K4: InstanceNorm2d(affine, eps inside the sqrt, biased variance) + LeakyReLU, in place on [C,H,W]. */
static void inorm_lrelu(float* y, int c, int hw, const float* g, const float* be, float eps, float slope) {
#pragma omp parallel for
    for (int ch = 0; ch < c; ++ch) {
        float* p = y + (size_t)ch * hw;
        double s = 0;
        for (int i = 0; i < hw; ++i) s += p[i];
        const double mean = s / hw;
        double v = 0;
        for (int i = 0; i < hw; ++i) { const double d = p[i] - mean; v += d * d; }
        const double invstd = 1.0 / sqrt(v / hw + (double)eps);
        const float alpha = (float)(invstd * g[ch]);
        const float beta = (float)(be[ch] - mean * invstd * g[ch]);
        for (int i = 0; i < hw; ++i) { const float t = p[i] * alpha + beta; p[i] = t > 0 ? t : t * slope; }
    }
}

/* K5: ConvTranspose2d kernel = stride = (ka, kb) + bias; w [Cin,Cout,ka,kb]; x [Cin,Hi,Wi] -> y [Cout,ka Hi,kb Wi]. */
static void convT2x2(const float* x, int cin, int hi, int wi, const float* w, const float* b, int cout, int ka, int kb, float* y,
                     int acc64) {
    const int ho = ka * hi, wo = kb * wi;
#pragma omp parallel for
    for (int co = 0; co < cout; ++co) {
        float* out = y + (size_t)co * ho * wo;
        for (int iy = 0; iy < hi; ++iy) for (int ix = 0; ix < wi; ++ix) for (int a = 0; a < ka; ++a) for (int bb = 0; bb < kb; ++bb) {
            double accd = b[co]; float accf = b[co];
            for (int ci = 0; ci < cin; ++ci) {
                const float xv = x[((size_t)ci * hi + iy) * wi + ix];
                const float wv = w[(((size_t)ci * cout + co) * ka + a) * kb + bb];
                if (acc64) accd += (double)xv * wv; else accf += xv * wv;
            }
            out[(size_t)(ka * iy + a) * wo + kb * ix + bb] = acc64 ? (float)accd : accf;
        }
    }
}

/* K7: 1x1 conv + bias. */
static void conv1x1(const float* x, int cin, int hw, const float* w, const float* b, int cout, float* y, int acc64) {
#pragma omp parallel for
    for (int co = 0; co < cout; ++co) {
        float* out = y + (size_t)co * hw;
        for (int i = 0; i < hw; ++i) {
            double accd = b[co]; float accf = b[co];
            for (int ci = 0; ci < cin; ++ci) {
                if (acc64) accd += (double)x[(size_t)ci * hw + i] * w[(size_t)co * cin + ci];
                else accf += x[(size_t)ci * hw + i] * w[(size_t)co * cin + ci];
            }
            out[i] = acc64 ? (float)accd : accf;
        }
    }
}

/* A6: PlainConvUNet.forward for a batch.  blob = tensors in UNetArch.param_specs() order.  Returns 0 on success. */
int ts2d_ref_forward(const ts2d_ref_arch* a, const float* blob, const float* x, int B, int H, int W, float* logits,
                     int acc64) {
    const int n = a->n_stages;
    if (n < 2 || n > 16) return 1;
    {
        int dy = 1, dx = 1;
        for (int s = 1; s < n; ++s) {
            if (a->strides[s][0] < 1 || a->strides[s][0] > 2 || a->strides[s][1] < 1 || a->strides[s][1] > 2) return 1;
            dy *= a->strides[s][0]; dx *= a->strides[s][1];
        }
        if (H % dy || W % dx) return 1;
    }
    float* skip[16];
    for (int b = 0; b < B; ++b) {
        const float* p = blob;
        const float* cur = x + (size_t)b * a->input_channels * H * W;
        int cin = a->input_channels, h = H, w = W;
        float* tmp = NULL;
        for (int s = 0; s < n; ++s) {
            const int f = a->features[s];
            for (int i = 0; i < a->n_conv_enc[s]; ++i) {
                const int sty = (i == 0 && s > 0) ? a->strides[s][0] : 1, stx = (i == 0 && s > 0) ? a->strides[s][1] : 1;
                const int ho = h / sty, wo = w / stx;
                float* y = (float*)malloc(sizeof(float) * (size_t)f * ho * wo);
                const float* wt = p; p += (size_t)f * cin * 9;
                const float* bi = p; p += f;
                const float* g = p; p += f;
                const float* be = p; p += f;
                conv3x3(cur, cin, h, w, wt, bi, f, sty, stx, y, acc64);
                inorm_lrelu(y, f, ho * wo, g, be, a->eps, a->slope);
                if (tmp) free(tmp);
                tmp = y; cur = y; cin = f; h = ho; w = wo;
            }
            skip[s] = tmp; tmp = NULL;   /* keep stage output */
        }
        for (int j = 0; j < n - 1; ++j) {
            const int lvl = n - 2 - j, f = a->features[lvl];
            const int ka = a->strides[lvl + 1][0], kb = a->strides[lvl + 1][1];
            const float* wt = p; p += (size_t)cin * f * ka * kb;
            const float* bi = p; p += f;
            const int ho = ka * h, wo = kb * w;
            float* cat = (float*)malloc(sizeof(float) * (size_t)2 * f * ho * wo);
            convT2x2(cur, cin, h, w, wt, bi, f, ka, kb, cat, acc64);                                   /* up first ... */
            memcpy(cat + (size_t)f * ho * wo, skip[lvl], sizeof(float) * (size_t)f * ho * wo); /* ... then skip */
            if (tmp) free(tmp);
            tmp = cat; cur = cat; cin = 2 * f; h = ho; w = wo;
            for (int i = 0; i < a->n_conv_dec[j]; ++i) {
                float* y = (float*)malloc(sizeof(float) * (size_t)f * h * w);
                const float* cw = p; p += (size_t)f * cin * 9;
                const float* cb = p; p += f;
                const float* g = p; p += f;
                const float* be = p; p += f;
                conv3x3(cur, cin, h, w, cw, cb, f, 1, 1, y, acc64);
                inorm_lrelu(y, f, h * w, g, be, a->eps, a->slope);
                free(tmp);
                tmp = y; cur = y; cin = f;
            }
        }
        {
            const int k = a->num_classes;
            const float* hw_ = p; p += (size_t)k * cin;
            const float* hb = p; p += k;
            conv1x1(cur, cin, h * w, hw_, hb, k, logits + (size_t)b * k * H * W, acc64);
        }
        if (tmp) free(tmp);
        for (int s = 0; s < n; ++s) free(skip[s]);
    }
    return 0;
}

/* A7 (multilabel export): mask = sigmoid(float(logit)) > 0.5.  On the ATen CPU build the oracle was pinned with,
 * that predicate is exactly  logit > 1.5 * 2^-24  (exhaustive fp32 scan, tests/test_oracle.py::test_threshold). */
void ts2d_ref_mask(const float* logits, size_t n, uint8_t* mask) {
    for (size_t i = 0; i < n; ++i) mask[i] = logits[i] > 0x1.8p-24f;
}
