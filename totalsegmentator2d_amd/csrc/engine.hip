// libts2d_engine.so - host side of the MI355X-native 2-D U-Net engine + the C-ABI of include/ts2d_engine.h.
// Replaces the network-forward part of nnUNetPredictor (reference seam: ts2d/core/inference/prediction_worker.py:209,
// network built at ts2d/core/inference/nnu.py:164-165).  gfx950 only; no CPU fallback exists in this library.
#include "../../include/ts2d_engine.h"
#include "kernels.h"
#include "kernels_f16x3.h"
#include "kernels_f16x3_convt.h"
#include "kernels_h32.h"
#include "kernels_head.h"
#include "kernels_f16x3_one.h"
#include "kernels_f16x3_qp.h"
#include "kernels_first.h"
#include "kernels_res32.h"
#include "kernels_s2v2.h"
#include "kernels_upc.h"
#include "kernels_up0.h"
#include "kernels_upq.h"
#include "kernels_upc_h.h"
#include "kernels_upc_h2.h"
#include "kernels_sw.h"
#include "kernels_project.h"

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace ts2d;

namespace {

thread_local std::string g_err;
// Generation counter of runs inside caller-provided (shared) workspaces: every such run stamps the workspace header with a fresh token
// and remembers it; an engine whose token is no longer in the header knows that ANOTHER engine of the set has overwritten its
// activations since (ts2d_engine_debug_tensor / ts2d_engine_check must not read them as its own - ADVICE r4).
std::atomic<uint32_t> g_ws_generation{0};
constexpr size_t kWsHeader = 256;         // bytes in front of the activation plan: [0] = token of the last run inside this memory

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return fail(_e == hipErrorOutOfMemory ? TS2D_ERR_NOMEM : TS2D_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(_e), __FILE__, __LINE__);                                \
    } while (0)

enum OpType { OP_CONV = 0, OP_CONVT = 1, OP_HEAD = 2 };

struct Tensor {            // an activation tensor of the program (NHWC fp32)
    std::string name;
    int C = 0, level = 0;
    int Cu = 0;            // channels of the CALLER's architecture (C is the width the kernels run: rounded up to 32 - pad_arch)
    int ly = 0, lx = 0;    // log2 of the cumulative stride along H / W: extent (H >> ly, W >> lx) (per-axis strides, ABI 7)
    bool normed = false;   // raw conv output that carries InstanceNorm scale/shift
    float* data = nullptr; float* scale = nullptr; float* shift = nullptr;
    bool resident = false; // its buffer still holds the values of the last run (no later tensor of the run was placed on it)
};

struct Op {
    OpType type; std::string name;
    int src, skip, dst;           // tensor indices (skip = -1 if none)
    int cin, cin_skip, cout, stride, level;      // stride: 1 = (1, 1), 2 = (2, 2) - the dedicated kernels; 3 = anisotropic (generic kernel only)
    int sy = 1, sx = 1;           // stride along H / W (conv: of the conv; transposed conv: kernel = stride of the upsampling)
    int ly = 0, lx = 0;           // shifts of the op's level (= of its output tensor; the head: level 0)
    int ck;                       // Cin chunk of the packed weight layout
    size_t blob_w, blob_b, blob_g, blob_be;     // offsets (floats) into the PyTorch-layout blob
    size_t dev_w, dev_b, dev_g, dev_be;         // offsets (floats) into the device weight arena
    size_t dev_w_floats;
    bool split_ok = false;        // eligible for the split-fp16 kernel (3x3, stride 1, Cin % 16 == 0, not the net input)
    size_t dev_wh = 0;            // offset (floats) of the split-fp16 weights in the arena
    size_t dev_wh32 = 0;          // offset (floats) of the fp16 weights in 32-channel chunks (precision mode f16, kernels_h32.h)
    bool h32_ok = false;
    size_t dev_ws = 0;            // offset (floats) of 1 / (power-of-two pre-scale of the split weights)
    size_t dev_wp = 0;            // offset (floats) of the split-fp16 weights in plane order [chunk16][column tile][tap][hi,lo][h][column][8 halves]
                                  // (conv3x3_f16x3_qp; stride-1 split_ok convs)
    bool s2v2_ok = false;         // stride-2 block: 512-thread kernel of kernels_s2v2.h
    size_t dev_w2 = 0;            // offset (floats) of its weight image [chunk16][column tile][tap][hi,lo][h][column][8 halves]
    int bn2 = 0;                  // its column tile (128 or 64)
    bool res_ok = false;          // 32 -> 32 stride-1 block: resident-weight persistent kernel (kernels_res32.h)
    size_t dev_wres = 0;          // offset (floats) of its weight image [tap][hi,lo][g][cout][8 halves]
    bool upc_ok = false;          // decoder c0 block whose "up" half is composed with the preceding ConvTranspose2d (kernels_upc.h)
    int up_idx = -1;              // ... index of that transposed conv in the program
    size_t dev_wc = 0;            // composed weights [chunk Cb/16][column tile][16 = (A,B,dI,dJ)][hi,lo][h][column][8 halves]
    size_t dev_wk = 0;            // skip half of the 3x3 weights [chunk Cs/16][column tile][tap][hi,lo][h][column][8 halves]
    size_t dev_wcs = 0;           // 1 / (their common power-of-two pre-scale)
    size_t dev_bvar = 0;          // [9 = (ry, rx)][Cout] bias variants
    bool up0_ok = false;          // ... with 64 coarse, 32 skip and 32 output channels (level 0 of the canonical net): kernels_up0.h
    size_t dev_w0c = 0;           // composed weights in fragment order [parity 4][k-step 8][hi,lo][cb 2][lane 64][8 halves]
    size_t dev_w0k = 0;           // skip half, resident image [tap 9][hi,lo][g 4][cout 32][8 halves]
    bool first_direct = false;    // first conv block handled by conv3x3_first (reads the NCHW boundary tensor)
    size_t dev_wraw = 0;
};

struct Launch { std::string name, kernel; hipEvent_t e0 = nullptr, e1 = nullptr; };

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: one process may drive engines on several GPUs
// (include/ts2d_engine.h: handles are independent), so the "already set" state is a bit per device, not one flag per process.
inline hipError_t allow_max_lds(const void* kern, std::atomic<uint64_t>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Pixel tiling of one level: tile = NIMG images x TH x TW pixels, NIMG * TH * TW <= 256 (a workgroup's 256 GEMM rows).
struct TileGeom { int TH, TW, NIMG, lgTH, lgTW, tiles_x, tiles_y, n_mtiles, PH, PW; };

enum Kern { K_NONE = 0, K_FUSED_AWAY, K_FIRST, K_FIRST_STATS, K_EXACT, K_S1_GENERIC, K_S1_ONE, K_S1_QP, K_S1_H32, K_S1_H2, K_S1_RES32, K_S1_RES32F,
            K_S2_V2, K_S2_ONE, K_S2_GENERIC, K_T_ONE, K_T_GENERIC, K_UP0, K_UPQ, K_UPC, K_UPC_H, K_UPC_H2, K_HEAD_MFMA, K_HEAD_1X1 };

struct Choice {
    Kern k = K_NONE;
    TileGeom g{};               // the pixel tiling the kernel walks
    int bn = 0;                 // output columns per workgroup
    int ksplit = 1;
    bool fused_stats = false;   // per-tile partial statistics come out of the kernel's epilogue (else stats_direct / splitk_reduce_stats)
    int ppt = 1;                // ... partials per tile and channel (conv3x3_up0: one per wave)
    bool first_full = false;    // (K_FIRST*) complete one-image 256-pixel tiles: the persistent variant
    bool flex = false;          // (K_UPC / K_UPC_H / K_S2_V2) the level is no multiple of 8 x 32: the FLEX instance on the tile c.g
};


inline int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

}  // namespace

struct ts2d_engine {
    ts2d_arch_desc arch{};
    int lvl_y[TS2D_MAX_STAGES] = {0}, lvl_x[TS2D_MAX_STAGES] = {0};      // per stage: log2 of the cumulative stride along H / W
    int device = 0;
    int cin_pad = 8;
    std::vector<Tensor> tensors;
    std::vector<Op> ops;
    size_t blob_floats = 0;
    ts2d_arch_desc user_arch{};        // the descriptor the caller gave; `arch` is what the kernels run (pad_arch)
    bool padded = false;
    size_t user_blob_floats = 0;       // floats of the caller's blob (== blob_floats unless padded)
    float* d_weights = nullptr; size_t weight_floats = 0;
    bool weights_ready = false;
    int precision = TS2D_PRECISION_F32_SPLIT_F16X3;
    int num_cus = 256;
    int tile_half = 0;            // sliding-window blend order (ts2d_engine_set_tile_dtype): 0 = fp32 tile (reference CPU path)
    int tiled_inf = 0;            // the last predict_tiled produced an infinite aggregated logit (ts2d_engine_tiled_inf_flag)
    // Kernel-dispatch options (ts2d_engine_set_option; ABI 6).  Each selects between two complete, parity-tested kernels for the ops it
    // names; the defaults are the measured-fastest ones.  They are per-HANDLE state set through the C-ABI - the caller's environment
    // never changes which kernel runs.
    bool use_h32 = true;          // "h32": precision mode f16, 32-channel-chunk kernel (0: the 16-channel one)
    bool use_one = true;          // "one": one-image-tile kernels (0: the generic ones; also switches every kernel built on them off)
    bool use_s2v2 = true;         // "s2v2": 512-thread stride-2 kernel
    bool use_q = true;            // "q": persistent 512-thread double-buffered stride-1 kernel on 16 x 32 tiles (0: conv3x3_f16x3_one)
    int upq_min = 256;            // "upq_min": least coarse channel count served by conv3x3_upq
    bool use_h2 = true;           // "h2": 16-bit plain C -> C blocks on 16 x 32 tiles (0: conv3x3_h32); "h2_min": least channel count
    int h2_min = 64;
    bool use_uh2 = true;          // "uh2": 16-bit composed block on 16 x 32 tiles (0: conv3x3_upc_h)
    bool use_s2k32 = true;        // "s2k32": 16-bit mode - the 512-thread stride-2 kernel on 32-channel chunks (0: 16-channel chunks)
    bool use_sbk = true;          // "sbk": small batches - split-K (and the two-kernel decoder entry) where the preferred kernel would leave most CUs idle (fill_ksplit)
    bool use_first_split = true;  // "first_split": the first block's contraction as one fp16 hi / lo split product (kernels_first.h SPLIT; 0: exact fp32 MFMA)
    bool use_up0 = true;          // "up0": dedicated persistent kernel of the level-0 composed block (0: conv3x3_upc<32>)
    int u0seg = 0;                // "u0seg": tiles per workgroup segment of conv3x3_up0 (0: chosen from the grid; tests force segments that end inside an image)
    bool use_upq = true;          // "upq": 512-thread double-buffered variant of the composed block on 16 x 32 tiles (0: conv3x3_upc)
    bool use_upc = true;          // "upc": decoder c0 blocks composed with their transposed conv (0: two kernels)
    bool use_res = true;          // "res": resident-weight kernel of the 32 -> 32 blocks
    bool use_fuse0 = true;        // "fuse0": first block recomputed inside the second (statistics-only pre-pass + conv3x3_res32 fused variant)
    int use_flex2 = 2;            // "flex2": conv3x3s2_v2 on level-dividing tiles where the level is no multiple of 8 x 32 (0: off, 1: 16-bit mode only, 2: both)
    bool use_flex = true;         // "flex": the composed decoder entry on extent-following tiles at levels that are no multiple of 8 x 32 (0: two kernels there)
    int dbg = 0;                  // TS2D_DBG (the one environment switch left): timing ablations / in-kernel phase stamps of diagnostic runs
    unsigned long long* d_prof = nullptr;     // TS2D_DBG=256: in-kernel phase counters, 8 per op (diagnostic)
    std::vector<char> fused_away; // per op of the last run: 1 = its output tensor was not materialised (composed into the next op)
    // workspace
    char* d_ws = nullptr; size_t ws_bytes = 0; int wsB = 0, wsH = 0, wsW = 0;
    bool ws_external = false;     // d_ws is the caller's memory (ts2d_engine_set_workspace): shared by the engines of a sub-model set
    uint32_t ws_token = 0;        // ... token this engine's last run wrote into its header (g_ws_generation)
    // the kernel choices of the last (B, H, W, precision, options): a forward of the same shape does not search tile shapes again
    // (62 launches per forward: at B = 1 the host side is a visible part of the 2.5 ms)
    std::vector<Choice> plan; int planB = 0, planH = 0, planW = 0, plan_prec = -1; unsigned plan_gen = 0, opt_gen = 1;
    char* d_stage = nullptr; size_t stage_bytes = 0;      // host-buffer forwards only (ensure_staging)
    int ws_precision = -1; bool ws_keep = false;      // the activation plan of the workspace was made for this mode (composition depends on it)
    bool keep_activations = false;                    // ts2d_engine_set_keep_activations: one buffer per tensor (debug access, full diagnosis)
    float* d_part = nullptr;
    float* d_partial = nullptr;   // split-K partial outputs
    float* d_up = nullptr;        // scratch for ONE upsampled tensor (small batches: a decoder entry that the reserved batch's plan composes runs as two kernels)
    float* d_in_stage = nullptr; float* d_logit_stage = nullptr; uint32_t* d_mask_stage = nullptr;
    hipStream_t stream = nullptr;
    bool profiling = false;
    std::vector<Launch> launches; size_t n_launched = 0;
    int lastB = 0, lastH = 0, lastW = 0; hipStream_t last_stream = nullptr; bool last_f16 = false;
    char* d_sw = nullptr; size_t sw_bytes = 0;     // sliding-window scratch (image, batch, tile logits, outputs)
    // The activation workspace is shared by every call on this handle, whatever stream the caller passes: the end of each run
    // is marked with an event, and a run issued on ANOTHER stream first waits for it (no host synchronisation).
    hipEvent_t ws_event = nullptr; hipStream_t ws_stream = nullptr; bool ws_busy = false;
    int* d_flags = nullptr;       // [0]: a logit of the last run was inf / NaN (set by the head kernels); [1]: scratch of the diagnosis
    bool checked_input = false; const float* last_input = nullptr;      // (synchronous paths: the staged input can be scanned too)
};

namespace {

int tensor_index(ts2d_engine* e, const std::string& name) {
    for (size_t i = 0; i < e->tensors.size(); ++i) if (e->tensors[i].name == name) return (int)i;
    return -1;
}

int add_tensor(ts2d_engine* e, const std::string& name, int C, int level, bool normed) {
    Tensor t; t.name = name; t.C = C; t.level = level; t.normed = normed; t.ly = e->lvl_y[level]; t.lx = e->lvl_x[level];
    t.Cu = (name == "input") ? C : e->user_arch.features[level];
    e->tensors.push_back(t);
    return (int)e->tensors.size() - 1;
}

// Feature counts the MFMA tilings do not divide ("a descriptor-driven engine needs a correct generic path for what it
// cannot make fast"): every stage's width is rounded up to a multiple of 32 (stage 0: to 32 or 64, what the head kernel reads) and the network
// runs at that width with ZERO weights, bias, gamma and beta in the added channels.  Exact, not approximate: an added channel's conv output is
// 0 in every pixel, its InstanceNorm scale gamma * rsqrt(0 + eps) = 0 and shift 0, so it enters the next layer as LeakyReLU(0) = 0 under zero
// weights; the caller's channels see the same sums with zeros added.  The caller's blob keeps the caller's layout (expand_blob below).
ts2d_arch_desc pad_arch(const ts2d_arch_desc& a, bool& padded) {
    ts2d_arch_desc p = a;
    padded = false;
    for (int s = 0; s < a.n_stages && s < TS2D_MAX_STAGES; ++s) {
        if (a.features[s] < 1) continue;                       // (build_program reports it)
        p.features[s] = (a.features[s] + 31) / 32 * 32;
        padded |= p.features[s] != a.features[s];
    }
    return p;
}

// The parameter tensors of a descriptor in state-dict order, as (rows, columns of the first / second source, inner) blocks; vectors have inner 0.
struct ParamSeg { int rows, ca, cb, inner; };
std::vector<ParamSeg> param_segs(const ts2d_arch_desc& a) {
    std::vector<ParamSeg> v;
    auto stride_of = [&](int s, int ax) { const int y = a.strides[s][0], x = a.strides[s][1]; return (y == 0 && x == 0) ? (s ? 2 : 1) : (ax ? x : y); };
    int cin = a.input_channels;
    for (int s = 0; s < a.n_stages; ++s)
        for (int i = 0; i < a.n_conv_enc[s]; ++i) {
            const int f = a.features[s];
            v.push_back({f, cin, 0, 9}); v.push_back({f, 0, 0, 0}); v.push_back({f, 0, 0, 0}); v.push_back({f, 0, 0, 0});
            cin = f;
        }
    for (int j = 0; j < a.n_stages - 1; ++j) {
        const int lvl = a.n_stages - 2 - j, f = a.features[lvl];
        v.push_back({cin, f, 0, stride_of(lvl + 1, 0) * stride_of(lvl + 1, 1)}); v.push_back({f, 0, 0, 0});      // ConvTranspose2d weight [Cin][Cout][sy][sx], bias
        for (int i = 0; i < a.n_conv_dec[j]; ++i) {
            v.push_back({f, f, i == 0 ? f : 0, 9}); v.push_back({f, 0, 0, 0}); v.push_back({f, 0, 0, 0}); v.push_back({f, 0, 0, 0});
        }
        cin = f;
    }
    v.push_back({a.num_classes, cin, 0, 1}); v.push_back({a.num_classes, 0, 0, 0});
    return v;
}

size_t segs_floats(const std::vector<ParamSeg>& v) {
    size_t n = 0;
    for (const ParamSeg& g : v) n += g.inner ? (size_t)g.rows * (g.ca + g.cb) * g.inner : (size_t)g.rows;
    return n;
}

// caller-layout blob -> the blob of the padded architecture (zeros in every added row / column; the second source of a decoder block's
// first conv - the skip half of cat((up, skip), 1) - starts at the PADDED width of the first)
void expand_blob(const ts2d_arch_desc& ua, const ts2d_arch_desc& pa, const float* ub, std::vector<float>& pb) {
    const std::vector<ParamSeg> us = param_segs(ua), ps = param_segs(pa);
    pb.assign(segs_floats(ps), 0.f);
    size_t uo = 0, po = 0;
    for (size_t k = 0; k < us.size(); ++k) {
        const ParamSeg &u = us[k], &q = ps[k];
        if (!u.inner) {
            memcpy(pb.data() + po, ub + uo, (size_t)u.rows * sizeof(float));
            uo += u.rows; po += q.rows;
            continue;
        }
        const int ucols = u.ca + u.cb, qcols = q.ca + q.cb;
        for (int r = 0; r < u.rows; ++r)
            for (int c = 0; c < ucols; ++c) {
                const int cq = c < u.ca ? c : q.ca + (c - u.ca);
                memcpy(pb.data() + po + ((size_t)r * qcols + cq) * q.inner, ub + uo + ((size_t)r * ucols + c) * u.inner, (size_t)u.inner * sizeof(float));
            }
        uo += (size_t)u.rows * ucols * u.inner; po += (size_t)q.rows * qcols * q.inner;
    }
}

// Mirror of UNetArch.program() (totalsegmentator2d_amd/arch.py): PlainConvUNet forward order.
int build_program(ts2d_engine* e) {
    const ts2d_arch_desc& a = e->arch;
    if (a.n_stages < 2 || a.n_stages > TS2D_MAX_STAGES) return fail(TS2D_ERR_INVALID, "n_stages %d out of range [2,%d]", a.n_stages, TS2D_MAX_STAGES);
    if (a.input_channels < 1 || a.num_classes < 1) return fail(TS2D_ERR_INVALID, "input_channels / num_classes must be positive");
    if (a.num_classes > 256) return fail(TS2D_ERR_INVALID, "num_classes %d > 256 is not supported", a.num_classes);
    for (int s = 0; s < a.n_stages; ++s) {
        if (e->user_arch.features[s] < 1) return fail(TS2D_ERR_INVALID, "features[%d] = %d must be positive", s, e->user_arch.features[s]);
        if (a.n_conv_enc[s] < 1) return fail(TS2D_ERR_INVALID, "n_conv_enc[%d] must be >= 1", s);
        if (s < a.n_stages - 1 && a.n_conv_dec[s] < 1) return fail(TS2D_ERR_INVALID, "n_conv_dec[%d] must be >= 1", s);
    }
    if (a.features[0] != 32 && a.features[0] != 64) return fail(TS2D_ERR_INVALID, "features[0] = %d: the head kernel supports at most 64 channels", e->user_arch.features[0]);
    // per-axis strides (ABI 7).  An all-zero entry = (2, 2): descriptors written for ABI <= 6 carry no strides.
    int st[TS2D_MAX_STAGES][2];
    for (int s = 0; s < a.n_stages; ++s) {
        st[s][0] = a.strides[s][0]; st[s][1] = a.strides[s][1];
        if (st[s][0] == 0 && st[s][1] == 0) { st[s][0] = st[s][1] = (s == 0) ? 1 : 2; }
        if (st[s][0] < 1 || st[s][0] > 2 || st[s][1] < 1 || st[s][1] > 2)
            return fail(TS2D_ERR_INVALID, "strides[%d] = (%d, %d): 1 or 2 per axis", s, st[s][0], st[s][1]);
        if (s == 0 && (st[0][0] != 1 || st[0][1] != 1)) return fail(TS2D_ERR_INVALID, "strides[0] must be (1, 1)");
        e->lvl_y[s] = (s ? e->lvl_y[s - 1] : 0) + (st[s][0] == 2); e->lvl_x[s] = (s ? e->lvl_x[s - 1] : 0) + (st[s][1] == 2);
    }
    auto stride_code = [](int sy, int sx) { return sy == 1 && sx == 1 ? 1 : (sy == 2 && sx == 2 ? 2 : 3); };
    e->cin_pad = (a.input_channels + 7) / 8 * 8;
    size_t bo = 0;
    int cur = add_tensor(e, "input", e->cin_pad, 0, false), cin = a.input_channels;
    std::vector<int> skips(a.n_stages);
    char nm[64];
    for (int s = 0; s < a.n_stages; ++s) {
        const int f = a.features[s];
        for (int i = 0; i < a.n_conv_enc[s]; ++i) {
            snprintf(nm, sizeof(nm), "enc%d.c%d", s, i);
            Op op{}; op.type = OP_CONV; op.name = nm; op.src = cur; op.skip = -1;
            op.cin = cin; op.cin_skip = 0; op.cout = f; op.level = s; op.ly = e->lvl_y[s]; op.lx = e->lvl_x[s];
            if (i == 0 && s > 0) { op.sy = st[s][0]; op.sx = st[s][1]; }
            op.stride = stride_code(op.sy, op.sx);
            op.dst = add_tensor(e, nm, f, s, true);
            op.blob_w = bo; bo += (size_t)f * cin * 9; op.blob_b = bo; bo += f; op.blob_g = bo; bo += f; op.blob_be = bo; bo += f;
            e->ops.push_back(op);
            cur = op.dst; cin = f;
        }
        skips[s] = cur;
    }
    for (int j = 0; j < a.n_stages - 1; ++j) {
        const int lvl = a.n_stages - 2 - j, f = a.features[lvl];
        snprintf(nm, sizeof(nm), "dec%d.up", lvl);
        Op up{}; up.type = OP_CONVT; up.name = nm; up.src = cur; up.skip = -1; up.cin = cin; up.cin_skip = 0; up.cout = f;
        up.sy = st[lvl + 1][0]; up.sx = st[lvl + 1][1];       // kernel = stride = the stride of the encoder stage below (upstream UNetDecoder)
        up.stride = stride_code(up.sy, up.sx); up.level = lvl; up.ly = e->lvl_y[lvl]; up.lx = e->lvl_x[lvl];
        up.dst = add_tensor(e, nm, f, lvl, false);
        up.blob_w = bo; bo += (size_t)cin * f * up.sy * up.sx; up.blob_b = bo; bo += f;
        e->ops.push_back(up);
        cur = up.dst;
        for (int i = 0; i < a.n_conv_dec[j]; ++i) {
            snprintf(nm, sizeof(nm), "dec%d.c%d", lvl, i);
            Op op{}; op.type = OP_CONV; op.name = nm; op.src = cur; op.skip = (i == 0) ? skips[lvl] : -1;
            op.cin = f; op.cin_skip = (i == 0) ? f : 0; op.cout = f; op.stride = 1; op.level = lvl; op.ly = e->lvl_y[lvl]; op.lx = e->lvl_x[lvl];
            op.dst = add_tensor(e, nm, f, lvl, true);
            const int ct = op.cin + op.cin_skip;
            op.blob_w = bo; bo += (size_t)f * ct * 9; op.blob_b = bo; bo += f; op.blob_g = bo; bo += f; op.blob_be = bo; bo += f;
            e->ops.push_back(op);
            cur = op.dst;
        }
        cin = f;
    }
    {
        Op hd{}; hd.type = OP_HEAD; hd.name = "head"; hd.src = cur; hd.skip = -1; hd.cin = cin; hd.cin_skip = 0;
        hd.cout = a.num_classes; hd.stride = 1; hd.level = 0; hd.dst = -1;
        hd.blob_w = bo; bo += (size_t)a.num_classes * cin; hd.blob_b = bo; bo += a.num_classes;
        e->ops.push_back(hd);
    }
    e->blob_floats = bo;
    e->user_blob_floats = e->padded ? segs_floats(param_segs(e->user_arch)) : bo;
    // device weight arena layout
    size_t wo = 0;
    for (Op& op : e->ops) {
        const int ct = op.cin + op.cin_skip;
        if (op.type == OP_CONV) {
            const int ctp = (op.src == 0) ? e->cin_pad : ct;             // first conv reads the zero-padded input
            op.ck = (op.stride != 1 || ctp % 16) ? 8 : 16;
            op.dev_w_floats = (size_t)ctp * 9 * op.cout;
        } else if (op.type == OP_CONVT) {
            op.ck = 16;
            op.dev_w_floats = (size_t)ct * op.sy * op.sx * op.cout;
        } else {
            op.ck = 0;
            op.dev_w_floats = (size_t)ct * op.cout;
        }
        op.dev_w = wo; wo = align_up(wo + op.dev_w_floats, 64);
        op.dev_b = wo; wo = align_up(wo + op.cout, 64);
        if (op.type == OP_CONV) { op.dev_g = wo; wo = align_up(wo + op.cout, 64); op.dev_be = wo; wo = align_up(wo + op.cout, 64); }
        if (op.type == OP_CONV && op.src != 0 && ((op.stride == 1 && ct % 16 == 0) || (op.stride == 2 && ct % 8 == 0))) {
            op.split_ok = true;        // stride 1: [chunk16][tap 9][Cout][32 halves]; stride 2: [chunk8][column tile][k-step 5][column][32 halves]
            const size_t recs = op.stride == 1 ? (size_t)(ct / 16) * 9 : (size_t)(ct / 8) * 5;
            op.dev_wh = wo; wo = align_up(wo + recs * op.cout * 16, 64);
            op.dev_ws = wo; wo = align_up(wo + 1, 64);                  // 1 / scale, read by the kernel
            if (op.stride == 1 && op.cout % 64 == 0) { op.dev_wp = wo; wo = align_up(wo + recs * op.cout * 16, 64); }      // plane order (conv3x3_f16x3_qp)
            if (op.stride == 2 && ct % 16 == 0 && op.cout % 64 == 0) {
                op.s2v2_ok = true; op.bn2 = op.cout % 128 == 0 ? 128 : 64;
                op.dev_w2 = wo; wo = align_up(wo + (size_t)ct * 9 * op.cout, 64);
            }
            if (op.stride == 1 && ct == 32 && op.cout == 32) {      // resident image of the 32 -> 32 block: [tap][hi,lo][g][cout][8 halves]
                op.res_ok = true;
                op.dev_wres = wo; wo = align_up(wo + 9 * 2 * 4 * 32 * 8 / 2, 64);
            }
            const size_t oi = (size_t)(&op - e->ops.data());
            if (op.stride == 1 && op.skip >= 0 && oi > 0 && e->ops[oi - 1].type == OP_CONVT && e->ops[oi - 1].dst == op.src &&
                e->ops[oi - 1].stride == 2 && e->ops[oi - 1].cin % 32 == 0 && op.cin % 16 == 0 && op.cin_skip % 16 == 0 && op.cout % 32 == 0 &&
                (double)op.cout * e->ops[oi - 1].cin * op.cin * 36.0 <= 6.0e9) {        // (host composition cost bound: 512 x 512 x 512 channels = 4.8 GFLOP, about a second)
                const int cb = e->ops[oi - 1].cin;
                op.upc_ok = true; op.up_idx = (int)oi - 1;
                op.dev_wc = wo; wo = align_up(wo + (size_t)cb * op.cout * 16, 64);
                op.dev_wk = wo; wo = align_up(wo + (size_t)op.cin_skip * op.cout * 9, 64);
                op.dev_wcs = wo; wo = align_up(wo + 1, 64);
                op.dev_bvar = wo; wo = align_up(wo + (size_t)9 * op.cout, 64);
                if (cb == 64 && op.cin == 32 && op.cin_skip == 32 && op.cout == 32) {
                    op.up0_ok = true;
                    op.dev_w0c = wo; wo = align_up(wo + 4 * 8 * 2 * 2 * 64 * 8 / 2, 64);
                    op.dev_w0k = wo; wo = align_up(wo + 9 * 2 * 4 * 32 * 8 / 2, 64);
                }
            }
            if (op.stride == 1 && op.cin % 32 == 0 && op.cin_skip % 32 == 0) {      // [chunk32][column tile][tap][column][32 halves]
                op.h32_ok = true;
                op.dev_wh32 = wo; wo = align_up(wo + (size_t)(ct / 32) * 9 * op.cout * 16, 64);
            }
        }
        if (op.type == OP_CONV && op.src == 0 && ct <= 4) {              // first block: PyTorch-layout fp32 copy for conv3x3_first
            op.dev_wraw = wo; wo = align_up(wo + (size_t)op.cout * ct * 9, 64);
            op.first_direct = true;
        }
        if (op.type == OP_HEAD && ct == 32 && op.cout <= 32) {          // head_mfma32: [k-step 2][column 32][16 hi | 16 lo] halves
            op.split_ok = true;
            op.dev_wh = wo; wo = align_up(wo + 2 * 32 * 16, 64);
            op.dev_ws = wo; wo = align_up(wo + 1, 64);
        }
        if (op.type == OP_CONVT && op.stride == 2 && ct % 32 == 0) {    // [chunk32][column tile of 64][k-step 2][column][32 halves]
            op.split_ok = true;
            op.dev_wh = wo; wo = align_up(wo + (size_t)(ct / 32) * 2 * 4 * op.cout * 16, 64);
            op.dev_ws = wo; wo = align_up(wo + 1, 64);
        }
    }
    e->weight_floats = wo;
    return TS2D_OK;
}

// fp32 -> fp16 bits, round to nearest even (host side of the weight split; the device uses v_cvt_f16_f32, also RNE)
uint16_t f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                       // rounds to >= 65520 -> inf
    if (x < 0x38800000u) {                                                          // subnormal half (or zero)
        if (x < 0x33000000u) return (uint16_t)sign;
        const int shift = 126 - (int)(x >> 23);                                     // 14..24
        uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
        const uint32_t half = m >> shift, rem = m & ((1u << shift) - 1), mid = 1u << (shift - 1);
        return (uint16_t)(sign | (half + ((rem > mid || (rem == mid && (half & 1))) ? 1 : 0)));
    }
    const uint32_t rem = x & 0x1FFFu;
    uint32_t hbits = (x - 0x38000000u) >> 13;
    if (rem > 0x1000u || (rem == 0x1000u && (hbits & 1))) ++hbits;
    return (uint16_t)(sign | hbits);
}
float f16_to_f32(uint16_t hb) {
    const uint32_t sign = (uint32_t)(hb & 0x8000u) << 16, ex = (hb >> 10) & 0x1F, m = hb & 0x3FFu;
    uint32_t x;
    if (ex == 0) {
        if (m == 0) x = sign;
        else { float v = (float)m * 5.9604644775390625e-08f; memcpy(&x, &v, 4); x |= sign; }
    } else if (ex == 31) x = sign | 0x7F800000u | (m << 13);
    else x = sign | ((ex + 112) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4); return f;
}

// Host threads for the weight packing (the fp64 composition of the transposed convs is 4.8 GFLOP per 512-channel level: 3.5 s
// single-threaded for the canonical net - VERDICT r2 item 14).  Static partition of [0, n): every index is written by one thread.
template <typename F>
void parallel_for(int n, F&& f) {
    int nt = (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(std::min(nt, 16), n));
    if (nt == 1) { for (int i = 0; i < n; ++i) f(i); return; }
    // Thread creation can fail (process / thread limits of a pool box): nothing may escape through the C-ABI, and a started thread
    // must be joined.  Thread t takes the indices t, t + nt, ...; the lanes whose thread could not start run here, serially.
    std::vector<std::thread> th;
    int started = 0;
    try {
        th.reserve(nt);
        for (int t = 0; t < nt; ++t) { th.emplace_back([&f, t, nt, n]() { for (int i = t; i < n; i += nt) f(i); }); ++started; }
    } catch (...) { /* fall through with `started` threads */ }
    for (int t = started; t < nt; ++t) for (int i = t; i < n; i += nt) f(i);
    for (auto& x : th) if (x.joinable()) x.join();
}

// PyTorch-layout blob -> packed device layouts (host staging buffer `out`, weight_floats long).
void pack_weights(const ts2d_engine* e, const float* blob, float* out) {
    memset(out, 0, e->weight_floats * sizeof(float));
    for (const Op& op : e->ops) {
        if (op.first_direct) memcpy(out + op.dev_wraw, blob + op.blob_w, (size_t)op.cout * op.cin * 9 * sizeof(float));
        if (op.type == OP_HEAD && op.split_ok) {
            const int ct = op.cin, co_n = op.cout;          // W[k][c]
            const float* w = blob + op.blob_w;
            float mx = 0.f;
            for (size_t i = 0; i < (size_t)ct * co_n; ++i) mx = std::max(mx, std::fabs(w[i]));
            const float wscale = (mx > 0.f && std::isfinite(mx)) ? std::exp2(std::floor(std::log2(16383.0f / mx))) : 1.f;
            out[op.dev_ws] = 1.0f / wscale;
            uint16_t* d = reinterpret_cast<uint16_t*>(out + op.dev_wh);         // zero-initialised: columns >= K stay 0
            for (int k = 0; k < co_n; ++k)
                for (int c = 0; c < ct; ++c) {
                    const float v = w[(size_t)k * ct + c] * wscale;
                    const uint16_t hi = f32_to_f16(v), lo = f32_to_f16(v - f16_to_f32(hi));
                    uint16_t* rec = d + ((size_t)(c / 16) * 32 + k) * 32;
                    rec[c % 16] = hi; rec[16 + c % 16] = lo;
                }
        }
        if (op.type == OP_CONVT && op.split_ok) {
            const int ct = op.cin, co_n = op.cout, N = 4 * co_n;
            const float* w = blob + op.blob_w;
            float mx = 0.f;
            for (size_t i = 0; i < (size_t)ct * co_n * 4; ++i) mx = std::max(mx, std::fabs(w[i]));
            const float wscale = (mx > 0.f && std::isfinite(mx)) ? std::exp2(std::floor(std::log2(16383.0f / mx))) : 1.f;
            out[op.dev_ws] = 1.0f / wscale;
            uint16_t* d = reinterpret_cast<uint16_t*>(out + op.dev_wh);
            for (int ci = 0; ci < ct; ++ci) {
                const int chunk = ci / 32, kk = (ci % 32) / 16, cc = ci % 16;
                for (int co = 0; co < co_n; ++co)
                    for (int ab = 0; ab < 4; ++ab) {
                        const float v = w[((size_t)ci * co_n + co) * 4 + ab] * wscale;
                        const uint16_t hi = f32_to_f16(v);
                        const uint16_t lo = f32_to_f16(v - f16_to_f32(hi));
                        const int n = ab * co_n + co;          // [chunk32][column tile of 64][k-step][column][32 halves]
                        uint16_t* rec = d + ((((size_t)chunk * (N / 64) + n / 64) * 2 + kk) * 64 + n % 64) * 32;
                        rec[cc] = hi; rec[16 + cc] = lo;
                    }
            }
        }
        if (op.type == OP_CONV && op.split_ok) {
            // split-fp16 image: w * S = hi + lo with S = 2^k chosen so that max|w| * S is in [8192, 16384): hi and lo of
            // typical weights stay in fp16's normal range, products stay far from fp32 overflow.
            const int ct = op.cin + op.cin_skip, co_n = op.cout;
            const float* w = blob + op.blob_w;
            float mx = 0.f;
            for (size_t i = 0; i < (size_t)co_n * ct * 9; ++i) mx = std::max(mx, std::fabs(w[i]));
            const float wscale = (mx > 0.f && std::isfinite(mx)) ? std::exp2(std::floor(std::log2(16383.0f / mx))) : 1.f;
            out[op.dev_ws] = 1.0f / wscale;
            uint16_t* d = reinterpret_cast<uint16_t*>(out + op.dev_wh);
            parallel_for(co_n, [&](int co) {                         // (every packed element belongs to one output channel)
                for (int ci = 0; ci < ct; ++ci) {
                    for (int tap = 0; tap < 9; ++tap) {
                        const float v = w[((size_t)co * ct + ci) * 9 + tap] * wscale;
                        const uint16_t hi = f32_to_f16(v);
                        const uint16_t lo = f32_to_f16(v - f16_to_f32(hi));
                        if (op.stride == 1) {   // [chunk16][column tile][tap][column in tile][16 hi | 16 lo]: one contiguous block per (chunk, tile)
                            const int chunk = ci / 16, cc = ci % 16, bn = co_n % 64 == 0 ? 64 : 32;
                            uint16_t* rec = d + ((((size_t)chunk * (co_n / bn) + co / bn) * 9 + tap) * bn + co % bn) * 32;
                            rec[cc] = hi; rec[16 + cc] = lo;
                            if (bn == 64) {   // plane order of conv3x3_f16x3_qp: [chunk][column tile][tap][hi,lo][h][column][8 halves]
                                uint16_t* wp = reinterpret_cast<uint16_t*>(out + op.dev_wp);
                                const size_t base = (((size_t)chunk * (co_n / bn) + co / bn) * 9 + tap) * 4;
                                wp[((base + 0 + cc / 8) * bn + co % bn) * 8 + cc % 8] = hi;
                                wp[((base + 2 + cc / 8) * bn + co % bn) * 8 + cc % 8] = lo;
                            }
                            if (op.res_ok) {
                                uint16_t* rw = reinterpret_cast<uint16_t*>(out + op.dev_wres);
                                rw[((((size_t)tap * 2 + 0) * 4 + ci / 8) * 32 + co) * 8 + ci % 8] = hi;
                                rw[((((size_t)tap * 2 + 1) * 4 + ci / 8) * 32 + co) * 8 + ci % 8] = lo;
                            }
                            if (op.h32_ok)          // same blocks with 32 real channels per record (the hi parts only)
                                reinterpret_cast<uint16_t*>(out + op.dev_wh32)[((((size_t)(ci / 32) * (co_n / bn) + co / bn) * 9 + tap) * bn + co % bn) * 32 + ci % 32] = hi;
                        } else {       // K packed as 8 channels x 2 taps: k = 8 * (tap & 1) + (ci % 8) of k-step tap / 2
                            if (op.s2v2_ok) {      // [chunk16][column tile][tap][hi,lo][h = (ci % 16) / 8][column][ci % 8]
                                const int bn2 = op.bn2;
                                uint16_t* w2 = reinterpret_cast<uint16_t*>(out + op.dev_w2);
                                const size_t blk = ((size_t)(ci / 16) * (co_n / bn2) + co / bn2) * (9 * 2 * 2 * bn2 * 8);
                                w2[blk + ((((size_t)tap * 2 + 0) * 2 + (ci % 16) / 8) * bn2 + co % bn2) * 8 + ci % 8] = hi;
                                w2[blk + ((((size_t)tap * 2 + 1) * 2 + (ci % 16) / 8) * bn2 + co % bn2) * 8 + ci % 8] = lo;
                            }
                            const int chunk = ci / 8, cc = (tap & 1) * 8 + ci % 8, bn = co_n % 64 == 0 ? 64 : 32;
                            uint16_t* rec = d + ((((size_t)chunk * (co_n / bn) + co / bn) * 5 + tap / 2) * bn + co % bn) * 32;
                            rec[cc] = hi; rec[16 + cc] = lo;
                        }
                    }
                }
            });
        }
    }
    for (const Op& op : e->ops) {
        if (!op.upc_ok) continue;
        // ConvTranspose2d (2x2, stride 2) composed with the "up" half of the following 3x3 conv (kernels_upc.h), in fp64:
        //   Weff[A][B][dI][dJ][co][cb] = sum_{(ky,kx) -> (dI,dJ)} sum_cu W3[co][cu][ky][kx] WT[cb][cu][a][b]
        // with, for output parity A and tap ky: up row 2I + A + ky - 1 = 2 (I + A - 1 + dI) + a.
        const Op& up = e->ops[op.up_idx];
        const int cb_n = up.cin, cu_n = op.cin, cs_n = op.cin_skip, ct = cu_n + cs_n, co_n = op.cout;
        const float* w3 = blob + op.blob_w;          // [co][ct][3][3], channels 0 .. cu_n-1 = the upsampled half (torch.cat((up, skip), 1))
        const float* wt = blob + up.blob_w;          // [cb][cu][2][2]
        const float* bt = blob + up.blob_b;
        std::vector<double> R((size_t)4 * cu_n * cb_n);              // R[ab][cu][cb]
        for (int cb = 0; cb < cb_n; ++cb)
            for (int cu = 0; cu < cu_n; ++cu)
                for (int ab = 0; ab < 4; ++ab) R[((size_t)ab * cu_n + cu) * cb_n + cb] = wt[((size_t)cb * cu_n + cu) * 4 + ab];
        std::vector<double> Weff((size_t)16 * co_n * cb_n, 0.0);     // [tapidx = (A*2+B)*4 + dI*2+dJ][co][cb]
        parallel_for(co_n, [&](int co) {                             // (rows of Weff are disjoint per output channel; same summation order as serial)
            for (int A = 0; A < 2; ++A)
                for (int Bp = 0; Bp < 2; ++Bp)
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const int fy = (A + ky + 1) / 2 - 1, fx = (Bp + kx + 1) / 2 - 1;          // floor((A + ky - 1) / 2)
                            const int dI = fy - A + 1, dJ = fx - Bp + 1, ta = (A + ky + 1) & 1, tb = (Bp + kx + 1) & 1;
                            double* arow = Weff.data() + ((size_t)((A * 2 + Bp) * 4 + dI * 2 + dJ) * co_n + co) * cb_n;
                            const double* Rm = R.data() + (size_t)(ta * 2 + tb) * cu_n * cb_n;
                            for (int cu = 0; cu < cu_n; ++cu) {
                                const double l = w3[((size_t)co * ct + cu) * 9 + ky * 3 + kx];
                                const double* rrow = Rm + (size_t)cu * cb_n;
                                for (int cb = 0; cb < cb_n; ++cb) arow[cb] += l * rrow[cb];
                            }
                        }
        });
        double mx = 0.0;
        for (double v : Weff) mx = std::max(mx, std::fabs(v));
        for (int co = 0; co < co_n; ++co)
            for (int cs = 0; cs < cs_n; ++cs)
                for (int tap = 0; tap < 9; ++tap) mx = std::max(mx, (double)std::fabs(w3[((size_t)co * ct + cu_n + cs) * 9 + tap]));
        const float wscale = (mx > 0.0 && std::isfinite(mx)) ? std::exp2(std::floor(std::log2(16383.0 / mx))) : 1.f;
        out[op.dev_wcs] = 1.0f / wscale;
        const int bn = co_n % 64 == 0 ? 64 : 32, nct = co_n / bn;
        uint16_t* dc = reinterpret_cast<uint16_t*>(out + op.dev_wc);
        uint16_t* dk = reinterpret_cast<uint16_t*>(out + op.dev_wk);
        parallel_for(co_n, [&](int co) {
            for (int t = 0; t < 16; ++t)
                for (int cb = 0; cb < cb_n; ++cb) {
                    const float v = (float)(Weff[((size_t)t * co_n + co) * cb_n + cb] * (double)wscale);
                    const uint16_t hi = f32_to_f16(v), lo = f32_to_f16(v - f16_to_f32(hi));
                    const size_t base = (((size_t)(cb / 16) * nct + co / bn) * 16 + t) * 4;
                    const int hh = (cb % 16) / 8;
                    dc[((base + 0 + hh) * bn + co % bn) * 8 + cb % 8] = hi;
                    dc[((base + 2 + hh) * bn + co % bn) * 8 + cb % 8] = lo;
                    if (op.up0_ok) {       // fragment order of conv3x3_up0: lane = k-group (cb % 32) / 8, row co % 16
                        uint16_t* d0 = reinterpret_cast<uint16_t*>(out + op.dev_w0c);
                        const size_t f = ((size_t)(t >> 2) * 8 + (cb / 32) * 4 + (t & 3)) * 4;      // (parity, k-step) x [hi,lo][cb 2]
                        // MFMA block (co % 8) / 4, row 4 (co / 8) + co % 4: the lane that holds rows 4 g .. 4 g + 3 of both blocks owns channels 8 g .. 8 g + 7
                        const int blk = (co % 8) / 4, row = 4 * (co / 8) + co % 4;
                        const int ln = ((cb % 32) / 8) * 16 + row;
                        d0[((f + 0 + blk) * 64 + ln) * 8 + cb % 8] = hi;
                        d0[((f + 2 + blk) * 64 + ln) * 8 + cb % 8] = lo;
                    }
                }
            for (int cs = 0; cs < cs_n; ++cs)
                for (int tap = 0; tap < 9; ++tap) {
                    const float v = w3[((size_t)co * ct + cu_n + cs) * 9 + tap] * wscale;
                    const uint16_t hi = f32_to_f16(v), lo = f32_to_f16(v - f16_to_f32(hi));
                    const size_t base = (((size_t)(cs / 16) * nct + co / bn) * 9 + tap) * 4;
                    const int hh = (cs % 16) / 8;
                    dk[((base + 0 + hh) * bn + co % bn) * 8 + cs % 8] = hi;
                    dk[((base + 2 + hh) * bn + co % bn) * 8 + cs % 8] = lo;
                    if (op.up0_ok) {
                        uint16_t* k0 = reinterpret_cast<uint16_t*>(out + op.dev_w0k);
                        const int prow = ((co % 8) / 4) * 16 + 4 * (co / 8) + co % 4;      // (the same row permutation)
                        k0[((((size_t)tap * 2 + 0) * 4 + cs / 8) * 32 + prow) * 8 + cs % 8] = hi;
                        k0[((((size_t)tap * 2 + 1) * 4 + cs / 8) * 32 + prow) * 8 + cs % 8] = lo;
                    }
                }
        });
        // bias variants: the transposed conv's bias reaches an output pixel through the taps that lie inside the image
        float* bv = out + op.dev_bvar;
        for (int co = 0; co < co_n; ++co) {
            double bc[3][3];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    double acc = 0.0;
                    for (int cu = 0; cu < cu_n; ++cu) acc += (double)w3[((size_t)co * ct + cu) * 9 + ky * 3 + kx] * (double)bt[cu];
                    bc[ky][kx] = acc;
                }
            for (int ry = 0; ry < 3; ++ry)
                for (int rx = 0; rx < 3; ++rx) {
                    double acc = (double)blob[op.blob_b + co];
                    for (int ky = (ry == 0 ? 1 : 0); ky < (ry == 2 ? 2 : 3); ++ky)
                        for (int kx = (rx == 0 ? 1 : 0); kx < (rx == 2 ? 2 : 3); ++kx) acc += bc[ky][kx];
                    bv[(size_t)(ry * 3 + rx) * co_n + co] = (float)acc;
                }
        }
    }
    for (const Op& op : e->ops) {
        const int ct = op.cin + op.cin_skip, co_n = op.cout;
        if (op.type == OP_CONV) {           // W[co][ci][ky][kx] -> [chunk][tap][kk][co][8]
            const int ck = op.ck, kkn = ck / 8;
            const float* w = blob + op.blob_w;
            float* d = out + op.dev_w;
            parallel_for(co_n, [&](int co) {
                for (int ci = 0; ci < ct; ++ci) {
                    const int chunk = ci / ck, cc = ci % ck, kk = cc / 8, el = cc % 8;
                    for (int tap = 0; tap < 9; ++tap)
                        d[((((size_t)chunk * 9 + tap) * kkn + kk) * co_n + co) * 8 + el] = w[((size_t)co * ct + ci) * 9 + tap];
                }
            });
            memcpy(out + op.dev_g, blob + op.blob_g, co_n * sizeof(float));
            memcpy(out + op.dev_be, blob + op.blob_be, co_n * sizeof(float));
        } else if (op.type == OP_CONVT) {   // W[ci][co][a][b] -> [chunk][kk][(a*KB+b)*Cout + co][8]   (kernel = stride = (KA, KB))
            const int ck = op.ck, kkn = ck / 8, nab = op.sy * op.sx, N = nab * co_n;
            const float* w = blob + op.blob_w;
            float* d = out + op.dev_w;
            for (int ci = 0; ci < ct; ++ci) {
                const int chunk = ci / ck, cc = ci % ck, kk = cc / 8, el = cc % 8;
                for (int co = 0; co < co_n; ++co)
                    for (int ab = 0; ab < nab; ++ab)
                        d[(((size_t)chunk * kkn + kk) * N + ab * co_n + co) * 8 + el] = w[((size_t)ci * co_n + co) * nab + ab];
            }
        } else {                            // head W[k][c] as is
            memcpy(out + op.dev_w, blob + op.blob_w, (size_t)ct * co_n * sizeof(float));
        }
        memcpy(out + op.dev_b, blob + op.blob_b, co_n * sizeof(float));
    }
}

int upload_weights(ts2d_engine* e, const float* blob, size_t n_floats) {
    // (pack_weights also records the per-layer split scale in the op table)
    if (n_floats != e->user_blob_floats)
        return fail(TS2D_ERR_INVALID, "weight blob has %zu floats, architecture needs %zu", n_floats, e->user_blob_floats);
    std::vector<float> staging, wide;
    try {
        staging.resize(e->weight_floats);
        if (e->padded) { expand_blob(e->user_arch, e->arch, blob, wide); blob = wide.data(); }
    } catch (...) { return fail(TS2D_ERR_NOMEM, "host staging allocation failed"); }
    pack_weights(e, blob, staging.data());
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpy(e->d_weights, staging.data(), e->weight_floats * sizeof(float), hipMemcpyHostToDevice));
    e->weights_ready = true;
    return TS2D_OK;
}

inline int ct_total(const Op& op) { return op.cin + op.cin_skip; }
inline int lg_exact(int v) { return (v > 0 && (v & (v - 1)) == 0) ? ilog2(v) : -1; }


inline void tile_finish(TileGeom& g, int B, int Ht, int Wt, int sy, int sx, int taps) {
    const bool p2 = lg_exact(g.TH) >= 0 && lg_exact(g.TW) >= 0;
    g.lgTH = p2 ? ilog2(g.TH) : -1; g.lgTW = p2 ? ilog2(g.TW) : -1;
    if (g.NIMG > 1) { g.tiles_x = 1; g.tiles_y = 1; }
    else { g.tiles_x = (Wt + g.TW - 1) / g.TW; g.tiles_y = (Ht + g.TH - 1) / g.TH; }
    g.n_mtiles = (B + g.NIMG - 1) / g.NIMG * g.tiles_x * g.tiles_y;
    const int halo = (taps == 9) ? 3 : 1;
    g.PH = (g.TH - 1) * sy + halo; g.PW = (g.TW - 1) * sx + halo;
}

// Rounds 1-4: power-of-two tiles (TW = min(32, pow2ceil(Wt)) ...) - complete on 512^2 / 1024^2, idle rows everywhere else.  Kept for
// the first block (level 0 of a real plan is a multiple of 64 or more along both axes) and as the first candidate below.
TileGeom tile_geom_pow2(int B, int Ht, int Wt, int sy, int sx, int taps) {
    TileGeom g{};
    g.TW = std::min(32, pow2ceil(Wt));
    g.TH = std::min(pow2ceil(Ht), kBM / g.TW);
    g.NIMG = std::min(16, kBM / (g.TH * g.TW));
    if (Wt > g.TW || Ht > g.TH) g.NIMG = 1;   // several images share a tile only when a whole image fits in it
    tile_finish(g, B, Ht, Wt, sy, sx, taps);
    return g;
}

// Round 5: the tile shape follows the level's extent (the reference runs whatever patch size plans.json names -
// ts2d/core/inference/prediction_worker.py:76-77 - and nnU-Net patches such as 640 x 384 or 448 x 576 give levels of 80 x 48, 40 x 24,
// 56 x 72, 28 x 36 pixels).  The power-of-two tiling is kept when it is waste-free (bit-compatible with rounds 1-4 on 512^2 / 1024^2);
// otherwise: whole images per tile when at least two fit (any extent: 5 x 6, 10 x 6 ...), else the TH x TW <= 256 with the fewest
// tiles per image whose haloed patch fits the staging budget of EVERY kernel that may serve the op in any precision mode (the
// shape depends on (Ht, Wt, stride) only - never on B or on the mode, so the statistics partials are laid out the same everywhere).
TileGeom tile_geom(int B, int Ht, int Wt, int sy, int sx, int taps) {
    TileGeom g = tile_geom_pow2(B, Ht, Wt, sy, sx, taps);
    if (g.NIMG == 1 ? (Ht % g.TH == 0 && Wt % g.TW == 0) : (g.TH == Ht && g.TW == Wt)) return g;
    const int halo = (taps == 9) ? 3 : 1;
    auto patch = [&](int th, int tw) { return ((th - 1) * sy + halo) * ((tw - 1) * sx + halo); };
    const bool s1 = sy == 1 && sx == 1;
    // staging budgets (patch pixels): conv_mfma_f32 576 / 1296 (ConvCfg::MAXP), conv3x3_f16x3 640, conv3x3s2_f16x3 1536; preferred (one-image
    // kernels): conv3x3_f16x3_one / conv3x3_h32 384, conv3x3s2_f16x3_one 1280
    const int hard = taps == 1 ? (1 << 30) : (s1 ? 576 : 1296), soft = taps == 1 ? (1 << 30) : (s1 ? 384 : 1280);
    if (Ht * Wt * 2 <= kBM) {
        int n = std::min(16, kBM / (Ht * Wt));
        while (n > 1 && n * patch(Ht, Wt) > hard) --n;
        if (n > 1) { g.TH = Ht; g.TW = Wt; g.NIMG = n; tile_finish(g, B, Ht, Wt, sy, sx, taps); return g; }
    }
    int best_th = 0, best_tw = 0; long best_cost = -1;
    for (int pass = 0; pass < 2 && best_cost < 0; ++pass) {
        const int lim = pass == 0 ? soft : hard;
        for (int tw = std::min(Wt, kBM); tw >= 1; --tw) {
            int th = std::min(Ht, kBM / tw);
            while (th > 1 && patch(th, tw) > lim) --th;
            if (patch(th, tw) > lim) continue;
            th = (Ht + (Ht + th - 1) / th - 1) / ((Ht + th - 1) / th);      // the smallest TH with the same number of tile rows (even tiles)
            const long tiles = (long)((Ht + th - 1) / th) * ((Wt + tw - 1) / tw);
            // fewest tiles; then whole rows of 4 pixels (the 16-byte stores of a lane stay inside a tile row), then the wider tile
            const long cost = tiles * 4 + (tw % 4 ? 1 : 0);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_th = th; best_tw = tw; }
        }
    }
    g.TH = best_th; g.TW = best_tw; g.NIMG = 1;
    tile_finish(g, B, Ht, Wt, sy, sx, taps);
    return g;
}

// Tile of the composed decoder entry on a level that is no multiple of 8 x 32 (conv3x3_upc / conv3x3_upc_h, FLEX instances): TH x TW
// output pixels, both even, <= 256, coarse patch (TH/2 + 2)(TW/2 + 2) <= 128 and skip patch (TH + 2)(TW + 2) <= 384 pixels (the
// kernels' staging units); fewest tiles per image.  ok = false: no shape fills its tiles to 70 %.
TileGeom tile_geom_upc(int B, int Ht, int Wt, bool& ok) {
    ok = false;
    TileGeom g{};
    if (Ht % 2 || Wt % 2) return g;
    int best_th = 0, best_tw = 0; long best = -1;
    for (int tw = std::min(Wt, 128) / 2 * 2; tw >= 2; tw -= 2) {
        int th = std::min(Ht, 256 / tw) / 2 * 2;
        while (th >= 2 && ((th / 2 + 2) * (tw / 2 + 2) > 128 || (th + 2) * (tw + 2) > 384)) th -= 2;
        if (th < 2) continue;
        const int rows = (Ht + th - 1) / th;
        th = ((Ht + rows - 1) / rows + 1) / 2 * 2;            // the smallest even TH with the same number of tile rows
        const long tiles = (long)rows * ((Wt + tw - 1) / tw);
        if (best < 0 || tiles < best) { best = tiles; best_th = th; best_tw = tw; }
    }
    if (best < 0 || (double)Ht * Wt < 0.70 * 256.0 * (double)best) return g;
    ok = true;
    g.TH = best_th; g.TW = best_tw; g.NIMG = 1;
    tile_finish(g, B, Ht, Wt, 1, 1, 9);
    return g;
}

// Output tile of conv3x3s2_v2<.., FLEX> on a level that is no multiple of 8 x 32: TH | Ht, TW | Wt, TW % 4 == 0, TH * TW <= 256, a patch of
// (2 TH + 1) x (2 TW + 2) <= 1122 slots; the largest such tile; ok = false below 3/4 of 256 pixels.
TileGeom tile_geom_s2v2(int B, int Ht, int Wt, bool& ok) {
    ok = false;
    TileGeom g{};
    int best = 0;
    for (int tw = 4; tw <= std::min(Wt, 124); tw += 4) {
        if (Wt % tw) continue;
        for (int th = 1; th <= std::min(Ht, 256 / tw); ++th) {
            if (Ht % th || (2 * th + 1) * (2 * tw + 2) > 1122) continue;
            if (th * tw > best || (th * tw == best && tw > g.TW)) { best = th * tw; g.TH = th; g.TW = tw; }
        }
    }
    if (best < 192) return g;
    ok = true;
    g.NIMG = 1;
    tile_finish(g, B, Ht, Wt, 2, 2, 9);
    return g;
}

// the 8 x 32 / 16 x 32 tilings of the kernels that walk complete tiles of a fixed shape
TileGeom tile_fixed(int B, int Ht, int Wt, int th, int tw, int sy, int sx) {
    TileGeom g{};
    g.TH = th; g.TW = tw; g.NIMG = 1;
    tile_finish(g, B, Ht, Wt, sy, sx, 9);
    return g;
}


template <int TAPS, int SY, int SX, int CK, int BN, int EPI, typename ST = float>
hipError_t launch_conv_inst(const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = conv_mfma_f32<TAPS, SY, SX, CK, BN, EPI, ST>;
    if (hipError_t e = allow_max_lds(reinterpret_cast<const void*>(kern), attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), smem, st, a);
    return hipGetLastError();
}

// The generic implicit-GEMM kernel: the whole exact mode (fp32 storage), and - in every mode - the stages whose stride is neither
// (1, 1) nor (2, 2) (f16 = fp16 storage with the 16-bit mode's operand rounding).
hipError_t launch_conv(int taps, int sy, int sx, int ck, int bn, bool f16, const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    const bool b64 = bn == 64;
    if (bn != 32 && bn != 64) return hipErrorInvalidConfiguration;
    if (taps == 9 && sy == 1 && sx == 1 && !f16) {
        if (ck == 16) return b64 ? launch_conv_inst<9, 1, 1, 16, 64, 0>(a, grid, smem, st) : launch_conv_inst<9, 1, 1, 16, 32, 0>(a, grid, smem, st);
        if (ck == 8) return b64 ? launch_conv_inst<9, 1, 1, 8, 64, 0>(a, grid, smem, st) : launch_conv_inst<9, 1, 1, 8, 32, 0>(a, grid, smem, st);
    }
    if (taps == 9 && sy == 2 && sx == 2 && ck == 8 && !f16)
        return b64 ? launch_conv_inst<9, 2, 2, 8, 64, 0>(a, grid, smem, st) : launch_conv_inst<9, 2, 2, 8, 32, 0>(a, grid, smem, st);
    if (taps == 9 && sy == 2 && sx == 1 && ck == 8) {
        if (f16) return b64 ? launch_conv_inst<9, 2, 1, 8, 64, 0, _Float16>(a, grid, smem, st) : launch_conv_inst<9, 2, 1, 8, 32, 0, _Float16>(a, grid, smem, st);
        return b64 ? launch_conv_inst<9, 2, 1, 8, 64, 0>(a, grid, smem, st) : launch_conv_inst<9, 2, 1, 8, 32, 0>(a, grid, smem, st);
    }
    if (taps == 9 && sy == 1 && sx == 2 && ck == 8) {
        if (f16) return b64 ? launch_conv_inst<9, 1, 2, 8, 64, 0, _Float16>(a, grid, smem, st) : launch_conv_inst<9, 1, 2, 8, 32, 0, _Float16>(a, grid, smem, st);
        return b64 ? launch_conv_inst<9, 1, 2, 8, 64, 0>(a, grid, smem, st) : launch_conv_inst<9, 1, 2, 8, 32, 0>(a, grid, smem, st);
    }
    if (taps == 1 && ck == 16) {      // transposed conv, kernel = stride = (a.KA, a.KB)
        if (f16) return b64 ? launch_conv_inst<1, 1, 1, 16, 64, 1, _Float16>(a, grid, smem, st) : launch_conv_inst<1, 1, 1, 16, 32, 1, _Float16>(a, grid, smem, st);
        return b64 ? launch_conv_inst<1, 1, 1, 16, 64, 1>(a, grid, smem, st) : launch_conv_inst<1, 1, 1, 16, 32, 1>(a, grid, smem, st);
    }
    return hipErrorInvalidConfiguration;
}

template <int BN, int MAXU, typename ST, int NP>
hipError_t launch_split_inst(const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = conv3x3_f16x3<BN, MAXU, 1, ST, NP>;
    if (hipError_t e = allow_max_lds(reinterpret_cast<const void*>(kern), attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid, a.ksplit), dim3(kBlock), smem, st, a);
    return hipGetLastError();
}

template <typename ST, int NP>
hipError_t launch_split_t(int bn, int maxu, const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    if (bn == 32 && maxu == 3) return launch_split_inst<32, 3, ST, NP>(a, grid, smem, st);   // (2-chunk prefetch measured slower)
    if (bn == 32 && maxu == 5) return launch_split_inst<32, 5, ST, NP>(a, grid, smem, st);
    if (bn == 64 && maxu == 3) return launch_split_inst<64, 3, ST, NP>(a, grid, smem, st);
    if (bn == 64 && maxu == 5) return launch_split_inst<64, 5, ST, NP>(a, grid, smem, st);
    return hipErrorInvalidConfiguration;
}
hipError_t launch_split(bool f16, int bn, int maxu, const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    return f16 ? launch_split_t<_Float16, 1>(bn, maxu, a, grid, smem, st) : launch_split_t<float, 3>(bn, maxu, a, grid, smem, st);
}

template <int BN, bool PFS>
hipError_t launch_one_inst(const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = conv3x3_f16x3_one<BN, PFS>;
    if (hipError_t e = allow_max_lds(reinterpret_cast<const void*>(kern), attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid, a.ksplit), dim3(kBlock), smem, st, a);
    return hipGetLastError();
}
template <int BN, bool PFS, bool PIPE, typename ST, int NP>
hipError_t launch_one_s2_inst(const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = conv3x3s2_f16x3_one<BN, PFS, PIPE, ST, NP>;
    if (hipError_t e = allow_max_lds(reinterpret_cast<const void*>(kern), attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid, a.ksplit), dim3(kBlock), smem, st, a);
    return hipGetLastError();
}
hipError_t launch_one_s2(bool f16, int bn, const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    if (bn == 32) return f16 ? launch_one_s2_inst<32, false, false, _Float16, 1>(a, grid, smem, st) : launch_one_s2_inst<32, false, false, float, 3>(a, grid, smem, st);
    // (measured: prefetching the scale/shift vectors or double-buffering the fragments costs registers and gains nothing here)
    if (bn == 64) return f16 ? launch_one_s2_inst<64, false, false, _Float16, 1>(a, grid, smem, st) : launch_one_s2_inst<64, false, false, float, 3>(a, grid, smem, st);
    return hipErrorInvalidConfiguration;
}
hipError_t launch_one(int bn, const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    if (bn == 32) return launch_one_inst<32, false>(a, grid, smem, st);     // 3 workgroups per CU: no room for the prefetched scale/shift
    if (bn == 64) return launch_one_inst<64, true>(a, grid, smem, st);
    return hipErrorInvalidConfiguration;
}

template <int BN, int MAXU>
hipError_t launch_h32_inst(const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = conv3x3_h32<BN, MAXU>;
    if (hipError_t e = allow_max_lds(reinterpret_cast<const void*>(kern), attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid, a.ksplit), dim3(kBlock), smem, st, a);
    return hipGetLastError();
}
hipError_t launch_h32(int bn, const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    if (bn == 32) return launch_h32_inst<32, 6>(a, grid, smem, st);
    if (bn == 64) return launch_h32_inst<64, 6>(a, grid, smem, st);
    return hipErrorInvalidConfiguration;
}

template <int BN, int MAXU, typename ST, int NP>
hipError_t launch_split_s2_inst(const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = conv3x3s2_f16x3<BN, MAXU, ST, NP>;
    if (hipError_t e = allow_max_lds(reinterpret_cast<const void*>(kern), attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid, a.ksplit), dim3(kBlock), smem, st, a);
    return hipGetLastError();
}

template <typename ST, int NP>
hipError_t launch_split_s2_t(int bn, int maxu, const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    if (bn == 32 && maxu == 5) return launch_split_s2_inst<32, 5, ST, NP>(a, grid, smem, st);
    if (bn == 32 && maxu == 6) return launch_split_s2_inst<32, 6, ST, NP>(a, grid, smem, st);
    if (bn == 64 && maxu == 5) return launch_split_s2_inst<64, 5, ST, NP>(a, grid, smem, st);
    if (bn == 64 && maxu == 6) return launch_split_s2_inst<64, 6, ST, NP>(a, grid, smem, st);
    return hipErrorInvalidConfiguration;
}
hipError_t launch_split_s2(bool f16, int bn, int maxu, const ConvArgs& a, int grid, size_t smem, hipStream_t st) {
    return f16 ? launch_split_s2_t<_Float16, 1>(bn, maxu, a, grid, smem, st) : launch_split_s2_t<float, 3>(bn, maxu, a, grid, smem, st);
}

// per-tile partials -> scale / shift; the tile-lane count depends on the layer geometry only (never on B)
template <typename... A>
void launch_finalize(int B, int cout, hipStream_t st, const float* part, int ntiles, A... rest) {
    // (8-channel blocks - four times the grid for the 32- / 64-channel levels - measured no faster: 13.2 vs 12.5 us)
    if (ntiles >= 128) hipLaunchKernelGGL(finalize_stats_t<32>, dim3(B, cout / 32), dim3(1024), 0, st, part, ntiles, rest...);
    else hipLaunchKernelGGL(finalize_stats_t<8>, dim3(B, cout / 32), dim3(256), 0, st, part, ntiles, rest...);
}

void launch_stats_direct(bool f16, int B, int C, int HW, const float* x, const float* g, const float* be, float eps, float* sc, float* sh, hipStream_t st) {
    if (f16) hipLaunchKernelGGL(stats_direct<_Float16>, dim3(B, C / 32), dim3(256), 0, st, reinterpret_cast<const _Float16*>(x), C, HW, g, be, eps, sc, sh);
    else hipLaunchKernelGGL(stats_direct<float>, dim3(B, C / 32), dim3(256), 0, st, x, C, HW, g, be, eps, sc, sh);
}

// Split-K factor for a split-fp16 conv whose grid would leave most CUs idle (8x8 / 4x4 bottleneck levels).
int choose_ksplit(const Op& op, const TileGeom& g) {
    if (op.type != OP_CONV || !op.split_ok || op.first_direct) return 1;
    // The factor depends on the layer geometry only (never on B), so a slice computes bit-identically alone or in a batch:
    // tiles that hold >= 4 whole images (<= 8x8 pixels per image) leave most CUs idle -> split K as far as 4 chunks/slice.
    if (g.NIMG < 4) return 1;
    const int nchunks = (op.cin + op.cin_skip) / (op.stride == 1 ? 16 : 8);
    int S = 1;
    while (S < 8 && nchunks / (S * 2) >= 4) S *= 2;
    return S;
}

// Small batches (round 6).  TS2D.predict runs tiles x mirrors = 8 slices per sub-model and the reference one (B = 1 per network() call,
// ts2d/core/inference/prediction_worker.py:209): at the <= 32 x 32 levels the persistent / composed kernels then launch 4 ... 64 workgroups on 256 CUs, each
// walking 16 ... 64 channel chunks in series (B = 1: enc5.c0 0.21 ms for 0.6 GMAC).  When the one-image kernel of an op would launch fewer workgroups than
// the chip has CUs, K is split until it does (>= 4 chunks per slice; deterministic two-phase reduction, splitk_reduce_stats), and a composed decoder entry
// runs as transposed conv + conv (the composed kernels have no split-K form).  The factor depends on B: a slice is bit-identical alone or in a batch only
// while both batches take the same path (tests compare across regimes at 2e-6); the same B always gives the same bits.
// Measured (profiles/r06_small_batch.txt, wall time of a forward, split mode): B = 1 2.51 -> 1.90 ms, B = 2 2.77 -> 2.15, B = 4 3.27 -> 2.82, B = 8 4.32 -> 3.98;
// filling to TWO workgroups per CU instead of one: equal (the reduction pass of each further split op costs what its conv gains).
constexpr size_t kSbkElems = (size_t)2 * 256 * 256 * 64;      // S x B x HW x Cout of any op this rule splits (n_wgs S < 2 x 256 CUs, <= 256 pixels x 64 columns each)
int fill_ksplit(const ts2d_engine* e, long long n_wgs, int nchunks) {
    if (!e->use_sbk || e->num_cus > 256) return 1;            // (the bound above assumes <= 256 CUs)
    int S = 1;
    while (S < 8 && n_wgs * S < e->num_cus && nchunks / (S * 2) >= 4) S *= 2;
    return S;
}

// ------------------------------------------------------------------------------------------------------------------ dispatch
// ONE function decides which kernel serves an op for (precision, options, B, H, W).  The activation plan (which tensors exist), the
// workspace sizing and the run (what is launched) all ask it, so they cannot disagree (the eligibility tests
// used to be written at plan time and again inline at launch; round 3's plan / run mismatch came from exactly that).
inline bool fits32(size_t bytes) { return bytes < ((size_t)1 << 31); }      // an image addressed through a 32-bit buffer offset
// the persistent kernels decode a tile number with udiv_magic (kernels.h): exact for tile number x tiles per image < 2^32
inline bool magic_ok(const TileGeom& g) { return (unsigned long long)g.n_mtiles * (unsigned)(g.tiles_x * g.tiles_y) < 0x100000000ull; }

// Does the first block run as a statistics-only pass, recomputed inside the second block (conv3x3_res32<.., FUSE>)?
bool fuse0_applies(const ts2d_engine* e, int H, int W) {
    // (split mode only.  Measured in the 16-bit mode, B = 64 canonical: 0.30 + 0.94 ms fused against 0.44 + 0.52 ms as two kernels - the
    //  16-bit second block is HBM-bound at a third of the split block's MFMA work, and the recompute is fp32 MFMA work either way)
    //  Round 6, with the recompute on the fp16 matrix path and the statistics pass at 0.25 ms: 0.25 + 0.79 fused against 0.33 + 0.54 - the 16-bit
    //  conversion of the recomputed values (round to fp16 as if stored, normalise, LeakyReLU in fp16) is what the fused kernel is bound by)
    if (!e->use_fuse0 || !e->use_res || !e->use_one || e->precision != TS2D_PRECISION_F32_SPLIT_F16X3 || e->ops.size() < 3) return false;
    const Op& o0 = e->ops[0]; const Op& o1 = e->ops[1];
    if (!o0.first_direct || o0.cout != 32 || e->arch.input_channels > 2) return false;
    if (o1.type != OP_CONV || !o1.res_ok || o1.src != o0.dst || o1.skip >= 0) return false;
    for (size_t i = 2; i < e->ops.size(); ++i) if (e->ops[i].src == o0.dst || e->ops[i].skip == o0.dst) return false;      // (a one-conv stage: the tensor is a skip)
    if (H % 8 || W % 32) return false;
    return fits32((size_t)H * W * 32 * 4) && fits32((size_t)e->arch.input_channels * H * W * 4);
}

// The decoder block `op` (3x3 conv over cat(up, skip)) as ONE kernel together with its transposed conv (kernels_upc.h ...), or K_NONE.
Kern composed_kernel(const ts2d_engine* e, const Op& op, int B, int H, int W) {
    if (!op.upc_ok || !e->use_upc || !e->use_one || e->precision == TS2D_PRECISION_F32_EXACT || B < 1) return K_NONE;
    const bool f16 = e->precision == TS2D_PRECISION_F16;
    const Op& up = e->ops[op.up_idx];
    const bool srcs_normed = e->tensors[up.src].normed && e->tensors[op.skip].normed;       // (`normed`, not the scale POINTERS: the plan asks
                                                                                           //  before the workspace - and the pointers - exist)
    if (f16 && (op.cin_skip % 32 || !srcs_normed)) return K_NONE;       // (the 16-bit kernels walk the skip channels in chunks of 32 and normalise both sources)
    const int Ht = H >> op.ly, Wt = W >> op.lx;
    if (!(op.up0_ok && e->use_up0)) {
        // small batch: the two-kernel path when its conv would run with split-K (fill_ksplit) and the upsampled tensor fits the scratch region
        const TileGeom g1 = tile_geom(B, Ht, Wt, 1, 1, 9);
        const int chunk = f16 ? 32 : 16, bn1 = op.cout % 64 == 0 ? 64 : 32;
        const bool two_ok = up.split_ok && (f16 ? op.h32_ok : op.split_ok) && g1.NIMG == 1 && (size_t)B * Ht * Wt * up.cout <= kSbkElems / 2;
        if (two_ok && fill_ksplit(e, (long long)g1.n_mtiles * (op.cout / bn1), ct_total(op) / chunk) > 1) return K_NONE;
    }
    if (Ht % 8 || Wt % 32) {                                            // no complete 8 x 32 tiles: tiles that follow the extent (FLEX instances)
        bool ok = false;
        (void)tile_geom_upc(B, Ht, Wt, ok);
        if (!ok || !e->use_flex || !srcs_normed || op.cout % 32) return K_NONE;
        if (!fits32((size_t)Ht * Wt * std::max(op.cout, op.cin_skip) * 4) || !fits32((size_t)(Ht / 2) * (Wt / 2) * up.cin * 4)) return K_NONE;
        return f16 ? K_UPC_H : K_UPC;
    }
    if (op.up0_ok && e->use_up0 && srcs_normed && fits32((size_t)Ht * Wt * 32 * 4)) return K_UP0;
    if (!fits32((size_t)Ht * Wt * std::max(op.cout, op.cin_skip) * 4) || !fits32((size_t)(Ht / 2) * (Wt / 2) * up.cin * 4)) return K_NONE;
    const int bn = op.cout % 64 == 0 ? 64 : 32;

    if (f16) return (e->use_uh2 && bn == 64 && Ht % 16 == 0) ? K_UPC_H2 : K_UPC_H;
    const bool upq = e->use_upq && bn == 64 && up.cin >= e->upq_min && Ht % 16 == 0 && srcs_normed && up.cin <= 512 && op.cin_skip <= 512;
    return upq ? K_UPQ : K_UPC;           // (Cb = 128: conv3x3_upq no faster than conv3x3_upc, measured)
}

Choice choose(const ts2d_engine* e, size_t oi, int B, int H, int W) {
    const Op& op = e->ops[oi];
    Choice c;
    const bool f16 = e->precision == TS2D_PRECISION_F16, exact = e->precision == TS2D_PRECISION_F32_EXACT;
    if (op.type == OP_HEAD) {
        const bool hm = op.split_ok && !exact && e->use_one && e->tensors[op.src].C == 32 && (H * W) % 32 == 0 && W % 32 == 0;
        c.k = hm ? K_HEAD_MFMA : K_HEAD_1X1;
        return c;
    }
    if (op.first_direct) {
        c.g = tile_geom_pow2(B, H, W, 1, 1, 9);
        const int kp = (op.cin + 1) / 2, P = c.g.PH * c.g.PW * c.g.NIMG;
        c.fused_stats = c.g.NIMG == 1;
        c.first_full = c.g.NIMG == 1 && c.g.lgTW >= 4 && c.g.lgTH + c.g.lgTW == 8 && H % c.g.TH == 0 && W % c.g.TW == 0 &&
                       P * 2 * kp <= 4 * kBlock && fits32((size_t)H * W * op.cout * 4);
        c.k = fuse0_applies(e, H, W) ? K_FIRST_STATS : K_FIRST;
        return c;
    }
    const Tensor& src = e->tensors[op.src];
    const int Hin = H >> src.ly, Win = W >> src.lx;
    if (op.type == OP_CONVT) {
        if (oi + 1 < e->ops.size() && e->ops[oi + 1].up_idx == (int)oi && composed_kernel(e, e->ops[oi + 1], B, H, W) != K_NONE) {
            c.k = K_FUSED_AWAY;             // composed into the next block: the upsampled tensor is never materialised
            return c;
        }
        c.g = tile_geom(B, Hin, Win, 1, 1, 1);          // (the transposed conv tiles its INPUT pixels)
        const int N = op.sy * op.sx * op.cout;
        if (!(op.split_ok && !exact)) {                 // exact mode, or a kernel other than 2 x 2: the generic kernel in every mode
            if (f16 && op.stride == 2) return c;        // (K_NONE: a 2 x 2 transposed conv without a 16-bit kernel - Cin % 32 != 0)
            c.k = K_EXACT; c.bn = (N % 64 == 0 && op.cout % 64 == 0) ? 64 : 32;
            return c;
        }
        c.bn = 64;                                      // N = 4 * Cout is a multiple of 128; each 32-column tile lies in one (a, b) tap
        c.k = (e->use_one && c.g.NIMG == 1 && fits32((size_t)4 * Hin * Win * op.cout * 4) && fits32((size_t)Hin * Win * op.cin * 4)) ? K_T_ONE : K_T_GENERIC;
        return c;
    }
    const int Ht = H >> op.ly, Wt = W >> op.lx, ct = ct_total(op);
    if (op.up_idx >= 0) {
        const Kern ck = composed_kernel(e, op, B, H, W);
        if (ck != K_NONE) {
            c.k = ck; c.fused_stats = true;
            if (ck == K_UP0 && f16) c.ppt = 4;
            c.bn = op.cout % 64 == 0 ? 64 : 32;
            if (Ht % 8 || Wt % 32) { bool ok = false; c.g = tile_geom_upc(B, Ht, Wt, ok); c.flex = true; }
            else c.g = tile_fixed(B, Ht, Wt, (ck == K_UPQ || ck == K_UPC_H2) ? 16 : 8, 32, 1, 1);
            return c;
        }
    }
    c.bn = op.cout % 64 == 0 ? 64 : 32;
    if (!(op.split_ok && !exact)) {
        if (f16 && op.stride != 3) return c;            // (K_NONE: no 16-bit kernel - channel counts must be multiples of 16)
        c.k = K_EXACT;
        c.g = tile_geom(B, Ht, Wt, op.sy, op.sx, 9);
        c.fused_stats = c.g.NIMG == 1;
        return c;
    }
    const bool img32 = fits32((size_t)Hin * Win * std::max(op.cin, op.cin_skip) * 4) && fits32((size_t)Ht * Wt * op.cout * 4);
    if (op.stride == 2) {
        // small batch: the one-image kernel with split-K where it leaves most CUs idle (fill_ksplit); else the 512-thread kernel
        int s2k = 1;
        {
            const TileGeom g1 = tile_geom(B, Ht, Wt, 2, 2, 9);
            const int P1 = g1.PH * g1.PW * g1.NIMG;
            if (e->use_one && g1.NIMG == 1 && P1 <= 5 * kBlock && img32) s2k = fill_ksplit(e, (long long)g1.n_mtiles * (op.cout / c.bn), op.cin / 8);
        }
        if (s2k == 1 && op.s2v2_ok && e->use_s2v2 && e->use_one && Ht % 8 == 0 && Wt % 32 == 0 && img32 && magic_ok(tile_fixed(B, Ht, Wt, 8, 32, 2, 2))) {
            // stride-2 block on complete 8 x 32 output tiles: one 512-thread workgroup per CU, up to 128 output columns
            c.k = K_S2_V2; c.bn = op.bn2; c.fused_stats = true;
            c.g = tile_fixed(B, Ht, Wt, 8, 32, 2, 2);
            return c;
        }
        if (s2k == 1 && op.s2v2_ok && e->use_s2v2 && e->use_one && e->use_flex2 && img32 && (f16 || e->use_flex2 > 1)) {
            // ... on the tile that divides the level (FLEX instance)
            bool ok = false;
            const TileGeom gs = tile_geom_s2v2(B, Ht, Wt, ok);
            if (ok && magic_ok(gs)) { c.k = K_S2_V2; c.bn = op.bn2; c.fused_stats = true; c.flex = true; c.g = gs; return c; }
        }
        c.g = tile_geom(B, Ht, Wt, 2, 2, 9);
        c.ksplit = std::max(choose_ksplit(op, c.g), s2k);
        c.fused_stats = c.g.NIMG == 1 && c.ksplit == 1;
        const int P = c.g.PH * c.g.PW * c.g.NIMG;
        c.k = (e->use_one && c.g.NIMG == 1 && P <= 5 * kBlock && img32) ? K_S2_ONE : K_S2_GENERIC;
        return c;
    }
    if (op.res_ok && e->use_res && e->use_one && src.normed && op.skip < 0 && Ht % 8 == 0 && Wt % 32 == 0 && fits32((size_t)Ht * Wt * 32 * 4)) {
        // 32 -> 32 stride-1 block on complete 8 x 32 tiles: persistent kernel with the layer's weights resident in LDS
        c.k = (oi == 1 && fuse0_applies(e, H, W)) ? K_S1_RES32F : K_S1_RES32;
        c.fused_stats = true;
        c.g = tile_fixed(B, Ht, Wt, 8, 32, 1, 1);
        return c;
    }
    c.g = tile_geom(B, Ht, Wt, 1, 1, 9);
    c.ksplit = choose_ksplit(op, c.g);
    c.fused_stats = c.g.NIMG == 1;
    const int P = c.g.PH * c.g.PW * c.g.NIMG;
    const bool srcs_normed = src.normed && (op.skip < 0 || e->tensors[op.skip].normed);
    const bool tiles16 = Ht % 16 == 0 && Wt % 32 == 0 && c.g.NIMG == 1 && img32;          // complete 16 x 32 tiles
    // small batch: split-K on the one-image kernels where they leave most CUs idle (fill_ksplit) - ahead of the 512-thread kernels, which have no split-K form
    if (c.g.NIMG == 1 && img32 && op.cout % c.bn == 0) {
        const bool can = f16 ? (op.h32_ok && e->use_h32 && P * 4 <= 6 * kBlock) : (e->use_one && P * 2 <= 3 * kBlock);
        const int sk = can ? fill_ksplit(e, (long long)c.g.n_mtiles * (op.cout / c.bn), ct / (f16 ? 32 : 16)) : 1;
        if (sk > 1) {
            c.ksplit = sk; c.fused_stats = false;
            c.k = f16 ? K_S1_H32 : K_S1_ONE;
            return c;
        }
    }
    if (f16) {
        if (e->use_h2 && tiles16 && op.cout % 64 == 0 && op.skip < 0 && ct % 32 == 0 && src.normed && ct >= e->h2_min) {
            c.k = K_S1_H2; c.bn = 64;       // plain C -> C block on 16 x 32 tiles: the skip phase of conv3x3_upc_h2
            c.g = tile_fixed(B, Ht, Wt, 16, 32, 1, 1);
            return c;
        }
        c.k = (op.h32_ok && e->use_h32 && c.g.NIMG == 1 && P * 4 <= 6 * kBlock && img32) ? K_S1_H32 : K_S1_GENERIC;
        return c;
    }
    const bool one = e->use_one && c.g.NIMG == 1 && P * 2 <= 3 * kBlock && img32;
    if (one && e->use_q && tiles16 && c.bn == 64 && ct >= 64 && srcs_normed && magic_ok(tile_fixed(B, Ht, Wt, 16, 32, 1, 1))) {       // (pays off from 4 chunks on: measured)
        // (round 5: an extent-following variant of this kernel was built and measured - tile = TH x TW <= 512 pixels, row -> pixel by division:
        //  correct, but at 256 registers per wave its two extra address registers spill, and a scratch reload per item in front of the
        //  MFMA stream made it 10-35 % SLOWER than conv3x3_f16x3_one on its extent-following 256-pixel tiles; ragged levels keep that kernel)
        c.k = K_S1_QP;                      // one persistent 512-thread workgroup per CU, patch and weights double-buffered
        c.g = tile_fixed(B, Ht, Wt, 16, 32, 1, 1);
        return c;
    }
    c.k = one ? K_S1_ONE : K_S1_GENERIC;
    return c;
}

// choose() for every op, cached per (B, H, W, precision, option generation)
const std::vector<Choice>& planned(ts2d_engine* e, int B, int H, int W) {
    if (e->plan.size() != e->ops.size() || e->planB != B || e->planH != H || e->planW != W || e->plan_prec != e->precision || e->plan_gen != e->opt_gen) {
        e->plan.resize(e->ops.size());
        for (size_t i = 0; i < e->ops.size(); ++i) e->plan[i] = choose(e, i, B, H, W);
        e->planB = B; e->planH = H; e->planW = W; e->plan_prec = e->precision; e->plan_gen = e->opt_gen;
    }
    return e->plan;
}

size_t partial_floats_needed(const ts2d_engine* e, int B, int H, int W) {
    size_t mx = 0;
    for (size_t i = 0; i < e->ops.size(); ++i) {
        const Op& op = e->ops[i];
        const Choice c = choose(e, i, B, H, W);
        if (c.ksplit > 1) mx = std::max(mx, (size_t)c.ksplit * B * (H >> op.ly) * (W >> op.lx) * op.cout);
    }
    // a workspace sized for B also serves every smaller batch, whose ops may split K further (fill_ksplit): the bound of that rule
    if (e->use_sbk) mx = std::max(mx, kSbkElems);
    return mx;
}

size_t part_floats_needed(const ts2d_engine* e, int B, int H, int W) {
    size_t mx = 0;
    for (size_t i = 0; i < e->ops.size(); ++i) {
        const Op& op = e->ops[i];
        if (op.type != OP_CONV) continue;
        const Choice c = choose(e, i, B, H, W);
        if (c.fused_stats) mx = std::max(mx, (size_t)B * c.g.tiles_x * c.g.tiles_y * c.ppt * op.cout * 4);      // (S, Q, K, n) per (tile, channel)
        if (e->use_sbk) mx = std::max(mx, kSbkElems / 256 * 4);      // small-batch split-K (any smaller batch in this workspace): a partial per 256 pixels and channel
    }
    return mx;
}


// Activation plan (round 3): liveness-based reuse.  An activation lives from the op that writes it to the last op that reads it
// (the encoder skips until their decoder block); its bytes then return to a first-fit free list inside ONE arena, sized by
// simulating the program.  Canonical net: 340 -> ~110 MB per slice.  A composed decoder entry (kernels_upc.h) reads the COARSE tensor
// and never materialises `decN.up`, so the plan depends on the precision mode - it is remade when that changes.
// keep_activations: no reuse (every tensor keeps its own buffer for ts2d_engine_debug_tensor / the non-finite diagnosis).
struct ActPlan { std::vector<size_t> off; std::vector<char> used, reused; size_t bytes = 0; };

ActPlan plan_activations(const ts2d_engine* e, int B, int H, int W, bool keep) {
    const size_t nt = e->tensors.size(), no = e->ops.size();
    ActPlan p; p.off.assign(nt, 0); p.used.assign(nt, 0); p.reused.assign(nt, 0);
    auto bytes_of = [&](size_t t) { const Tensor& x = e->tensors[t]; return align_up((size_t)B * (H >> x.ly) * (W >> x.lx) * x.C * sizeof(float), 256); };
    // which ops run, what they read
    std::vector<char> skipped(no, 0);
    std::vector<std::vector<int>> reads(no);
    const bool fused0 = fuse0_applies(e, H, W);            // first block recomputed inside the second: its output tensor is never materialised
    for (size_t i = 0; i < no; ++i) {
        const Op& op = e->ops[i];
        if (fused0 && i == 1) continue;                    // (reads the network input, which is not part of the arena)
        if (op.type == OP_CONVT && choose(e, i, B, H, W).k == K_FUSED_AWAY) { skipped[i] = 1; continue; }
        if (op.type == OP_CONV && op.up_idx >= 0 && skipped[op.up_idx]) { reads[i] = {e->ops[op.up_idx].src, op.skip}; continue; }
        if (!(op.first_direct)) reads[i].push_back(op.src);
        if (op.skip >= 0) reads[i].push_back(op.skip);
    }
    std::vector<int> last(nt, -1);
    for (size_t i = 0; i < no; ++i) for (int t : reads[i]) last[t] = (int)i;
    struct Blk { size_t off, size; };
    std::vector<Blk> freel;
    size_t top = 0;
    auto alloc = [&](size_t sz, bool& recycled) {
        int best = -1;
        for (size_t k = 0; k < freel.size(); ++k) if (freel[k].size >= sz && (best < 0 || freel[k].size < freel[best].size)) best = (int)k;
        if (best >= 0) { const size_t o = freel[best].off; freel[best].off += sz; freel[best].size -= sz; if (!freel[best].size) freel.erase(freel.begin() + best); recycled = true; return o; }
        if (!freel.empty() && freel.back().off + freel.back().size == top) {      // grow the arena from a free block at its end
            const size_t o = freel.back().off; top = o + sz; freel.pop_back(); recycled = true; return o;
        }
        const size_t o = top; top += sz; recycled = false; return o;
    };
    auto release = [&](size_t off, size_t sz) {
        size_t k = 0;
        while (k < freel.size() && freel[k].off < off) ++k;
        freel.insert(freel.begin() + k, Blk{off, sz});
        if (k + 1 < freel.size() && freel[k].off + freel[k].size == freel[k + 1].off) { freel[k].size += freel[k + 1].size; freel.erase(freel.begin() + k + 1); }
        if (k > 0 && freel[k - 1].off + freel[k - 1].size == freel[k].off) { freel[k - 1].size += freel[k].size; freel.erase(freel.begin() + k); }
    };
    std::vector<std::pair<size_t, size_t>> live(nt, {0, 0});
    if (!e->ops[0].first_direct) { bool r; p.off[0] = alloc(bytes_of(0), r); p.used[0] = 1; live[0] = {p.off[0], bytes_of(0)}; }
    for (size_t i = 0; i < no; ++i) {
        if (skipped[i]) continue;
        const Op& op = e->ops[i];
        if (fused0 && i == 0) continue;                      // statistics only: no output tensor
        if (op.dst >= 0) {                                   // the output is placed while the inputs are still allocated: never on top of them
            bool r = false;
            p.off[op.dst] = alloc(bytes_of(op.dst), r); p.used[op.dst] = 1; live[op.dst] = {p.off[op.dst], bytes_of(op.dst)};
            if (last[op.dst] < 0 && !keep) release(live[op.dst].first, live[op.dst].second);       // (never read: e.g. a net whose last tensor feeds nothing)
        }
        if (!keep)
            for (int t : reads[i]) if (last[t] == (int)i && live[t].second) { release(live[t].first, live[t].second); live[t].second = 0; }
    }
    // a tensor's values survive the run unless a LATER tensor overlaps its block
    for (size_t a = 0; a < nt; ++a) {
        if (!p.used[a]) continue;
        for (size_t b2 = 0; b2 < nt; ++b2) {
            if (!p.used[b2] || a == b2) continue;
            const bool later = b2 > a;                       // tensors are numbered in program order of their producers
            if (later && p.off[b2] < p.off[a] + bytes_of(a) && p.off[a] < p.off[b2] + bytes_of(b2)) p.reused[a] = 1;
        }
    }
    p.bytes = top;
    return p;
}

// Layout of the activation workspace for (B, H, W) under the engine's current precision mode, options and keep flag.
struct WsLayout { ActPlan plan; std::vector<size_t> o_sc, o_sh; size_t o_part = 0, o_pk = 0, o_up = 0, bytes = 0; };

WsLayout workspace_layout(const ts2d_engine* e, int B, int H, int W) {
    WsLayout L;
    L.plan = plan_activations(e, B, H, W, e->keep_activations);
    for (size_t& o : L.plan.off) o += kWsHeader;          // (the header: owner token of a shared workspace)
    size_t off = align_up(kWsHeader + L.plan.bytes, 256);
    L.o_sc.assign(e->tensors.size(), 0); L.o_sh.assign(e->tensors.size(), 0);
    for (size_t i = 0; i < e->tensors.size(); ++i) {
        const Tensor& t = e->tensors[i];
        if (t.normed) {
            L.o_sc[i] = off; off = align_up(off + (size_t)B * t.C * sizeof(float), 256);
            L.o_sh[i] = off; off = align_up(off + (size_t)B * t.C * sizeof(float), 256);
        }
    }
    L.o_part = off; off = align_up(off + part_floats_needed(e, B, H, W) * sizeof(float) + 256, 256);
    L.o_pk = off; off = align_up(off + partial_floats_needed(e, B, H, W) * sizeof(float) + 256, 256);   // split-K partials
    // one upsampled tensor of a small batch's two-kernel decoder entry (composed_kernel): the plan above is the reserved batch's, where that entry is composed
    L.o_up = off; if (e->use_sbk) off = align_up(off + kSbkElems / 2 * sizeof(float) + 256, 256);
    L.bytes = off;
    return L;
}

int ensure_workspace(ts2d_engine* e, int B, int H, int W) {
    const bool keep = e->keep_activations;
    if (e->d_ws && e->wsB >= B && e->wsH == H && e->wsW == W && e->ws_precision == e->precision && e->ws_keep == keep) return TS2D_OK;
    HIP_TRY(hipSetDevice(e->device));
    if (e->d_ws && e->wsH == H && e->wsW == W && e->wsB > B && !e->ws_external) B = e->wsB;      // same geometry, another mode: keep the larger batch capacity
    const WsLayout L = workspace_layout(e, B, H, W);
    if (e->d_ws) {      // the old workspace may still be in use by a run on ANY stream: wait for its end-of-run event
        if (e->ws_busy) { HIP_TRY(hipEventSynchronize(e->ws_event)); e->ws_busy = false; }
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    if (e->ws_external) {      // caller-provided memory (ts2d_engine_set_workspace): never allocated or freed here
        if (e->ws_bytes < L.bytes)
            return fail(TS2D_ERR_NOMEM, "the workspace given to ts2d_engine_set_workspace holds %zu bytes, B=%d H=%d W=%d needs %zu "
                        "(ts2d_engine_workspace_bytes)", e->ws_bytes, B, H, W, L.bytes);
    } else {
        if (e->d_ws && e->ws_bytes < L.bytes) { HIP_TRY(hipFree(e->d_ws)); e->d_ws = nullptr; e->ws_bytes = 0; }      // (a mode change that fits re-maps the same memory)
        if (!e->d_ws) { HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_ws), L.bytes)); e->ws_bytes = L.bytes; }
    }
    e->wsB = B; e->wsH = H; e->wsW = W; e->ws_precision = e->precision; e->ws_keep = keep;
    for (size_t i = 0; i < e->tensors.size(); ++i) {
        Tensor& t = e->tensors[i];
        t.data = L.plan.used[i] ? reinterpret_cast<float*>(e->d_ws + L.plan.off[i]) : nullptr;
        t.resident = L.plan.used[i] && !L.plan.reused[i];
        t.scale = t.normed ? reinterpret_cast<float*>(e->d_ws + L.o_sc[i]) : nullptr;
        t.shift = t.normed ? reinterpret_cast<float*>(e->d_ws + L.o_sh[i]) : nullptr;
    }
    e->d_part = reinterpret_cast<float*>(e->d_ws + L.o_part);
    e->d_partial = reinterpret_cast<float*>(e->d_ws + L.o_pk);
    e->d_up = e->use_sbk ? reinterpret_cast<float*>(e->d_ws + L.o_up) : nullptr;
    return TS2D_OK;
}

// Staging memory of the HOST-buffer forward (input, logits, masks of one batch): allocated on the first such call only - a caller
// that passes device pointers (the bench, the multi-GPU stream, SubModelSet) never pays for it (1.3 GB at B = 64, K = 18).
int ensure_staging(ts2d_engine* e, int B, int H, int W) {
    const int K = e->arch.num_classes;
    size_t off = 0;
    const size_t o_in = off; off = align_up(off + (size_t)B * e->arch.input_channels * H * W * sizeof(float), 256);
    const size_t o_lg = off; off = align_up(off + (size_t)B * K * H * W * sizeof(float), 256);
    const size_t o_mk = off; off = align_up(off + (size_t)B * K * H * ((W + 31) / 32) * sizeof(uint32_t), 256);
    if (off > e->stage_bytes) {
        HIP_TRY(hipSetDevice(e->device));
        if (e->ws_busy) { HIP_TRY(hipEventSynchronize(e->ws_event)); e->ws_busy = false; }
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (e->d_stage) { HIP_TRY(hipFree(e->d_stage)); e->d_stage = nullptr; e->stage_bytes = 0; }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_stage), off));
        e->stage_bytes = off;
    }
    e->d_in_stage = reinterpret_cast<float*>(e->d_stage + o_in);
    e->d_logit_stage = reinterpret_cast<float*>(e->d_stage + o_lg);
    e->d_mask_stage = reinterpret_cast<uint32_t*>(e->d_stage + o_mk);
    return TS2D_OK;
}

int prof_begin(ts2d_engine* e, const std::string& name, hipStream_t st) {
    if (!e->profiling) return TS2D_OK;
    if (e->n_launched >= e->launches.size()) {
        Launch l; l.name = name;
        HIP_TRY(hipEventCreate(&l.e0)); HIP_TRY(hipEventCreate(&l.e1));
        e->launches.push_back(l);
    }
    e->launches[e->n_launched].name = name;
    e->launches[e->n_launched].kernel.clear();
    HIP_TRY(hipEventRecord(e->launches[e->n_launched].e0, st));
    return TS2D_OK;
}
// which kernel served the op being profiled (ts2d_engine_op_kernel): called between prof_begin and prof_end
void prof_kernel(ts2d_engine* e, const std::string& kernel) {
    if (e->profiling && e->n_launched < e->launches.size()) e->launches[e->n_launched].kernel = kernel;
}
int prof_end(ts2d_engine* e, hipStream_t st) {
    if (!e->profiling) return TS2D_OK;
    HIP_TRY(hipEventRecord(e->launches[e->n_launched].e1, st));
    e->n_launched++;
    return TS2D_OK;
}

#define TRY(expr) do { int _rc = (expr); if (_rc != TS2D_OK) return _rc; } while (0)

int run_forward_impl(ts2d_engine* e, const float* d_in, int B, int H, int W, float* d_logits, uint32_t* d_mask, hipStream_t st);

// Order this run after the previous user of the workspace if that one ran on a different stream.
int workspace_acquire(ts2d_engine* e, hipStream_t st) {
    if (e->ws_busy && e->ws_stream != st) HIP_TRY(hipStreamWaitEvent(st, e->ws_event, 0));
    return TS2D_OK;
}
int workspace_release(ts2d_engine* e, hipStream_t st) {
    HIP_TRY(hipEventRecord(e->ws_event, st));
    e->ws_stream = st; e->ws_busy = true;
    return TS2D_OK;
}

// clear_flags: first batch of a call.  The non-finite flag is cleared HERE, behind workspace_acquire: the previous run's head kernel
// (possibly on another stream) sets it with atomicOr, and a memset issued before the stream is ordered behind that run could wipe
// or pre-empt it.  ts2d_engine_predict_tiled clears it once per call, not per chunk, so an earlier chunk's inf / NaN survives.
// Is the activation memory still this engine's?  Always for an allocation of its own; a shared workspace carries the token of the last
// run of ANY engine inside it.  (Synchronises: debug / diagnosis paths only.)
bool workspace_is_mine(ts2d_engine* e) {
    if (!e->ws_external) return true;
    if (!e->d_ws || !e->ws_token) return false;
    // Another engine of the set may have ENQUEUED a run on a caller's stream that has not executed yet (its stamp is a stream-ordered
    // memset): this engine's own event says nothing about that, so the token is read behind everything outstanding on the device.
    // (A fresh host thread's current device is 0: select the engine's own first.)
    if (hipSetDevice(e->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return false;
    uint32_t tok = 0;
    if (hipMemcpy(&tok, e->d_ws, sizeof(tok), hipMemcpyDeviceToHost) != hipSuccess) return false;
    return tok == e->ws_token;
}

int run_forward(ts2d_engine* e, const float* d_in, int B, int H, int W, float* d_logits, uint32_t* d_mask, hipStream_t st, bool clear_flags = true) {
    TRY(workspace_acquire(e, st));
    if (e->ws_external) {           // stamp the shared workspace: whatever another engine of the set left in it is gone after this run
        e->ws_token = ++g_ws_generation;
        if (!e->ws_token) e->ws_token = ++g_ws_generation;
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(e->d_ws), (int)e->ws_token, 1, st));
    }
    if (clear_flags) HIP_TRY(hipMemsetAsync(e->d_flags, 0, 2 * sizeof(int), st));
    const int rc = run_forward_impl(e, d_in, B, H, W, d_logits, d_mask, st);
    const int rc2 = workspace_release(e, st);       // also after a failed launch: earlier kernels of the run may be in flight
    return rc != TS2D_OK ? rc : rc2;
}

// ConvArgs fields that describe the pixel tiling / the (tile, column tile) grid
inline void set_tiling(ConvArgs& ca, const TileGeom& g, int n_ctiles) {
    ca.TH = g.TH; ca.TW = g.TW; ca.NIMG = g.NIMG; ca.lgTH = g.lgTH; ca.lgTW = g.lgTW;
    ca.inv_tw = 1.0f / (float)g.TW; ca.inv_thw = 1.0f / (float)(g.TH * g.TW);
    ca.tiles_x = g.tiles_x; ca.tiles_y = g.tiles_y; ca.n_mtiles = g.n_mtiles; ca.n_ctiles = n_ctiles;
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
    ca.mg_tx = magic(g.tiles_x); ca.mg_tpi = magic(g.tiles_x * g.tiles_y);
    ca.PH = g.PH; ca.PW = g.PW;
}

int run_forward_impl(ts2d_engine* e, const float* d_in, int B, int H, int W, float* d_logits, uint32_t* d_mask, hipStream_t st) {
    const ts2d_arch_desc& a = e->arch;
    e->n_launched = 0;
    e->fused_away.assign(e->ops.size(), 0);
    // (the input is scanned by ts2d_engine_check only when it lives in the engine's own staging memory)
    e->last_input = ((e->d_in_stage && d_in == e->d_in_stage) || (e->d_sw && reinterpret_cast<const char*>(d_in) >= e->d_sw &&
                                               reinterpret_cast<const char*>(d_in) < e->d_sw + e->sw_bytes)) ? d_in : nullptr;
    e->lastB = B; e->lastH = H; e->lastW = W; e->last_stream = st;
    const bool f16 = e->precision == TS2D_PRECISION_F16;      // fp16 storage, one fp16 MFMA product, fp32 accumulate/statistics
    e->last_f16 = f16;
    if (f16 && !e->ops[0].first_direct) return fail(TS2D_ERR_INVALID, "fp16 mode needs <= 4 input channels");
    if (!e->ops[0].first_direct) {   // boundary layout change NCHW -> NHWC (channels zero-padded to 8); > 4 input channels only
        const long long total = (long long)B * H * W;
        TRY(prof_begin(e, "input.nhwc", st));
        prof_kernel(e, "nchw_to_nhwc_pad");
        hipLaunchKernelGGL(nchw_to_nhwc_pad, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           d_in, a.input_channels, H * W, total, e->cin_pad, e->tensors[0].data);
        HIP_TRY(hipGetLastError());
        TRY(prof_end(e, st));
    }
    const float* wts = e->d_weights;
    const std::vector<Choice>& plan = planned(e, B, H, W);
    // a transposed conv that runs on its own although the workspace's plan (made for the reserved batch) composed it away: its output goes to the
    // scratch region (small batches, composed_kernel; one such tensor is alive at a time - it is read by the very next op only)
    for (size_t oi = 0; oi < e->ops.size(); ++oi) {
        const Op& op = e->ops[oi];
        if (op.type != OP_CONVT || plan[oi].k == K_FUSED_AWAY) continue;
        Tensor& t = e->tensors[op.dst];
        if (t.data != nullptr && t.data != e->d_up) continue;
        if (!e->d_up || (size_t)B * (H >> t.ly) * (W >> t.lx) * t.C > kSbkElems / 2)
            return fail(TS2D_ERR_STATE, "internal: op %s has no output buffer in the workspace plan", op.name.c_str());
        t.data = e->d_up; t.resident = false;
    }
    for (size_t oi = 0; oi < e->ops.size(); ++oi) {
        const Op& op = e->ops[oi];
        const Tensor& src = e->tensors[op.src];
        const Choice& c = plan[oi];
        const TileGeom& g = c.g;
        unsigned long long* const prof = (e->dbg == 256 && e->d_prof) ? e->d_prof + 512 * oi : nullptr;
        // per-tile partials -> scale / shift of the op's output (the launches of an op end with it)
        auto finalize = [&](int Ht, int Wt) -> int {
            Tensor& dst = e->tensors[op.dst];
            TRY(prof_begin(e, op.name + ".stats", st)); prof_kernel(e, "finalize_stats");
            launch_finalize(B, op.cout, st, e->d_part, g.tiles_x * g.tiles_y * c.ppt,
                            op.cout, B, Ht * Wt, wts + op.dev_g, wts + op.dev_be, a.norm_eps, dst.scale, dst.shift);
            HIP_TRY(hipGetLastError());
            return prof_end(e, st);
        };
        switch (c.k) {
        case K_NONE:
            return fail(TS2D_ERR_INVALID, "op %s has no fp16 kernel (channel counts must be multiples of 16)", op.name.c_str());
        case K_FUSED_AWAY:
            e->fused_away[oi] = 1;          // composed into the next block (kernels_upc.h): the upsampled tensor is never materialised
            break;
        case K_FIRST: case K_FIRST_STATS: {
            Tensor& dst = e->tensors[op.dst];
            FirstArgs fa{};
            fa.x = d_in; fa.w = wts + op.dev_wraw; fa.bias = wts + op.dev_b; fa.dst = dst.data;
            fa.part = c.fused_stats ? e->d_part : nullptr;
            fa.B = B; fa.C = op.cin; fa.H = H; fa.W = W; fa.Cout = op.cout;
            fa.lgTH = g.lgTH; fa.lgTW = g.lgTW; fa.lgNIMG = ilog2(g.NIMG); fa.tiles_x = g.tiles_x; fa.tiles_y = g.tiles_y;
            fa.n_mtiles = g.n_mtiles; fa.PH = g.PH; fa.PW = g.PW;
            const int kp = (op.cin + 1) / 2, nt = op.cout / 32, P = g.PH * g.PW * g.NIMG;
            size_t smem = std::max((size_t)P * (2 * kp + 1) * sizeof(float), (size_t)4 * op.cout * 4 * sizeof(float));
            if (f16 && c.first_full && nt == 1) {          // per-wave transpose regions of the 16-byte-store epilogue (kernels_first.h)
                fa.tr_off = (int)align_up(smem, 16);
                smem = fa.tr_off + 4 * kFirstTr;
            }
            // the K = 9 C contraction on the fp16 matrix path (hi / lo split input, one K = 54 product; kernels_first.h SPLIT): every mode but the exact one
            const bool fsplit = e->use_first_split && e->precision != TS2D_PRECISION_F32_EXACT && c.first_full && nt == 1 && kp == 1;
            if (fsplit) smem = (size_t)first_split_lds(P, f16 && c.k != K_FIRST_STATS);
            TRY(prof_begin(e, op.name, st)); prof_kernel(e, "conv3x3_first");
            // complete one-image 256-pixel tiles everywhere: persistent workgroups (4 per CU) with the next tile's patch in flight
            const bool first_full = c.first_full;
            const int grid_first = first_full ? std::min(g.n_mtiles, e->num_cus * (nt == 1 ? 4 : 2)) : g.n_mtiles;
            if (c.k == K_FIRST_STATS) {        // statistics only; the block is recomputed inside the second one (implies first_full, nt == 1, kp == 1)
                if (!first_full || !c.fused_stats) return fail(TS2D_ERR_INVALID, "internal: fused first block on a geometry without complete tiles");
                prof_kernel(e, "conv3x3_first_stats");
                if (fsplit) {
                    if (f16) hipLaunchKernelGGL((conv3x3_first_split<_Float16, false>), dim3(grid_first), dim3(kBlock), smem, st, fa);
                    else hipLaunchKernelGGL((conv3x3_first_split<float, false>), dim3(grid_first), dim3(kBlock), smem, st, fa);
                } else if (f16) hipLaunchKernelGGL((conv3x3_first<1, 1, _Float16, true, false>), dim3(grid_first), dim3(kBlock), smem, st, fa);
                else hipLaunchKernelGGL((conv3x3_first<1, 1, float, true, false>), dim3(grid_first), dim3(kBlock), smem, st, fa);
                e->fused_away[0] = 1;
            } else if (fsplit) {
                prof_kernel(e, "conv3x3_first_split");
                if (f16) hipLaunchKernelGGL((conv3x3_first_split<_Float16, true>), dim3(grid_first), dim3(kBlock), smem, st, fa);
                else hipLaunchKernelGGL((conv3x3_first_split<float, true>), dim3(grid_first), dim3(kBlock), smem, st, fa);
            }
#define TS2D_FIRST(NT_, KP_) do { \
                if (first_full) { if (f16) hipLaunchKernelGGL((conv3x3_first<NT_, KP_, _Float16, true>), dim3(grid_first), dim3(kBlock), smem, st, fa); \
                                  else hipLaunchKernelGGL((conv3x3_first<NT_, KP_, float, true>), dim3(grid_first), dim3(kBlock), smem, st, fa); } \
                else { if (f16) hipLaunchKernelGGL((conv3x3_first<NT_, KP_, _Float16, false>), dim3(grid_first), dim3(kBlock), smem, st, fa); \
                       else hipLaunchKernelGGL((conv3x3_first<NT_, KP_, float, false>), dim3(grid_first), dim3(kBlock), smem, st, fa); } } while (0)
            else if (nt == 1 && kp == 1) TS2D_FIRST(1, 1);
            else if (nt == 1 && kp == 2) TS2D_FIRST(1, 2);
            else if (nt == 2 && kp == 1) TS2D_FIRST(2, 1);
            else if (nt == 2 && kp == 2) TS2D_FIRST(2, 2);
            else return fail(TS2D_ERR_INVALID, "first block: unsupported Cout %d / Cin %d", op.cout, op.cin);
#undef TS2D_FIRST
            HIP_TRY(hipGetLastError());
            TRY(prof_end(e, st));
            if (c.fused_stats) TRY(finalize(H, W));
            else {
                TRY(prof_begin(e, op.name + ".stats", st)); prof_kernel(e, "finalize_stats");
                launch_stats_direct(f16, B, op.cout, H * W, dst.data, wts + op.dev_g, wts + op.dev_be, a.norm_eps, dst.scale, dst.shift, st);
                HIP_TRY(hipGetLastError());
                TRY(prof_end(e, st));
            }
            break;
        }
        case K_UP0: case K_UPQ: case K_UPC: case K_UPC_H: case K_UPC_H2: {
            const Op& up = e->ops[op.up_idx];
            const Tensor& xc = e->tensors[up.src]; const Tensor& sk = e->tensors[op.skip]; Tensor& dst = e->tensors[op.dst];
            const int Ht = H >> op.ly, Wt = W >> op.lx, bn = c.bn;
            UpcArgs ua{};
            ua.xc = xc.data; ua.scc = xc.scale; ua.shc = xc.shift; ua.Cb = up.cin;
            ua.xs = sk.data; ua.scs = sk.scale; ua.shs = sk.shift; ua.Cs = op.cin_skip;
            ua.wc = wts + op.dev_wc; ua.wk = wts + op.dev_wk; ua.bvar = wts + op.dev_bvar; ua.oscale = wts + op.dev_wcs;
            ua.dst = dst.data; ua.part = e->d_part;
            ua.B = B; ua.H = Ht; ua.W = Wt; ua.Cout = op.cout;
            ua.tiles_x = g.tiles_x; ua.tiles_y = g.tiles_y; ua.n_mtiles = g.n_mtiles; ua.n_ctiles = op.cout / bn;
            ua.TH = g.TH; ua.TW = g.TW; ua.inv_twc = 1.0f / (float)(g.TW / 2);
            const bool flex = c.flex;                 // extent-following tile (the level is no multiple of 8 x 32): masked, division-addressed instance
            ua.slope = a.leaky_slope;
            ua.prof = prof; ua.dbg = e->dbg;
            const int grid = (ua.n_mtiles + 7) / 8 * 8 * ua.n_ctiles;
            TRY(prof_begin(e, op.name, st));
            if (c.k == K_UP0) {       // level 0: persistent, resident skip weights, 16x16x32 transposed product (kernels_up0.h)
                Up0Args u0{};
                u0.xc = xc.data; u0.scc = xc.scale; u0.shc = xc.shift; u0.xs = sk.data; u0.scs = sk.scale; u0.shs = sk.shift;
                u0.wc0 = wts + op.dev_w0c; u0.wk0 = wts + op.dev_w0k; u0.bvar = wts + op.dev_bvar; u0.oscale = wts + op.dev_wcs;
                u0.dst = dst.data; u0.part = e->d_part;
                u0.B = B; u0.H = Ht; u0.W = Wt; u0.tiles_x = g.tiles_x; u0.tiles_y = g.tiles_y; u0.n_tiles = g.n_mtiles;
                u0.slope = a.leaky_slope;
                u0.prof = prof;
                const int tpi0 = u0.tiles_x * u0.tiles_y, want = std::min(u0.n_tiles, 2 * e->num_cus);
                int seg = 1;          // (segments as conv3x3_res32: the largest divisor of an image's tiles that leaves >= 2 workgroups per CU)
                for (int d = 1; d <= tpi0; ++d) if (tpi0 % d == 0 && u0.n_tiles / d >= want) seg = d;
                if (e->u0seg > 0 && tpi0 % e->u0seg == 0) seg = e->u0seg;      // (option "u0seg": tests)
                u0.seg = seg;
                prof_kernel(e, "conv3x3_up0");
                if (f16) {
                    static std::atomic<uint64_t> done0h{0};
                    HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(conv3x3_up0<_Float16, 1>), done0h));
                    hipLaunchKernelGGL((conv3x3_up0<_Float16, 1>), dim3(u0.n_tiles / seg), dim3(kBlock), 9 * 4 * 512 + 8 * kResPS, st, u0);
                } else {
                    static std::atomic<uint64_t> done0{0};
                    HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(conv3x3_up0<float, 3>), done0));
                    hipLaunchKernelGGL((conv3x3_up0<float, 3>), dim3(u0.n_tiles / seg), dim3(kBlock), 9 * 2 * 4 * 512 + 8 * kResPS, st, u0);
                }
            } else if (c.k == K_UPC_H2) {
                // 16-bit mode on 16 x 32 tiles: four M tiles per wave, skip-half weights by DMA (kernels_upc_h2.h)
                const int ks = up.cin % 64 == 0 ? 4 : 2;
                prof_kernel(e, "conv3x3_upc_h2");
                static std::atomic<uint64_t> doneh4{0}, doneh2{0};
                if (ks == 4) { HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(conv3x3_upc_h2<4>), doneh4));
                               hipLaunchKernelGGL(conv3x3_upc_h2<4>, dim3(grid), dim3(kBlock), kUh2Lds, st, ua); }
                else { HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(conv3x3_upc_h2<2>), doneh2));
                       hipLaunchKernelGGL(conv3x3_upc_h2<2>, dim3(grid), dim3(kBlock), kUh2Lds, st, ua); }
            } else if (c.k == K_UPC_H) {       // 16-bit mode: fp16 storage, one product (kernels_upc_h.h)
                const int ks = up.cin % 64 == 0 ? 4 : 2;
                const size_t smem_h = std::max((size_t)ks * 2 * kUcPlane, (size_t)4 * kUsPlane + (size_t)2 * 9 * 2 * bn * 16);
                prof_kernel(e, bn == 64 ? "conv3x3_upc_h<64>" : "conv3x3_upc_h<32>");
#define TS2D_UPCH(BN_, KS_) do { static std::atomic<uint64_t> done_{0}, donef_{0}; \
                    if (flex) { HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(conv3x3_upc_h<BN_, KS_, true>), donef_)); \
                                hipLaunchKernelGGL((conv3x3_upc_h<BN_, KS_, true>), dim3(grid), dim3(kBlock), smem_h, st, ua); } \
                    else { HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(conv3x3_upc_h<BN_, KS_>), done_)); \
                           hipLaunchKernelGGL((conv3x3_upc_h<BN_, KS_>), dim3(grid), dim3(kBlock), smem_h, st, ua); } } while (0)
                if (bn == 64) { if (ks == 4) TS2D_UPCH(64, 4); else TS2D_UPCH(64, 2); }
                else { if (ks == 4) TS2D_UPCH(32, 4); else TS2D_UPCH(32, 2); }
#undef TS2D_UPCH
            } else if (c.k == K_UPQ) {       // 16 x 32 tiles, one 512-thread workgroup per CU, double-buffered staging (kernels_upq.h)
                prof_kernel(e, "conv3x3_upq");
                static std::atomic<uint64_t> doneq{0};
                HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(conv3x3_upq), doneq));
                hipLaunchKernelGGL(conv3x3_upq, dim3(grid), dim3(kUqThreads), kUqLds, st, ua);
            } else {
                const size_t smem_u = std::max((size_t)8 * kUcPlane, (size_t)4 * kUsPlane + (size_t)9 * 4 * bn * 16);
                prof_kernel(e, bn == 64 ? "conv3x3_upc<64>" : "conv3x3_upc<32>");
#define TS2D_UPC(BN_, FLEX_) do { static std::atomic<uint64_t> done_{0}; \
                    HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(conv3x3_upc<BN_, FLEX_>), done_)); \
                    hipLaunchKernelGGL((conv3x3_upc<BN_, FLEX_>), dim3(grid), dim3(kBlock), smem_u, st, ua); } while (0)
                if (bn == 64) { if (flex) TS2D_UPC(64, true); else TS2D_UPC(64, false); }
                else { if (flex) TS2D_UPC(32, true); else TS2D_UPC(32, false); }
#undef TS2D_UPC
            }
            HIP_TRY(hipGetLastError());
            TRY(prof_end(e, st));
            TRY(finalize(Ht, Wt));
            break;
        }
        case K_S1_RES32: case K_S1_RES32F: {
            Tensor& dst = e->tensors[op.dst];
            const int Ht = H >> op.ly, Wt = W >> op.lx;
            Res32Args ra{};
            ra.src = src.data; ra.sc = src.scale; ra.sh = src.shift; ra.wres = wts + op.dev_wres; ra.bias = wts + op.dev_b;
            ra.oscale = wts + op.dev_ws; ra.dst = dst.data; ra.part = e->d_part;
            ra.B = B; ra.H = Ht; ra.W = Wt; ra.tiles_x = g.tiles_x; ra.tiles_y = g.tiles_y; ra.n_tiles = g.n_mtiles;
            ra.slope = a.leaky_slope;
            ra.prof = prof;
            // segment = the largest divisor of the tiles of one image that still leaves >= 2 workgroups per CU (or all tiles)
            const int tpi_r = ra.tiles_x * ra.tiles_y, want = std::min(ra.n_tiles, 2 * e->num_cus);
            int seg = 1;
            for (int d = 1; d <= tpi_r; ++d) if (tpi_r % d == 0 && ra.n_tiles / d >= want) seg = d;
            ra.seg = seg;
            const int nbk = ra.n_tiles / seg;
            const bool fuse0 = c.k == K_S1_RES32F;       // the first block was a statistics-only pass: recompute it here
            if (fuse0) {
                if (!e->fused_away[0]) return fail(TS2D_ERR_INVALID, "internal: fused second block without the statistics-only first pass");
                const Op& o0 = e->ops[0];
                ra.src = nullptr; ra.x0 = d_in; ra.C0 = o0.cin; ra.w0 = wts + o0.dev_wraw; ra.b0 = wts + o0.dev_b;
            }
            TRY(prof_begin(e, op.name, st)); prof_kernel(e, fuse0 ? "conv3x3_res32f" : "conv3x3_res32");
#define TS2D_RES32(ST_, NP_, FUSE_, LDS_) do { static std::atomic<uint64_t> done_{0}; \
                HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(conv3x3_res32<ST_, NP_, FUSE_>), done_)); \
                hipLaunchKernelGGL((conv3x3_res32<ST_, NP_, FUSE_>), dim3(nbk), dim3(kBlock), LDS_, st, ra); } while (0)
            if (f16) { if (fuse0) TS2D_RES32(_Float16, 1, true, 9 * 4 * 512 + 4 * kResPS + 1536); else TS2D_RES32(_Float16, 1, false, 9 * 4 * 512 + 4 * kResPS + 1536); }
            else { if (fuse0) TS2D_RES32(float, 3, true, 9 * 2 * 4 * 512 + 8 * kResPS); else TS2D_RES32(float, 3, false, 9 * 2 * 4 * 512 + 8 * kResPS); }
#undef TS2D_RES32
            HIP_TRY(hipGetLastError());
            TRY(prof_end(e, st));
            TRY(finalize(Ht, Wt));
            break;
        }
        case K_HEAD_MFMA: case K_HEAD_1X1: {
            HeadArgs ha{};
            ha.src = src.data; ha.sc = src.scale; ha.sh = src.shift; ha.w = wts + op.dev_w; ha.bias = wts + op.dev_b;
            ha.logits = d_logits; ha.mask = d_mask; ha.C = src.C; ha.K = op.cout; ha.HW = H * W;
            ha.total = (long long)B * H * W; ha.slope = a.leaky_slope; ha.nonfinite = e->d_flags;
            const unsigned grid = (unsigned)((ha.total + 255) / 256);
            const size_t smem = ((size_t)256 * (src.C + 1) + (size_t)op.cout * src.C + op.cout) * sizeof(float);
            TRY(prof_begin(e, op.name, st));
            prof_kernel(e, "head");
            if (c.k == K_HEAD_MFMA) {      // matrix-core head (split / f16 modes)
                ha.wph = wts + op.dev_wh; ha.oscale = wts + op.dev_ws;
                const int bpw = 16;          // (4 ... 128 measured: 16 is as good as any - fewer, longer waves lose DRAM locality)
                const long long nblk = ha.total / 32;
                const unsigned gridm = (unsigned)((nblk + 4 * bpw - 1) / (4 * bpw));
                if (f16) hipLaunchKernelGGL((head_mfma32<_Float16, 1>), dim3(gridm), dim3(256), 0, st, ha, bpw);
                else hipLaunchKernelGGL((head_mfma32<float, 3>), dim3(gridm), dim3(256), 0, st, ha, bpw);
            } else if (src.C == 32) {
                static std::atomic<uint64_t> set32a{0}, set32b{0};
                HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(head_1x1<32, float>), set32a));
                HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(head_1x1<32, _Float16>), set32b));
                if (f16) hipLaunchKernelGGL((head_1x1<32, _Float16>), dim3(grid), dim3(256), smem, st, ha);
                else hipLaunchKernelGGL((head_1x1<32, float>), dim3(grid), dim3(256), smem, st, ha);
            } else {
                static std::atomic<uint64_t> set64a{0}, set64b{0};
                HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(head_1x1<64, float>), set64a));
                HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(head_1x1<64, _Float16>), set64b));
                if (f16) hipLaunchKernelGGL((head_1x1<64, _Float16>), dim3(grid), dim3(256), smem, st, ha);
                else hipLaunchKernelGGL((head_1x1<64, float>), dim3(grid), dim3(256), smem, st, ha);
            }
            HIP_TRY(hipGetLastError());
            TRY(prof_end(e, st));
            break;
        }
        default: {      // every kernel that takes ConvArgs
            const bool conv = op.type == OP_CONV;
            Tensor& dst = e->tensors[op.dst];
            const int Hin = H >> src.ly, Win = W >> src.lx;
            const int Ht = conv ? (H >> op.ly) : Hin, Wt = conv ? (W >> op.lx) : Win;
            const int taps = conv ? 9 : 1, bn = c.bn;
            ConvArgs ca{};
            ca.ksplit = 1; ca.dbg = e->dbg; ca.prof = prof;
            ca.src0 = src.data; ca.sc0 = src.scale; ca.sh0 = src.shift; ca.C0 = src.C;
            if (op.skip >= 0) { const Tensor& sk = e->tensors[op.skip]; ca.src1 = sk.data; ca.sc1 = sk.scale; ca.sh1 = sk.shift; ca.C1 = sk.C; }
            ca.wp = wts + op.dev_w; ca.bias = wts + op.dev_b; ca.dst = dst.data;
            ca.part = (conv && c.fused_stats) ? e->d_part : nullptr;
            ca.B = B; ca.Hin = Hin; ca.Win = Win; ca.Ht = Ht; ca.Wt = Wt;
            ca.KA = conv ? 1 : op.sy; ca.KB = conv ? 1 : op.sx;
            ca.N = conv ? op.cout : op.sy * op.sx * op.cout; ca.Cout = op.cout;
            set_tiling(ca, g, ca.N / bn);
            ca.slope = a.leaky_slope;
            const int P = g.PH * g.PW * g.NIMG;
            const int grid = (g.n_mtiles + 7) / 8 * 8 * ca.n_ctiles;
            hipError_t le = hipSuccess;
            TRY(prof_begin(e, op.name, st));
            switch (c.k) {
            case K_S2_V2: {
                ca.wph = wts + op.dev_w2; ca.oscale = wts + op.dev_ws;
                // 16-bit mode: chunks of 32 channels (the second part of the LDS images = channels 16-31 instead of the split mode's lo parts; kernels_s2v2.h K32)
                const bool k32 = f16 && e->use_s2k32 && op.cin % 32 == 0;
                const int npp = (f16 && !k32) ? 1 : 2, nch = op.cin / (k32 ? 32 : 16);
                // persistent: one workgroup per CU walks its tiles; every chunk's weights resident in LDS when they fit beside the patch
                const size_t wchunk = (size_t)9 * npp * 2 * op.bn2 * 16;
                const bool resw = (size_t)npp * 2 * kS2Plane + (size_t)nch * wchunk <= (size_t)160 * 1024;
                const int grid2 = std::min(grid, 8 * ca.n_ctiles * std::max(1, e->num_cus / (8 * ca.n_ctiles)));
                const size_t smem2 = (size_t)npp * 2 * kS2Plane + (resw ? (size_t)nch : 1) * wchunk;
                prof_kernel(e, op.bn2 == 128 ? (k32 ? "conv3x3s2_v2<128,k32>" : "conv3x3s2_v2<128>") : (k32 ? "conv3x3s2_v2<64,k32>" : "conv3x3s2_v2<64>"));
#define TS2D_S2V2_INST(BN_, ST_, NP_, RW_, FX_, K_) do { static std::atomic<uint64_t> done_{0}; \
                    HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(conv3x3s2_v2<BN_, ST_, NP_, RW_, FX_, K_>), done_)); \
                    hipLaunchKernelGGL((conv3x3s2_v2<BN_, ST_, NP_, RW_, FX_, K_>), dim3(grid2), dim3(kS2Threads), smem2, st, ca); } while (0)
#define TS2D_S2V2_LAUNCH(BN_, ST_, NP_, K_) do { \
                    if (resw) { if (c.flex) TS2D_S2V2_INST(BN_, ST_, NP_, true, true, K_); else TS2D_S2V2_INST(BN_, ST_, NP_, true, false, K_); } \
                    else { if (c.flex) TS2D_S2V2_INST(BN_, ST_, NP_, false, true, K_); else TS2D_S2V2_INST(BN_, ST_, NP_, false, false, K_); } } while (0)
                if (op.bn2 == 128) { if (k32) TS2D_S2V2_LAUNCH(128, _Float16, 1, true); else if (f16) TS2D_S2V2_LAUNCH(128, _Float16, 1, false); else TS2D_S2V2_LAUNCH(128, float, 3, false); }
                else { if (k32) TS2D_S2V2_LAUNCH(64, _Float16, 1, true); else if (f16) TS2D_S2V2_LAUNCH(64, _Float16, 1, false); else TS2D_S2V2_LAUNCH(64, float, 3, false); }
#undef TS2D_S2V2_LAUNCH
#undef TS2D_S2V2_INST
                le = hipGetLastError();
                break;
            }
            case K_T_ONE: case K_T_GENERIC: {
                ca.wph = wts + op.dev_wh; ca.oscale = wts + op.dev_ws;
                const int Pt = g.TH * g.TW * g.NIMG;
                const size_t smem_t = (size_t)2 * Pt * kRec + (size_t)2 * 64 * kRec;
                prof_kernel(e, "convT2x2_f16x3");
                if (c.k == K_T_ONE) {
                    if (f16) hipLaunchKernelGGL((convT2x2_f16x3_one<_Float16, 1>), dim3(grid), dim3(kBlock), smem_t, st, ca);
                    else hipLaunchKernelGGL((convT2x2_f16x3_one<float, 3>), dim3(grid), dim3(kBlock), smem_t, st, ca);
                } else if (f16) hipLaunchKernelGGL((convT2x2_f16x3<64, _Float16, 1>), dim3(grid), dim3(kBlock), smem_t, st, ca);
                else hipLaunchKernelGGL((convT2x2_f16x3<64, float, 3>), dim3(grid), dim3(kBlock), smem_t, st, ca);
                le = hipGetLastError();
                break;
            }
            case K_S1_H2: {
                // 16-bit mode, plain C -> C block on 16 x 32 tiles: the skip phase of conv3x3_upc_h2 (four M tiles per wave, weights by LDS-DMA)
                UpcArgs ua{};
                ua.xs = src.data; ua.scs = src.scale; ua.shs = src.shift; ua.Cs = src.C;
                ua.wk = wts + op.dev_wp; ua.bvar = wts + op.dev_b; ua.oscale = wts + op.dev_ws;
                ua.dst = dst.data; ua.part = e->d_part;
                ua.B = B; ua.H = Ht; ua.W = Wt; ua.Cout = op.cout;
                ua.tiles_x = g.tiles_x; ua.tiles_y = g.tiles_y; ua.n_mtiles = g.n_mtiles; ua.n_ctiles = op.cout / 64;
                ua.slope = a.leaky_slope; ua.dbg = e->dbg;
                static std::atomic<uint64_t> doneh2p{0};
                HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(conv3x3_upc_h2<4, false>), doneh2p));
                hipLaunchKernelGGL((conv3x3_upc_h2<4, false>), dim3((ua.n_mtiles + 7) / 8 * 8 * ua.n_ctiles), dim3(kBlock), kUh2Lds, st, ua);
                le = hipGetLastError(); prof_kernel(e, "conv3x3_h2");
                break;
            }
            case K_S1_QP: {
                // complete 16 x 32 tiles x 64 columns, normalised sources: one 512-thread workgroup per CU, patch and weights
                // double-buffered (weights by LDS-DMA), one barrier per chunk
                ca.wph = wts + op.dev_wp; ca.oscale = wts + op.dev_ws;
                const int gridp = std::min(grid, 8 * ca.n_ctiles * std::max(1, e->num_cus / (8 * ca.n_ctiles)));     // one persistent workgroup per CU
                static std::atomic<uint64_t> doneqp{0};
                HIP_TRY(allow_max_lds(reinterpret_cast<const void*>(conv3x3_f16x3_qp), doneqp));
                hipLaunchKernelGGL(conv3x3_f16x3_qp, dim3(gridp), dim3(kQThreads), kQpLds, st, ca);
                le = hipGetLastError(); prof_kernel(e, "conv3x3_f16x3_qp");
                break;
            }
            default: {
                const bool s2 = op.stride == 2;
                size_t smem = std::max((size_t)(((P * (op.ck + 4) + 3) & ~3) + taps * (op.ck / 8) * bn * 8) * sizeof(float),
                                       (size_t)4 * bn * 4 * sizeof(float));
                if (c.k != K_EXACT) {
                    smem = !s2 ? (size_t)P * kRec + (size_t)9 * bn * kRec : (size_t)P * kRec8 + (size_t)5 * bn * kRec;
                    smem = std::max(smem, (size_t)4 * bn * 4 * sizeof(float));
                    ca.wph = wts + op.dev_wh; ca.oscale = wts + op.dev_ws;
                }
                if (c.ksplit > 1) {
                    ca.ksplit = c.ksplit; ca.kslice_stride = (long long)B * Ht * Wt * op.cout;
                    ca.dst = e->d_partial; ca.part = nullptr;
                }
                if (smem > 160 * 1024) return fail(TS2D_ERR_INVALID, "op %s: LDS tile of %zu bytes exceeds 160 KiB", op.name.c_str(), smem);
                if (c.k == K_S1_ONE) {     // tile inside one image: lean staging path
                    le = launch_one(bn, ca, grid, smem, st); prof_kernel(e, bn == 64 ? "conv3x3_f16x3_one<64>" : "conv3x3_f16x3_one<32>");
                } else if (c.k == K_S2_ONE) {
                    le = launch_one_s2(f16, bn, ca, grid, smem, st); prof_kernel(e, "conv3x3s2_f16x3_one");
                } else if (c.k == K_S1_H32) {     // fp16 storage: 32-channel chunks, one product
                    ca.wph = wts + op.dev_wh32;
                    le = launch_h32(bn, ca, grid, smem, st); prof_kernel(e, bn == 64 ? "conv3x3_h32<64>" : "conv3x3_h32<32>");
                } else if (c.k == K_S1_GENERIC) {
                    le = launch_split(f16, bn, P * 2 <= 3 * kBlock ? 3 : 5, ca, grid, smem, st); prof_kernel(e, "conv3x3_f16x3");
                } else if (c.k == K_S2_GENERIC) {
                    le = launch_split_s2(f16, bn, P <= 5 * kBlock ? 5 : 6, ca, grid, smem, st); prof_kernel(e, "conv3x3s2_f16x3");
                } else {      // K_EXACT: the whole exact mode; in every mode the stages whose stride is neither (1, 1) nor (2, 2)
                    const bool aniso = conv ? op.stride == 3 : op.stride != 2;
                    le = launch_conv(taps, conv ? op.sy : 1, conv ? op.sx : 1, op.ck, bn, f16 && aniso, ca, grid, smem, st);
                    prof_kernel(e, conv ? "conv_mfma_f32" : "convT_mfma_f32");
                }
                break;
            }
            }
            if (le != hipSuccess) return fail(TS2D_ERR_HIP, "launch of %s failed: %s", op.name.c_str(), hipGetErrorString(le));
            TRY(prof_end(e, st));
            if (conv) {
                const int HW = Ht * Wt;
                if (c.ksplit > 1) {
                    TRY(prof_begin(e, op.name + ".stats", st)); prof_kernel(e, "finalize_stats");
                    if (HW == 256) {            // one block of 32 pixel lanes per (image, 32 channels): output + scale / shift in one launch
                        if (f16) hipLaunchKernelGGL((splitk_reduce_stats<_Float16, 32>), dim3(B, op.cout / 32), dim3(1024), 0, st, e->d_partial, c.ksplit, ca.kslice_stride,
                                                    wts + op.dev_b, op.cout, HW, wts + op.dev_g, wts + op.dev_be, a.norm_eps, reinterpret_cast<_Float16*>(dst.data), dst.scale, dst.shift);
                        else hipLaunchKernelGGL((splitk_reduce_stats<float, 32>), dim3(B, op.cout / 32), dim3(1024), 0, st, e->d_partial, c.ksplit, ca.kslice_stride,
                                                wts + op.dev_b, op.cout, HW, wts + op.dev_g, wts + op.dev_be, a.norm_eps, dst.data, dst.scale, dst.shift);
                    } else if (HW % 256 == 0 && HW > 256) {
                        // images of >= 256 pixels (the small-batch rule): one block per 256 pixels writes the output and a shifted partial, finalize_stats
                        // combines them - the one-block-per-image kernel below would walk the image 8 pixels per memory round trip
                        const int nblk = HW / 256;
                        if (f16) hipLaunchKernelGGL(splitk_reduce_part<_Float16>, dim3(nblk, op.cout / 32, B), dim3(256), 0, st, e->d_partial, c.ksplit, ca.kslice_stride,
                                                    wts + op.dev_b, op.cout, HW, reinterpret_cast<_Float16*>(dst.data), e->d_part);
                        else hipLaunchKernelGGL(splitk_reduce_part<float>, dim3(nblk, op.cout / 32, B), dim3(256), 0, st, e->d_partial, c.ksplit, ca.kslice_stride,
                                                wts + op.dev_b, op.cout, HW, dst.data, e->d_part);
                        launch_finalize(B, op.cout, st, e->d_part, nblk, op.cout, B, HW, wts + op.dev_g, wts + op.dev_be, a.norm_eps, dst.scale, dst.shift);
                    } else if (f16) hipLaunchKernelGGL(splitk_reduce_stats<_Float16>, dim3(B, op.cout / 32), dim3(256), 0, st, e->d_partial, c.ksplit, ca.kslice_stride,
                                                wts + op.dev_b, op.cout, HW, wts + op.dev_g, wts + op.dev_be, a.norm_eps, reinterpret_cast<_Float16*>(dst.data), dst.scale, dst.shift);
                    else hipLaunchKernelGGL(splitk_reduce_stats<float>, dim3(B, op.cout / 32), dim3(256), 0, st, e->d_partial, c.ksplit, ca.kslice_stride,
                                            wts + op.dev_b, op.cout, HW, wts + op.dev_g, wts + op.dev_be, a.norm_eps, dst.data, dst.scale, dst.shift);
                    HIP_TRY(hipGetLastError());
                    TRY(prof_end(e, st));
                } else if (c.fused_stats) {
                    TRY(finalize(Ht, Wt));
                } else {
                    TRY(prof_begin(e, op.name + ".stats", st)); prof_kernel(e, "finalize_stats");
                    launch_stats_direct(f16, B, op.cout, HW, dst.data, wts + op.dev_g, wts + op.dev_be, a.norm_eps, dst.scale, dst.shift, st);
                    HIP_TRY(hipGetLastError());
                    TRY(prof_end(e, st));
                }
            }
            break;
        }
        }
    }
    return TS2D_OK;
}


}  // namespace

// ------------------------------------------------------------------------------------------------- C-ABI
extern "C" {

int ts2d_abi_version(void) { return 7; }

const char* ts2d_last_error(void) { return g_err.c_str(); }

int ts2d_engine_create(const ts2d_arch_desc* arch, const float* weights, size_t n_floats, int device, ts2d_engine** out) {
    if (!arch || !out) return fail(TS2D_ERR_INVALID, "ts2d_engine_create: null argument");
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(TS2D_ERR_INVALID, "device %d out of range (%d HIP devices visible)", device, ndev);
    ts2d_engine* e = new (std::nothrow) ts2d_engine();
    if (!e) return fail(TS2D_ERR_NOMEM, "host allocation failed");
    e->user_arch = *arch; e->device = device;
    if (arch->n_stages >= 2 && arch->n_stages <= TS2D_MAX_STAGES) e->arch = pad_arch(*arch, e->padded); else e->arch = *arch;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) e->num_cus = prop.multiProcessorCount;
        if (getenv("TS2D_DBG")) e->dbg = atoi(getenv("TS2D_DBG"));
    }
    int rc = build_program(e);
    if (rc != TS2D_OK) { delete e; return rc; }
    hipError_t he = hipSetDevice(device);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ws_event, hipEventDisableTiming);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&e->d_flags), 2 * sizeof(int));
    if (he == hipSuccess) he = hipMemset(e->d_flags, 0, 2 * sizeof(int));
    if (he == hipSuccess && e->dbg == 256) he = hipMalloc(reinterpret_cast<void**>(&e->d_prof), 512 * 128 * sizeof(unsigned long long));
    if (he == hipSuccess && e->d_prof) he = hipMemset(e->d_prof, 0, 512 * 128 * sizeof(unsigned long long));
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&e->d_weights), e->weight_floats * sizeof(float));
    if (he != hipSuccess) {
        rc = fail(he == hipErrorOutOfMemory ? TS2D_ERR_NOMEM : TS2D_ERR_HIP, "engine setup failed: %s", hipGetErrorString(he));
        ts2d_engine_destroy(e);
        return rc;
    }
    if (weights) {
        rc = upload_weights(e, weights, n_floats);
        if (rc != TS2D_OK) { ts2d_engine_destroy(e); return rc; }
    }
    *out = e;
    return TS2D_OK;
}

int ts2d_engine_load_weights(ts2d_engine* e, const float* weights, size_t n_floats) {
    if (!e || !weights) return fail(TS2D_ERR_INVALID, "ts2d_engine_load_weights: null argument");
    HIP_TRY(hipSetDevice(e->device));
    if (e->ws_busy) { HIP_TRY(hipEventSynchronize(e->ws_event)); e->ws_busy = false; }     // a run on any stream may still read the old weights
    HIP_TRY(hipStreamSynchronize(e->stream));
    return upload_weights(e, weights, n_floats);
}

int ts2d_engine_weight_buffer(ts2d_engine* e, void** dev_ptr, size_t* n_bytes) {
    if (!e || !dev_ptr || !n_bytes) return fail(TS2D_ERR_INVALID, "ts2d_engine_weight_buffer: null argument");
    *dev_ptr = e->d_weights; *n_bytes = e->weight_floats * sizeof(float);
    return TS2D_OK;
}

int ts2d_engine_set_precision(ts2d_engine* e, int mode) {
    if (!e) return fail(TS2D_ERR_INVALID, "ts2d_engine_set_precision: null engine");
    if (mode != TS2D_PRECISION_F32_EXACT && mode != TS2D_PRECISION_F32_SPLIT_F16X3 && mode != TS2D_PRECISION_F16)
        return fail(TS2D_ERR_INVALID, "unknown precision mode %d", mode);
    e->precision = mode;
    return TS2D_OK;
}

int ts2d_engine_set_option(ts2d_engine* e, const char* name, int value) {
    if (!e || !name) return fail(TS2D_ERR_INVALID, "ts2d_engine_set_option: null argument");
    struct B { const char* n; bool* p; };
    struct I { const char* n; int* p; int lo, hi; };
    const B bools[] = {{"h32", &e->use_h32}, {"one", &e->use_one}, {"s2v2", &e->use_s2v2}, {"q", &e->use_q}, {"h2", &e->use_h2},  {"uh2", &e->use_uh2},
                       {"first_split", &e->use_first_split}, {"sbk", &e->use_sbk}, {"s2k32", &e->use_s2k32}, {"up0", &e->use_up0}, {"upq", &e->use_upq}, {"upc", &e->use_upc}, {"res", &e->use_res}, {"fuse0", &e->use_fuse0}, {"flex", &e->use_flex}};
    const I ints[] = {{"upq_min", &e->upq_min, 0, 1 << 20}, {"h2_min", &e->h2_min, 0, 1 << 20}, {"u0seg", &e->u0seg, 0, 1 << 20}, {"flex2", &e->use_flex2, 0, 2}};
    bool found = false;
    for (const B& b : bools) if (!strcmp(name, b.n)) { *b.p = value != 0; found = true; }
    for (const I& i : ints) if (!strcmp(name, i.n)) {
        if (value < i.lo || value > i.hi) return fail(TS2D_ERR_INVALID, "option %s = %d out of range [%d, %d]", name, value, i.lo, i.hi);
        *i.p = value; found = true;
    }
    if (!found) return fail(TS2D_ERR_INVALID, "unknown option '%s' (h32 one s2v2 q h2 h2_min uh2 up0 u0seg upq upq_min upc res fuse0 flex flex2 first_split sbk s2k32)", name);
    e->ws_precision = -1;         // which ops compose (and with it the activation plan) depends on the options: re-plan at the next reserve / forward
    ++e->opt_gen;
    return TS2D_OK;
}

int ts2d_engine_set_keep_activations(ts2d_engine* e, int enable) {
    if (!e) return fail(TS2D_ERR_INVALID, "ts2d_engine_set_keep_activations: null engine");
    e->keep_activations = enable != 0;          // takes effect at the next reserve / forward (the workspace is re-planned)
    return TS2D_OK;
}

int ts2d_engine_set_tile_dtype(ts2d_engine* e, int mode) {
    if (!e) return fail(TS2D_ERR_INVALID, "ts2d_engine_set_tile_dtype: null engine");
    if (mode != TS2D_TILE_F32 && mode != TS2D_TILE_F16) return fail(TS2D_ERR_INVALID, "unknown tile dtype %d", mode);
    e->tile_half = mode == TS2D_TILE_F16;
    return TS2D_OK;
}

int ts2d_engine_weights_ready(ts2d_engine* e) {
    if (!e) return fail(TS2D_ERR_INVALID, "ts2d_engine_weights_ready: null engine");
    e->weights_ready = true;
    return TS2D_OK;
}

int ts2d_engine_reserve(ts2d_engine* e, int B, int H, int W) {
    if (!e) return fail(TS2D_ERR_INVALID, "ts2d_engine_reserve: null engine");
    const int divy = 1 << e->lvl_y[e->arch.n_stages - 1], divx = 1 << e->lvl_x[e->arch.n_stages - 1];
    if (B < 1 || H < divy || W < divx || H % divy || W % divx)
        return fail(TS2D_ERR_INVALID, "shape B=%d H=%d W=%d: H and W must be positive multiples of %d and %d", B, H, W, divy, divx);
    if ((H / divy) * (W / divx) <= 1)   // torch InstanceNorm2d raises "Expected more than 1 spatial element" here too
        return fail(TS2D_ERR_INVALID, "shape %dx%d leaves a single bottleneck pixel: InstanceNorm needs more than 1 spatial element", H, W);
    if ((long long)B * H * W >= (1LL << 31)) return fail(TS2D_ERR_INVALID, "B*H*W = %lld exceeds 2^31 pixels per call", (long long)B * H * W);
    return ensure_workspace(e, B, H, W);
}

int ts2d_engine_workspace_bytes(ts2d_engine* e, int B, int H, int W, size_t* n_bytes) {
    if (!e || !n_bytes) return fail(TS2D_ERR_INVALID, "ts2d_engine_workspace_bytes: null argument");
    const int divy = 1 << e->lvl_y[e->arch.n_stages - 1], divx = 1 << e->lvl_x[e->arch.n_stages - 1];
    if (B < 1 || H < divy || W < divx || H % divy || W % divx)
        return fail(TS2D_ERR_INVALID, "shape B=%d H=%d W=%d: H and W must be positive multiples of %d and %d", B, H, W, divy, divx);
    *n_bytes = workspace_layout(e, B, H, W).bytes;
    return TS2D_OK;
}

int ts2d_engine_set_workspace(ts2d_engine* e, void* dev_ptr, size_t n_bytes) {
    if (!e) return fail(TS2D_ERR_INVALID, "ts2d_engine_set_workspace: null engine");
    if (dev_ptr && (reinterpret_cast<uintptr_t>(dev_ptr) & 255)) return fail(TS2D_ERR_INVALID, "ts2d_engine_set_workspace: the pointer must be 256-byte aligned");
    HIP_TRY(hipSetDevice(e->device));
    if (e->ws_busy) { HIP_TRY(hipEventSynchronize(e->ws_event)); e->ws_busy = false; }      // a run may still use the old memory
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->d_ws && !e->ws_external) HIP_TRY(hipFree(e->d_ws));
    e->d_ws = reinterpret_cast<char*>(dev_ptr); e->ws_bytes = dev_ptr ? n_bytes : 0; e->ws_external = dev_ptr != nullptr;
    e->ws_precision = -1; e->wsB = 0;           // re-plan inside the new memory at the next reserve / forward
    for (Tensor& t : e->tensors) { t.data = nullptr; t.resident = false; }
    e->lastB = 0;
    return TS2D_OK;
}

int ts2d_engine_forward(ts2d_engine* e, const float* input, int B, int H, int W, float* logits, uint32_t* mask_packed,
                        int on_device, void* stream) {
    if (!e || !input) return fail(TS2D_ERR_INVALID, "ts2d_engine_forward: null argument");
    if (!e->weights_ready) return fail(TS2D_ERR_STATE, "ts2d_engine_forward: weights not loaded (create with weights, or broadcast + ts2d_engine_weights_ready)");
    if (!logits && !mask_packed) return fail(TS2D_ERR_INVALID, "ts2d_engine_forward: both outputs are null");
    if (mask_packed && (W % 32 || (H * W) % 64)) return fail(TS2D_ERR_INVALID, "packed mask output needs W %% 32 == 0 (W = %d)", W);
    TRY(ts2d_engine_reserve(e, B, H, W));
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : e->stream;
    const int K = e->arch.num_classes;
    if (on_device) return run_forward(e, input, B, H, W, logits, mask_packed, st);
    // host buffers: staged on the device, synchronous
    TRY(ensure_staging(e, B, H, W));
    HIP_TRY(hipMemcpyAsync(e->d_in_stage, input, (size_t)B * e->arch.input_channels * H * W * sizeof(float), hipMemcpyHostToDevice, st));
    TRY(run_forward(e, e->d_in_stage, B, H, W, logits ? e->d_logit_stage : nullptr, mask_packed ? e->d_mask_stage : nullptr, st));
    if (logits) HIP_TRY(hipMemcpyAsync(logits, e->d_logit_stage, (size_t)B * K * H * W * sizeof(float), hipMemcpyDeviceToHost, st));
    if (mask_packed) HIP_TRY(hipMemcpyAsync(mask_packed, e->d_mask_stage, (size_t)B * K * H * (W / 32) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return ts2d_engine_check(e);          // never a silent inf / NaN (asynchronous device-pointer calls: the caller runs the check)
}

int ts2d_engine_predict_tiled(ts2d_engine* e, const float* image, int Hp, int Wp, int ph, int pw, int n_tiles,
                              const int32_t* tile_y, const int32_t* tile_x, int mirror_mask, const uint16_t* gaussian_f16,
                              uint16_t* logits_f16, uint8_t* seg_u8) {
    if (!e || !image || !tile_y || !tile_x) return fail(TS2D_ERR_INVALID, "ts2d_engine_predict_tiled: null argument");
    if (!e->weights_ready) return fail(TS2D_ERR_STATE, "ts2d_engine_predict_tiled: weights not loaded");
    if (!logits_f16 && !seg_u8) return fail(TS2D_ERR_INVALID, "ts2d_engine_predict_tiled: both outputs are null");
    if (n_tiles < 1 || ph > Hp || pw > Wp) return fail(TS2D_ERR_INVALID, "bad tiling: %d tiles of %dx%d on %dx%d", n_tiles, ph, pw, Hp, Wp);
    for (int t = 0; t < n_tiles; ++t)
        if (tile_y[t] < 0 || tile_x[t] < 0 || tile_y[t] + ph > Hp || tile_x[t] + pw > Wp)
            return fail(TS2D_ERR_INVALID, "tile %d at (%d,%d) leaves the %dx%d image", t, tile_y[t], tile_x[t], Hp, Wp);
    const int C = e->arch.input_channels, K = e->arch.num_classes;
    int vflip[4] = {0, 0, 0, 0}, V = 1;
    if ((mirror_mask & 3) == 3) { vflip[1] = 1; vflip[2] = 2; vflip[3] = 3; V = 4; }
    else if (mirror_mask & 1) { vflip[1] = 1; V = 2; }
    else if (mirror_mask & 2) { vflip[1] = 2; V = 2; }
    const int rows = n_tiles * V, chunk = std::min(rows, 64);
    TRY(ts2d_engine_reserve(e, chunk, ph, pw));
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t st = e->stream;
    // scratch layout
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_img = take((size_t)C * Hp * Wp * 4), o_batch = take((size_t)rows * C * ph * pw * 4);
    const size_t o_log = take((size_t)rows * K * ph * pw * 4), o_g = take((size_t)ph * pw * 2);
    const size_t o_o16 = take((size_t)K * Hp * Wp * 2), o_seg = take((size_t)K * Hp * Wp);
    const size_t o_ty = take((size_t)n_tiles * 4), o_tx = take((size_t)n_tiles * 4), o_vf = take(16), o_flag = take(4);
    if (off > e->sw_bytes) {
        HIP_TRY(hipStreamSynchronize(st));
        if (e->d_sw) { HIP_TRY(hipFree(e->d_sw)); e->d_sw = nullptr; e->sw_bytes = 0; }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_sw), off));
        e->sw_bytes = off;
    }
    char* b = e->d_sw;
    float* d_img = reinterpret_cast<float*>(b + o_img); float* d_batch = reinterpret_cast<float*>(b + o_batch);
    float* d_log = reinterpret_cast<float*>(b + o_log); __half* d_g = reinterpret_cast<__half*>(b + o_g);
    __half* d_o16 = reinterpret_cast<__half*>(b + o_o16); uint8_t* d_seg = reinterpret_cast<uint8_t*>(b + o_seg);
    int* d_ty = reinterpret_cast<int*>(b + o_ty); int* d_tx = reinterpret_cast<int*>(b + o_tx); int* d_vf = reinterpret_cast<int*>(b + o_vf);
    HIP_TRY(hipMemcpyAsync(d_img, image, (size_t)C * Hp * Wp * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_ty, tile_y, (size_t)n_tiles * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_tx, tile_x, (size_t)n_tiles * 4, hipMemcpyHostToDevice, st));
    int* d_flag = reinterpret_cast<int*>(b + o_flag);
    HIP_TRY(hipMemsetAsync(d_flag, 0, 4, st));
    HIP_TRY(hipMemcpyAsync(d_vf, vflip, 16, hipMemcpyHostToDevice, st));
    if (gaussian_f16) HIP_TRY(hipMemcpyAsync(d_g, gaussian_f16, (size_t)ph * pw * 2, hipMemcpyHostToDevice, st));
    {
        const long long total = (long long)rows * C * ph * pw;
        hipLaunchKernelGGL(sw_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_img, C, Hp, Wp, ph, pw, V, d_ty, d_tx, d_vf, d_batch, total);
        HIP_TRY(hipGetLastError());
    }
    for (int r0 = 0; r0 < rows; r0 += chunk) {
        const int nb = std::min(chunk, rows - r0);
        TRY(run_forward(e, d_batch + (size_t)r0 * C * ph * pw, nb, ph, pw, d_log + (size_t)r0 * K * ph * pw, nullptr, st, r0 == 0));
    }
    {
        const long long total = (long long)K * Hp * Wp;
        hipLaunchKernelGGL(sw_aggregate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_log, K, Hp, Wp, ph, pw, n_tiles, V,
                           d_ty, d_tx, d_vf, gaussian_f16 ? d_g : nullptr, logits_f16 ? d_o16 : nullptr, seg_u8 ? d_seg : nullptr,
                           kSigmoidHalfThreshold, total, d_flag, e->tile_half);
        HIP_TRY(hipGetLastError());
    }
    e->tiled_inf = 0;
    HIP_TRY(hipMemcpyAsync(&e->tiled_inf, d_flag, 4, hipMemcpyDeviceToHost, st));
    if (logits_f16) HIP_TRY(hipMemcpyAsync(logits_f16, d_o16, (size_t)K * Hp * Wp * 2, hipMemcpyDeviceToHost, st));
    if (seg_u8) HIP_TRY(hipMemcpyAsync(seg_u8, d_seg, (size_t)K * Hp * Wp, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return ts2d_engine_check(e);
}

int ts2d_engine_check(ts2d_engine* e) {
    if (!e) return fail(TS2D_ERR_INVALID, "ts2d_engine_check: null engine");
    if (!e->lastB) return TS2D_OK;
    HIP_TRY(hipSetDevice(e->device));
    if (e->ws_busy) HIP_TRY(hipEventSynchronize(e->ws_event));
    int flags[2] = {0, 0};
    HIP_TRY(hipMemcpy(flags, e->d_flags, sizeof(flags), hipMemcpyDeviceToHost));
    if (!flags[0]) return TS2D_OK;
    // Diagnosis (slow path, the activations of the run are still resident): the first tensor in program order that holds inf / NaN.
    auto has_nonfinite = [&](const void* p, size_t n, bool half) -> int {
        if (hipMemset(e->d_flags + 1, 0, sizeof(int)) != hipSuccess) return -1;
        const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
        if (half) hipLaunchKernelGGL(scan_nonfinite<_Float16>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const _Float16*>(p), n, e->d_flags + 1);
        else hipLaunchKernelGGL(scan_nonfinite<float>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const float*>(p), n, e->d_flags + 1);
        int f = 0;
        if (hipMemcpy(&f, e->d_flags + 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        return f;
    };
    const int B = e->lastB, H = e->lastH, W = e->lastW;
    // (only the LAST batch of the call is still resident: after a multi-chunk ts2d_engine_predict_tiled an earlier chunk's inf / NaN
    //  is reported, but located only if the last chunk shows it too)
    std::string where = "the head (or an earlier batch of the same call: the diagnosis sees the last batch only)";
    bool input_bad = false, named = false;
    if (e->last_input && has_nonfinite(e->last_input, (size_t)B * e->arch.input_channels * H * W, false) == 1) { where = "the network input"; input_bad = true; }
    float* d_copy = nullptr;
    const bool was_keep = e->keep_activations;
    const bool mine = workspace_is_mine(e);        // a shared workspace may hold ANOTHER engine's activations by now: nothing in it is scanned then
    if (!mine && !input_bad)
        return fail(TS2D_ERR_INVALID, "non-finite logits (inf / NaN) in the last forward; its activations cannot be scanned for the first bad layer: the "
                    "shared workspace (ts2d_engine_set_workspace) has been overwritten by another engine's run since - re-run this engine with a "
                    "workspace of its own (ts2d_engine_set_workspace(e, NULL, 0)) to localise");
    // (shared workspace: the diagnostic re-run with one buffer per tensor needs ~3x the memory the caller sized - skipped; the surviving
    //  tensors of this engine's own last run are scanned below)
    if (!input_bad && !was_keep && e->last_input && !e->ws_external) {
        // activations share buffers by liveness: most of the run has been overwritten.  The input of a synchronous call is still
        // there - run it once more with one buffer per tensor (slow path, taken only after an inf / NaN was flagged).
        const size_t nb = (size_t)B * e->arch.input_channels * H * W * sizeof(float);
        if (hipMalloc(reinterpret_cast<void**>(&d_copy), nb) == hipSuccess && hipMemcpy(d_copy, e->last_input, nb, hipMemcpyDeviceToDevice) == hipSuccess) {
            e->keep_activations = true;
            const bool was_fuse0 = e->use_fuse0;
            e->use_fuse0 = false; ++e->opt_gen;      // (the first block as its own kernel: its output can be scanned and named)
            hipStream_t st = e->last_stream ? e->last_stream : e->stream;
            const bool prof = e->profiling; e->profiling = false;
            if (ensure_workspace(e, B, H, W) != TS2D_OK || run_forward(e, d_copy, B, H, W, nullptr, nullptr, st, false) != TS2D_OK ||
                hipStreamSynchronize(st) != hipSuccess) { /* keep the generic message */ }
            e->profiling = prof;
            e->use_fuse0 = was_fuse0; ++e->opt_gen;
            e->last_input = nullptr;
        }
    }
    if (!input_bad)
        for (const Op& op : e->ops) {
            if (op.dst < 0) continue;
            const size_t oi = (size_t)(&op - e->ops.data());
            if (oi < e->fused_away.size() && e->fused_away[oi]) continue;        // not materialised in the last run
            const Tensor& t = e->tensors[op.dst];
            if (!t.resident || !t.data) continue;                              // overwritten by a later activation of the run
            const size_t n = (size_t)B * (H >> t.ly) * (W >> t.lx) * t.C;
            const int f = has_nonfinite(t.data, n, e->last_f16);
            const int g = (f == 0 && t.normed) ? has_nonfinite(t.scale, (size_t)B * t.C, false) : 0;
            if (f == 1 || g == 1) { where = "layer " + op.name + (f == 1 ? "" : " (InstanceNorm statistics)"); named = true; break; }
        }
    if (d_copy) {
        (void)hipFree(d_copy);
        // the diagnostic re-run grew the workspace to one buffer per tensor (~3x): give it back, the next forward re-plans the shared arena
        if (e->ws_busy) { (void)hipEventSynchronize(e->ws_event); e->ws_busy = false; }
        if (e->d_ws && !e->ws_external) { (void)hipFree(e->d_ws); e->d_ws = nullptr; e->ws_bytes = 0; }
        e->ws_precision = -1;     // (re-plan at the next forward)
        for (Tensor& t : e->tensors) { t.data = nullptr; t.resident = false; }
        e->lastB = 0;             // (nothing of that run is readable any more)
    }
    e->keep_activations = was_keep;
    if (named && !was_keep && !d_copy)
        // asynchronous device-pointer call: the input is the caller's, no re-run with private buffers was possible, and most
        // activations of the run have been recycled - the tensor named below is only the first SURVIVING one
        return fail(TS2D_ERR_INVALID, "non-finite logits: first surviving tensor with inf / NaN: %s (earlier activations were recycled; call "
                    "ts2d_engine_set_keep_activations(e, 1) and re-run to localise)%s", where.c_str(), "");
    return fail(TS2D_ERR_INVALID, "non-finite logits: inf / NaN first appears in %s%s", where.c_str(),
                e->precision == TS2D_PRECISION_F32_EXACT ? "" :
                " (the fp16 products of this precision mode need |activation| < 65504 at every conv input: "
                "an overflowing activation - e.g. an un-normalised transposed-conv output - becomes inf; use TS2D_PRECISION_F32_EXACT for such weights)");
}

int ts2d_engine_tiled_inf_flag(const ts2d_engine* e) { return e ? (e->tiled_inf != 0) : 0; }

static int project_coronal_impl(int device, const void* volume, size_t n_elems, int dtype, int nz, int ny, int nx, long long sz,
                                long long sy, long long sx, long long base, float* out_max, float* out_mean, float* out_norm, double* out_stats, int* out_box) {
    if (!volume || !out_max || !out_mean) return fail(TS2D_ERR_INVALID, "ts2d_project_coronal: null argument");
    static const int esize[5] = {2, 1, 4, 2, 4};
    if (dtype < 0 || dtype > 4 || nz < 1 || ny < 1 || nx < 1) return fail(TS2D_ERR_INVALID, "ts2d_project_coronal: bad dtype / extents");
    {   // every element the view can touch must lie inside the buffer
        long long lo = base, hi = base;
        const long long ext[3] = {(long long)(nz - 1) * sz, (long long)(ny - 1) * sy, (long long)(nx - 1) * sx};
        for (int k = 0; k < 3; ++k) { if (ext[k] < 0) lo += ext[k]; else hi += ext[k]; }
        if (lo < 0 || hi >= (long long)n_elems) return fail(TS2D_ERR_INVALID, "ts2d_project_coronal: the strided view leaves the buffer");
    }
    HIP_TRY(hipSetDevice(device));
    const size_t vbytes = n_elems * esize[dtype], obytes = (size_t)nz * nx * sizeof(float);
    char* d = nullptr;
    // [volume | max | mean (contiguous: the two channels of the z-score) | normalised x 2 | partial sums | stats | box]
    const size_t o_proj = align_up(vbytes, 256), o_norm = align_up(o_proj + 2 * obytes, 256), o_part = align_up(o_norm + 2 * obytes, 256);
    const size_t o_stats = o_part + 2 * kZBlocks * sizeof(double), o_box = o_stats + 4 * sizeof(double);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), o_box + 4 * sizeof(int)));
    float* d_max = reinterpret_cast<float*>(d + o_proj);
    float* d_mean = d_max + (size_t)nz * nx;
    hipError_t he = hipMemcpy(d, volume, vbytes, hipMemcpyHostToDevice);
    if (he == hipSuccess) {
        const unsigned grid = (unsigned)(((long long)nz * nx + 255) / 256);
        switch (dtype) {
            case 0: hipLaunchKernelGGL(project_coronal<int16_t>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const int16_t*>(d), nz, ny, nx, sz, sy, sx, base, d_max, d_mean); break;
            case 1: hipLaunchKernelGGL(project_coronal<uint8_t>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const uint8_t*>(d), nz, ny, nx, sz, sy, sx, base, d_max, d_mean); break;
            case 2: hipLaunchKernelGGL(project_coronal<float>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const float*>(d), nz, ny, nx, sz, sy, sx, base, d_max, d_mean); break;
            case 3: hipLaunchKernelGGL(project_coronal<uint16_t>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const uint16_t*>(d), nz, ny, nx, sz, sy, sx, base, d_max, d_mean); break;
            default: hipLaunchKernelGGL(project_coronal<int32_t>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const int32_t*>(d), nz, ny, nx, sz, sy, sx, base, d_max, d_mean); break;
        }
        he = hipGetLastError();
    }
    if (he == hipSuccess && out_norm) {          // per-channel z-score of (max, mean): float64 two-pass statistics, deterministic
        const long long n = (long long)nz * nx;
        float* d_norm = reinterpret_cast<float*>(d + o_norm);
        double* d_part = reinterpret_cast<double*>(d + o_part); double* d_stats = reinterpret_cast<double*>(d + o_stats);
        int* d_box = reinterpret_cast<int*>(d + o_box);
        const int box0[4] = {nz, -1, nx, -1};
        he = hipMemcpy(d_box, box0, sizeof(box0), hipMemcpyHostToDevice);
        if (he == hipSuccess) {
            hipLaunchKernelGGL(zs_partial<0>, dim3(kZBlocks, 2), dim3(256), 0, 0, d_max, n, d_stats, d_part);
            hipLaunchKernelGGL(zs_combine<0>, dim3(1), dim3(2), 0, 0, d_part, n, d_stats);
            hipLaunchKernelGGL(zs_partial<1>, dim3(kZBlocks, 2), dim3(256), 0, 0, d_max, n, d_stats, d_part);
            hipLaunchKernelGGL(zs_combine<1>, dim3(1), dim3(2), 0, 0, d_part, n, d_stats);
            hipLaunchKernelGGL(zs_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_max, n, nx, 2, d_stats, d_norm, d_box);
            he = hipGetLastError();
        }
        if (he == hipSuccess) he = hipMemcpy(out_norm, d_norm, 2 * obytes, hipMemcpyDeviceToHost);
        if (he == hipSuccess && out_stats) he = hipMemcpy(out_stats, d_stats, 4 * sizeof(double), hipMemcpyDeviceToHost);
        if (he == hipSuccess && out_box) he = hipMemcpy(out_box, d_box, 4 * sizeof(int), hipMemcpyDeviceToHost);
    }
    if (he == hipSuccess) he = hipMemcpy(out_max, d_max, obytes, hipMemcpyDeviceToHost);
    if (he == hipSuccess) he = hipMemcpy(out_mean, d_mean, obytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (he != hipSuccess) return fail(TS2D_ERR_HIP, "ts2d_project_coronal failed: %s", hipGetErrorString(he));
    return TS2D_OK;
}


int ts2d_project_coronal(int device, const void* volume, size_t n_elems, int dtype, int nz, int ny, int nx, long long sz,
                         long long sy, long long sx, long long base, float* out_max, float* out_mean) {
    return project_coronal_impl(device, volume, n_elems, dtype, nz, ny, nx, sz, sy, sx, base, out_max, out_mean, nullptr, nullptr, nullptr);
}

int ts2d_project_coronal_zscore(int device, const void* volume, size_t n_elems, int dtype, int nz, int ny, int nx, long long sz,
                                long long sy, long long sx, long long base, float* out_max, float* out_mean, float* out_norm,
                                double* out_stats, int32_t* out_box) {
    if (!out_norm) return fail(TS2D_ERR_INVALID, "ts2d_project_coronal_zscore: null argument");
    return project_coronal_impl(device, volume, n_elems, dtype, nz, ny, nx, sz, sy, sx, base, out_max, out_mean, out_norm, out_stats, out_box);
}
int ts2d_synth_slices(int device, unsigned long long key, unsigned long long first_element, unsigned long long n_elements,
                      float* out_device, void* stream) {
    if (!out_device) return fail(TS2D_ERR_INVALID, "ts2d_synth_slices: null output");
    if (n_elements == 0) return TS2D_OK;
    if (n_elements > (1ull << 40)) return fail(TS2D_ERR_INVALID, "ts2d_synth_slices: %llu elements in one call", n_elements);
    HIP_TRY(hipSetDevice(device));
    const unsigned long long per = 1ull << 30;                      // <= 2^30 elements per launch (grid of 2^22 blocks)
    for (unsigned long long o = 0; o < n_elements; o += per) {
        const unsigned long long m = std::min(per, n_elements - o);
        hipLaunchKernelGGL(synth_normal, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           out_device + o, key, first_element + o, m);
        HIP_TRY(hipGetLastError());
    }
    return TS2D_OK;
}

int ts2d_engine_set_profiling(ts2d_engine* e, int enable) {
    if (!e) return fail(TS2D_ERR_INVALID, "ts2d_engine_set_profiling: null engine");
    e->profiling = enable != 0;
    return TS2D_OK;
}

int ts2d_engine_num_ops(ts2d_engine* e) { return e ? (int)e->n_launched : 0; }

const char* ts2d_engine_op_kernel(ts2d_engine* e, int op) {
    if (!e || op < 0 || (size_t)op >= e->n_launched) return "";
    return e->launches[op].kernel.c_str();
}

const char* ts2d_engine_op_name(ts2d_engine* e, int op) {
    if (!e || op < 0 || (size_t)op >= e->n_launched) return "";
    return e->launches[op].name.c_str();
}

int ts2d_engine_op_times(ts2d_engine* e, float* ms, int n_ops) {
    if (!e || !ms) return fail(TS2D_ERR_INVALID, "ts2d_engine_op_times: null argument");
    if (e->d_prof) {        // TS2D_DBG=256: print and reset the in-kernel phase counters (cycles of wave 0, averaged over workgroups)
        std::vector<unsigned long long> raw(512 * 128), hp(8 * 128, 0);
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(raw.data(), e->d_prof, raw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(e->d_prof, 0, raw.size() * sizeof(unsigned long long)));
        for (size_t oi = 0; oi < 128; ++oi) for (int k = 0; k < 64; ++k) for (int i = 0; i < 8; ++i) hp[8 * oi + i] += raw[512 * oi + 8 * k + i];
        for (size_t oi = 0; oi < e->ops.size() && oi < 128; ++oi)
            if (hp[8 * oi + 7]) {
                const double n = (double)hp[8 * oi + 7];
                // conv3x3_upc: 0/1 phase-1 staging / MFMAs, 2/3 phase 2, 4 stores, 5 / 6 barriers.  p / s2v2: 0 staging (barrier to barrier), 1 MFMAs
                // (+ wait at the next barrier), 3 last MFMAs, 4 bias + stores, 5 statistics.  q: 0 barrier wait, 1 chunk body, 3-5 as p.
                fprintf(stderr, "[phases] %-8s wgs %6.0f  cycles/wg: [0] %8.0f  [1] %8.0f  [2] %8.0f  [3] %8.0f  [4] %8.0f  [5] %8.0f  [6] %8.0f\n",
                        e->ops[oi].name.c_str(), n, hp[8 * oi] / n, hp[8 * oi + 1] / n, hp[8 * oi + 2] / n, hp[8 * oi + 3] / n, hp[8 * oi + 4] / n, hp[8 * oi + 5] / n, hp[8 * oi + 6] / n);
            }
    }
    if ((size_t)n_ops > e->n_launched) n_ops = (int)e->n_launched;
    for (int i = 0; i < n_ops; ++i) {
        HIP_TRY(hipEventSynchronize(e->launches[i].e1));
        HIP_TRY(hipEventElapsedTime(&ms[i], e->launches[i].e0, e->launches[i].e1));
    }
    return TS2D_OK;
}

int ts2d_engine_debug_tensor(ts2d_engine* e, const char* name, float* out, size_t capacity, int32_t dims[4]) {
    if (!e || !name || !out || !dims) return fail(TS2D_ERR_INVALID, "ts2d_engine_debug_tensor: null argument");
    const int ti = tensor_index(e, name);
    if (ti < 0) return fail(TS2D_ERR_INVALID, "no tensor '%s'", name);
    if (!e->lastB) return fail(TS2D_ERR_STATE, "no forward has run on this engine (or its workspace was replaced since): nothing to read for '%s'", name);
    if (!workspace_is_mine(e))
        return fail(TS2D_ERR_STATE, "tensor '%s': the shared workspace (ts2d_engine_set_workspace) has been overwritten by another engine's forward "
                    "since this engine's last run - run this engine last, or give it a workspace of its own", name);
    const Tensor& t = e->tensors[ti];
    if (t.data && !t.resident)
        return fail(TS2D_ERR_STATE, "tensor '%s' was overwritten by a later activation of the same run (buffers are shared by liveness): "
                    "call ts2d_engine_set_keep_activations(e, 1) before the forward", name);
    for (size_t oi = 0; oi < e->ops.size() && oi < e->fused_away.size(); ++oi)
        if (e->fused_away[oi] && e->ops[oi].dst == ti && oi == 0)
            return fail(TS2D_ERR_INVALID, "tensor '%s' was not materialised by the last run: the first block is recomputed inside the second "
                        "(ts2d_engine_set_option(e, \"fuse0\", 0) runs it as its own kernel)", name);
    for (size_t oi = 0; oi < e->ops.size() && oi < e->fused_away.size(); ++oi)
        if (e->fused_away[oi] && e->ops[oi].dst == ti)
            return fail(TS2D_ERR_INVALID, "tensor '%s' was not materialised by the last run: the transposed conv is composed into the next block "
                        "(ts2d_engine_set_option(e, \"upc\", 0) runs it as its own kernel)", name);
    const int B = e->lastB, h = e->lastH >> t.ly, w = e->lastW >> t.lx, C = t.C, Cu = t.Cu;      // (the caller's channels; C - Cu added ones hold zeros)
    dims[0] = B; dims[1] = Cu; dims[2] = h; dims[3] = w;
    const size_t n = (size_t)B * C * h * w;
    if (capacity < (size_t)B * Cu * h * w) return fail(TS2D_ERR_INVALID, "tensor '%s' needs %zu floats, capacity is %zu", name, (size_t)B * Cu * h * w, capacity);
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->last_stream ? e->last_stream : e->stream));
    std::vector<float> raw(n), sc, sh;
    if (e->last_f16) {
        std::vector<uint16_t> rh(n);
        HIP_TRY(hipMemcpy(rh.data(), t.data, n * sizeof(uint16_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) raw[i] = f16_to_f32(rh[i]);
    } else {
        HIP_TRY(hipMemcpy(raw.data(), t.data, n * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (t.normed) {
        sc.resize((size_t)B * C); sh.resize((size_t)B * C);
        HIP_TRY(hipMemcpy(sc.data(), t.scale, sc.size() * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(sh.data(), t.shift, sh.size() * sizeof(float), hipMemcpyDeviceToHost));
    }
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < h * w; ++p)
            for (int c = 0; c < Cu; ++c) {
                float v = raw[((size_t)b * h * w + p) * C + c];
                if (t.normed) { v = v * sc[(size_t)b * C + c] + sh[(size_t)b * C + c]; v = v > 0.f ? v : v * e->arch.leaky_slope; }
                out[((size_t)b * Cu + c) * h * w + p] = v;
            }
    return TS2D_OK;
}

size_t ts2d_engine_device_bytes(ts2d_engine* e) {
    return e ? e->weight_floats * sizeof(float) + (e->ws_external ? 0 : e->ws_bytes) + e->stage_bytes + e->sw_bytes : 0;
}

int ts2d_engine_destroy(ts2d_engine* e) {
    if (!e) return TS2D_OK;
    (void)hipSetDevice(e->device);
    if (e->ws_busy) (void)hipEventSynchronize(e->ws_event);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->ws_event) (void)hipEventDestroy(e->ws_event);
    for (Launch& l : e->launches) { if (l.e0) (void)hipEventDestroy(l.e0); if (l.e1) (void)hipEventDestroy(l.e1); }
    if (e->d_ws && !e->ws_external) (void)hipFree(e->d_ws);
    if (e->d_stage) (void)hipFree(e->d_stage);
    if (e->d_flags) (void)hipFree(e->d_flags);
    if (e->d_prof) (void)hipFree(e->d_prof);
    if (e->d_sw) (void)hipFree(e->d_sw);
    if (e->d_weights) (void)hipFree(e->d_weights);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return TS2D_OK;
}

}  // extern "C"
