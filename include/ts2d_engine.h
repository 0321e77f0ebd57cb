/* ts2d_engine.h - C-ABI of the MI355X-native 2-D U-Net inference engine (libts2d_engine.so).
 *
 * Drop-in boundary.  The reference has no FFI: its seam is the duck-typed predictor object
 * consumed by `_run_predict` (reference ts2d/core/inference/prediction_worker.py:177-242).  These entry points are
 * what a ctypes/cffi binding placed at that seam binds; each cites the reference interface it replaces.
 * Plain pointers and sizes only - no torch / HIP types in any signature (streams travel as void*).
 *
 * Error convention (replaces Python exceptions, reference prediction_worker.py:183-242 / nnu.py:217-219): every
 * function returns 0 on success or a negative ts2d_status; ts2d_last_error() returns a thread-local message that
 * the Python layer converts to RuntimeError.  The library never aborts the process.
 *
 * Threading: one caller thread per engine handle (reference: one worker process per sub-model, one task at a
 * time, prediction_worker.py:127-165).  Handles are independent (different sub-models / GPUs).
 */
#ifndef TS2D_ENGINE_H
#define TS2D_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TS2D_MAX_STAGES 16

typedef enum {
    TS2D_OK = 0,
    TS2D_ERR_INVALID = -1,     /* bad argument / unsupported architecture or shape */
    TS2D_ERR_HIP = -2,         /* a HIP runtime call or kernel launch failed */
    TS2D_ERR_NOMEM = -3,       /* device or host allocation failed */
    TS2D_ERR_STATE = -4        /* engine not initialised (e.g. weights not yet broadcast) */
} ts2d_status;

/* Architecture descriptor = the `arch_kwargs` of nnU-Net's PlainConvUNet that
 * `nnUNetPredictor.initialize_from_trained_model_folder` (reference call site ts2d/core/inference/nnu.py:165) reads
 * from plans.json.  Supported subset: Conv2d 3x3, stride (1, 1) in the first stage and 1 or 2 PER AXIS afterwards (nnU-Net's
 * planner pools each axis separately: (2, 2) until one axis is exhausted, then (2, 1) / (1, 2)), InstanceNorm2d(affine), LeakyReLU,
 * ConvTranspose2d upsampling with kernel = stride = the stride of the stage below, 1x1 head.  features[]: any positive width, features[0] <= 64; a width that
 * is not a multiple of 32 runs rounded up to one with zero weights in the added channels (exact; the weight blob keeps the caller's layout).
 * (2, 2) stages run the dedicated kernels; any other stride runs a generic implicit-GEMM kernel (correct, not tuned). */
typedef struct {
    int32_t input_channels;                 /* C: len(dataset_json['channel_names']) (prediction_worker.py:78) */
    int32_t num_classes;                    /* K: number of segmentation heads (multilabel: one per label) */
    int32_t n_stages;
    int32_t features[TS2D_MAX_STAGES];
    int32_t n_conv_enc[TS2D_MAX_STAGES];
    int32_t n_conv_dec[TS2D_MAX_STAGES];    /* n_stages-1 entries, bottom-up (decoder.stages.{j}) */
    float norm_eps;                         /* InstanceNorm2d eps (1e-5) */
    float leaky_slope;                      /* LeakyReLU negative_slope (0.01) */
    int32_t strides[TS2D_MAX_STAGES][2];    /* ABI 7: arch_kwargs['strides'][s] = (along H, along W), 1 or 2 each; [0] = (1, 1).
                                             * An all-zero entry s >= 1 reads as (2, 2) (descriptors written for ABI <= 6). */
} ts2d_arch_desc;

/* Arithmetic of the dense 3x3 contractions (storage, accumulation, statistics and I/O are fp32 in both modes):
 *   TS2D_PRECISION_F32_EXACT       v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 FMA chain (157 TFLOP/s peak).
 *   TS2D_PRECISION_F32_SPLIT_F16X3 (default) operands split into fp16 hi + lo (22 significant bits) and multiplied
 *       as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation; measured as close to the fp64
 *       truth as ATen's native fp32 conv on the canonical net.  Activations must stay below
 *       65504 in magnitude (always true after InstanceNorm for |gamma| < 127). */
#define TS2D_PRECISION_F32_EXACT 0
#define TS2D_PRECISION_F32_SPLIT_F16X3 1
/*   TS2D_PRECISION_F16  "mixed fp16" (BASELINE configs 3 and 5): activations stored as fp16 in HBM (half the traffic), weights
 *       rounded to fp16, ONE fp16 MFMA product per MAC, fp32 accumulation and fp32 InstanceNorm statistics; logits are
 *       still returned as fp32.  NOT within the fp32 parity tolerance (logit error ~1e-2); selected explicitly. */
#define TS2D_PRECISION_F16 2

typedef struct ts2d_engine ts2d_engine;

/* Create an engine on HIP device `device`.
 * Replaces: nnUNetPredictor(...).initialize_from_trained_model_folder(model, folds, checkpoint) for ONE fold
 * (reference nnu.py:164-165): network construction + load_state_dict.
 * `weights`: host pointer to the fp32 blob = every parameter tensor in PyTorch layout, concatenated in program order
 * (encoder.stages.{s}.0.convs.{i}.{conv.weight,conv.bias,norm.weight,norm.bias} ..., decoder.transpconvs.{j}.{weight,
 * bias}, decoder.stages.{j}.convs.{i}..., decoder.seg_layers.{n-2}.{weight,bias}); n_floats must match exactly.
 * `weights` may be NULL: the engine is then created with uninitialised device weights that MUST be filled by a
 * broadcast into ts2d_engine_weight_buffer() followed by ts2d_engine_weights_ready() (multi-GPU replicas). */
int ts2d_engine_create(const ts2d_arch_desc* arch, const float* weights, size_t n_floats, int device,
                       ts2d_engine** out);

/* Replace the weights of an existing engine (fold switch: `network.load_state_dict(params)` in
 * predict_logits_from_preprocessed_data, reference call site prediction_worker.py:209). */
int ts2d_engine_load_weights(ts2d_engine* e, const float* weights, size_t n_floats);

/* Select the arithmetic mode (see TS2D_PRECISION_*); takes effect at the next forward. */
int ts2d_engine_set_precision(ts2d_engine* e, int mode);

/* Kernel-dispatch options (ABI 6; replaces the TS2D_* environment switches of ABI <= 5 - a product library must not change kernels
 * because of its caller's environment).  Several ops have two complete, parity-tested kernels (e.g. the decoder entry composed
 * with its ConvTranspose2d, or as two kernels); an option picks one for THIS handle, takes effect at the next reserve / forward and
 * never changes results beyond fp32 summation order (every switch has a parity test on each of its sides).  Names (value 0 / 1 unless noted):
 *   "upc" composed decoder entry | "upq" its 512-thread variant | "upq_min" (int) least coarse channels for it | "up0" dedicated
 *   level-0 composed kernel | "u0seg" (int) its tiles per workgroup segment, 0 = automatic | "q" persistent 16x32-tile stride-1 kernel |
 *   "one" one-image-tile kernels | "res" resident-weight 32 -> 32 kernel | "fuse0" first block recomputed inside the second |
 *   "s2v2" 512-thread stride-2 kernel | "h32", "h2", "h2_min" (int), "uh2": the 16-bit mode's variants | "flex" the composed decoder
 *   entry on tiles that follow the level's extent where it is no multiple of 8 x 32 pixels (0: transposed conv + conv there) |
 *   "flex2" (int) the 512-thread stride-2 kernel on tiles that divide such a level (0: off, 1: 16-bit mode only, 2: every mode) |
 *   "first_split" the first block's K = 9 C contraction as one fp16 hi / lo split product (0: exact fp32 MFMA) |
 *   "sbk" the small-batch dispatch: where the preferred kernel of an op would launch fewer workgroups than the device has CUs (the
 *   <= 32 x 32 levels of one ... eight slices - what TS2D.predict and the reference's B = 1 loop run), K is split over more workgroups
 *   (deterministic two-phase reduction) and a composed decoder entry runs as transposed conv + conv.  With "sbk" on (the default) a
 *   slice's result is bit-identical between two batches only if both take the same path (e.g. any two batches >= 32 of the canonical
 *   net); across paths it agrees to fp32 summation order (3e-5 on the logits).  The same batch always reproduces its bits.
 * Most of these have had one measured winner for rounds ("q", "res", "one", "s2v2", "h32", "upq", "up0" ...): they are test scaffolding -
 * the way the parity suite reaches the second kernel of an op - not tuning knobs of the product.
 * Unknown names and out-of-range values return TS2D_ERR_INVALID.  The reference has no counterpart (one code path through torch:
 * ts2d/core/inference/prediction_worker.py:209); the callers are this repo's tests and A/B scripts. */
int ts2d_engine_set_option(ts2d_engine* e, const char* name, int value);

/* Weight-broadcast hook: device pointer + byte size of the packed weight arena.  Rank 0 creates with
 * weights, the other ranks with NULL; all ranks broadcast this buffer (RCCL, root 0), then call
 * ts2d_engine_weights_ready(). */
int ts2d_engine_weight_buffer(ts2d_engine* e, void** dev_ptr, size_t* n_bytes);
int ts2d_engine_weights_ready(ts2d_engine* e);

/* One batched forward pass = `network(x)` (reference: nnUNetPredictor.network.__call__ inside
 * _internal_maybe_mirror_and_predict; the reference always uses B = 1).
 *   input        [B, C, H, W] fp32 NCHW (what `data[None]` is in the reference), host or device memory
 *   logits       [B, K, H, W] fp32 NCHW or NULL
 *   mask_packed  [B, K, H, W/32] uint32 or NULL: bit (x & 31) of word x>>5 = (sigmoid(float(logit)) > 0.5), the
 *                multilabel export predicate (reference export_prediction_from_logits, prediction_worker.py:215-221)
 *   H, W         multiples of the product of the strides along their axis (2^(n_stages-1) for an isotropic plan); W multiple of 32
 *                when mask_packed != NULL
 *   on_device    nonzero: input/logits/mask_packed are device pointers; zero: host pointers (staged by the engine)
 *   stream       hipStream_t as void* (NULL = the engine's own stream).  The call is asynchronous when on_device
 *                != 0 (caller synchronises the stream) and synchronous otherwise. */
int ts2d_engine_forward(ts2d_engine* e, const float* input, int B, int H, int W, float* logits,
                        uint32_t* mask_packed, int on_device, void* stream);

/* Result check of the LAST forward / predict_tiled (synchronises it): TS2D_OK, or TS2D_ERR_INVALID when a logit came out inf / NaN,
 * with ts2d_last_error() naming the first layer (program order) whose output holds a non-finite value.  The split and f16 modes
 * multiply fp16 operands: an activation of magnitude >= 65504 at a conv input (impossible after InstanceNorm for |gamma| < 127,
 * possible for an un-normalised transposed-conv output with adversarial weights) overflows to inf - this check turns that into an
 * error instead of a silent inf; TS2D_PRECISION_F32_EXACT has no such limit.  (The network INPUT is not under that limit: the first block
 * checks every tile's input inside its kernel and computes a tile that holds |x| >= 262 016, an inf or a NaN with the exact fp32
 * MFMAs - same result, slower.)  Host-buffer forwards and ts2d_engine_predict_tiled
 * run it themselves; after an asynchronous device-pointer forward the caller may.  (Reference convention: a failing prediction
 * raises, ts2d/core/inference/prediction_worker.py:211-212; upstream nnU-Net only checks the aggregated array for inf.) */
int ts2d_engine_check(ts2d_engine* e);

/* Sliding-window inference of ONE preprocessed 2-D image on the device (replaces the body of nnU-Net's
 * predict_sliding_window_return_logits + _internal_maybe_mirror_and_predict for one fold: reference call site
 * ts2d/core/inference/prediction_worker.py:209; SURVEY.md rows A3-A5, and A7 for `seg`).
 *   image          host [C, Hp, Wp] fp32, already padded to at least the patch (pad_nd_image is the caller's job)
 *   tile_y/tile_x  host, n_tiles tile origins in upstream order (compute_steps_for_sliding_window)
 *   mirror_mask    bit 0: mirror spatial axis 0 (H), bit 1: axis 1 (W); variants run in upstream order H, W, HW
 *   gaussian_f16   host [patch_h, patch_w] IEEE half bits (compute_gaussian), or NULL for no weighting
 *   logits_f16     host [K, Hp, Wp] half bits: aggregated logits / n_predictions in upstream's float16 buffers (rounding
 *                  order: ts2d_engine_set_tile_dtype; equal bit for bit to the repo's ATen-pinned oracle); may be NULL
 *   seg_u8         host [K, Hp, Wp]: sigmoid(float(logit)) > 0.5 of the aggregated logits (multilabel export); may be NULL
 * All tiles x mirror variants go through the network as one batch (chunks of at most 64 rows).  Synchronous. */
int ts2d_engine_predict_tiled(ts2d_engine* e, const float* image, int Hp, int Wp, int patch_h, int patch_w, int n_tiles,
                              const int32_t* tile_y, const int32_t* tile_x, int mirror_mask, const uint16_t* gaussian_f16,
                              uint16_t* logits_f16, uint8_t* seg_u8);

/* Blend order of ts2d_engine_predict_tiled (upstream `prediction *= gaussian; predicted_logits[sl] += prediction` with
 * float16 `predicted_logits`; reached from ts2d/core/inference/prediction_worker.py:209):
 *   TS2D_TILE_F32 (default) the reference's CPU path (nnu.py:161-163: device=cpu when torch.cuda.is_available() is false - no autocast): the tile prediction is
 *       fp32, the product is fp32 x float(half gaussian) in fp32, the sum is a float add with ONE rounding into the half buffer;
 *   TS2D_TILE_F16 the reference's CUDA path (fp16 autocast): the tile is half, the product and the sum each round to half. */
#define TS2D_TILE_F32 0
#define TS2D_TILE_F16 1
int ts2d_engine_set_tile_dtype(ts2d_engine* e, int mode);

/* Activation memory (ABI 5).  By default the activations of a forward share one arena by LIVENESS: a tensor's bytes are reused once
 * its last reader has run (the encoder skips live until their decoder block) - ~110 instead of ~340 MB per 2x512x512 slice of the
 * canonical net.  enable != 0 gives every tensor its own buffer again: needed before a forward whose intermediate tensors are to be
 * read back with ts2d_engine_debug_tensor (a reused tensor reports TS2D_ERR_STATE there).  ts2d_engine_check needs no switch: when a
 * synchronous call flagged inf / NaN it re-runs that input once with private buffers to name the first bad layer.  Takes effect at
 * the next ts2d_engine_reserve / forward.  (The reference keeps every intermediate alive only as long as torch's autograd-free
 * forward does: ts2d/core/inference/prediction_worker.py:209.) */
int ts2d_engine_set_keep_activations(ts2d_engine* e, int enable);

/* 1 if the last ts2d_engine_predict_tiled call produced an infinite aggregated float16 logit - upstream's
 * "Encountered inf in predicted array" check of predict_sliding_window_return_logits (reached from
 * ts2d/core/inference/prediction_worker.py:209), evaluated on the device instead of a host pass over the array. */
int ts2d_engine_tiled_inf_flag(const ts2d_engine* e);

/* Coronal maximum + mean projection of a volume on the device (reference ts2d/tool.py:152-160 -> ts2d/core/util/image.py:46-101).
 *   volume     host pointer to the ORIGINAL contiguous buffer of `n_elems` elements of type `dtype`
 *              (0 = int16, 1 = uint8, 2 = float32, 3 = uint16, 4 = int32)
 *   nz, ny, nx extents of the REORIENTED view [z][y][x] (DICOMOrient 'RAI'); sz, sy, sx its signed ELEMENT strides and `base` the
 *              element offset of view[0][0][0] in the buffer (so axis flips / permutations need no host copy)
 *   out_max, out_mean  host [nz][nx] float32: projections along y.  The mean is real-valued for every input type: the sum in index
 *              order (exact for integer volumes) divided in double, rounded once to float - ITK's MeanProjectionImageFilter followed by
 *              the reference's Cast to Float32 (ts2d/tool.py:182-185); the reference's pre-projected sample assets pin it
 *              (oracle/input_oracle.py).  Synchronous. */
int ts2d_project_coronal(int device, const void* volume, size_t n_elems, int dtype, int nz, int ny, int nx, long long sz,
                         long long sy, long long sx, long long base, float* out_max, float* out_mean);

/* The same projection followed, on the device, by the per-channel z-score nnU-Net applies to the 2-channel (max, mean) image
 * before the network (ZScoreNormalization without mask: (x - mean) / max(std, 1e-8), population std; reference flow
 * ts2d/tool.py:152-160,182-185 -> DefaultPreprocessor.run_case, ts2d/core/inference/prediction_worker.py:194-199): mean and std in
 * float64, two passes, fixed summation order.  out_norm = [2][nz][nx] float (the network input when nz, nx need no padding),
 * out_stats = {mean_max, std_max, mean_mean, std_mean} (may be NULL), out_box = {first, last non-zero row, first, last non-zero
 * column} over both channels (may be NULL): nnU-Net crops to that box BEFORE normalising, so the caller uses out_norm only when
 * the box is the whole image.  (ABI 4) */
int ts2d_project_coronal_zscore(int device, const void* volume, size_t n_elems, int dtype, int nz, int ny, int nx, long long sz,
                                long long sy, long long sx, long long base, float* out_max, float* out_mean, float* out_norm,
                                double* out_stats, int32_t* out_box);

/* Synthetic slice stream on the device (BASELINE config 4: "synthetic 10k-slice stream", generated per rank from (seed, slice
 * index) so that no host transfer skews the timing).  Writes n_elements fp32 values, approximately N(0,1), to device memory:
 * element i of the call = element (first_element + i) of the stream identified by `key`; a value depends on (key, element index)
 * only, so every rank can produce any block of the stream, and it is bit-identical to the host generator
 * totalsegmentator2d_amd/prng.py (key = prng.key(seed, stream)).  Asynchronous on `stream` (hipStream_t as void*, NULL = default).
 * The reference has no counterpart (it reads files); the bench and the multi-GPU tests are the callers. */
int ts2d_synth_slices(int device, unsigned long long key, unsigned long long first_element, unsigned long long n_elements,
                      float* out_device, void* stream);

/* Pre-allocate the activation workspace for (B, H, W) (reference warm-up contract: a zero patch is pushed through
 * the predictor once at start-up, prediction_worker.py:74-96,136-138). */
int ts2d_engine_reserve(ts2d_engine* e, int B, int H, int W);

/* Caller-provided activation workspace (ABI 6).  ts2d_engine_workspace_bytes: the bytes ts2d_engine_reserve(B, H, W) would
 * allocate under the engine's current precision mode / options.  ts2d_engine_set_workspace: use [dev_ptr, dev_ptr + n_bytes) (256-byte
 * aligned device memory owned by the caller, alive until it is replaced or the engine destroyed) instead of an allocation of the
 * engine's own; dev_ptr = NULL returns to that.  A forward that needs more than n_bytes fails with TS2D_ERR_NOMEM.  Purpose: the five
 * sub-models of ts2d-v2 run one after the other on one stream (the reference drives them sequentially, ts2d/tool.py:110-112), so ONE
 * workspace of the largest size serves all five engines.  Engines that share a workspace must be driven on the same stream (or be
 * ordered by the caller): the engine's own cross-stream ordering covers the runs of ONE handle only. */
int ts2d_engine_workspace_bytes(ts2d_engine* e, int B, int H, int W, size_t* n_bytes);
int ts2d_engine_set_workspace(ts2d_engine* e, void* dev_ptr, size_t n_bytes);

/* Per-op device timing (HIP events on the launch stream).  enable != 0 brackets every kernel of subsequent forwards
 * with events; ts2d_engine_op_times returns for the LAST forward the elapsed ms per op (n_ops entries, program
 * order; see ts2d_engine_op_name) - synchronises the stream. */
int ts2d_engine_set_profiling(ts2d_engine* e, int enable);
int ts2d_engine_num_ops(ts2d_engine* e);
const char* ts2d_engine_op_name(ts2d_engine* e, int op);
/* The kernel that served entry `op` of the last profiled forward ("conv3x3_f16x3_q", "conv3x3_upc<64>", "finalize_stats", ...):
 * the dispatch depends on precision mode, channel counts and tile geometry, bench.py groups its roofline blocks by this. */
const char* ts2d_engine_op_kernel(ts2d_engine* e, int op);
int ts2d_engine_op_times(ts2d_engine* e, float* ms, int n_ops);

/* Test/debug accessor (not on the product path): copies activation tensor `name` ("enc0.c1", "dec3.up", ... - the
 * op names of ts2d_engine_op_name) of the LAST forward to host as NCHW fp32 [B,C,h,w], with the InstanceNorm +
 * LeakyReLU that its consumer applies on load already applied (i.e. what torch holds after the block).
 * `capacity` = floats available in out; *dims receives {B, C, h, w}.  Synchronises the engine. */
int ts2d_engine_debug_tensor(ts2d_engine* e, const char* name, float* out, size_t capacity, int32_t dims[4]);

/* Bytes of device memory currently held (weights + workspace). */
size_t ts2d_engine_device_bytes(ts2d_engine* e);

int ts2d_engine_destroy(ts2d_engine* e);

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* ts2d_last_error(void);

/* ABI version of this header (bumped on any signature change). */
int ts2d_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TS2D_ENGINE_H */
