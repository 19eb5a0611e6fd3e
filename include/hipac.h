/*
 * hipac.h -- C ABI of libhipac_hip.so, the MI355X (gfx950) implementation of
 * HiPAC's patch-inference hot path.
 *
 * The reference (anacarsi/ss25_Hierarchical_Multiscale_Image_Classification)
 * exposes no plugin / FFI interface for this path: it is reached through plain
 * Python call sites.  Each entry point below names the reference code whose
 * arithmetic it replaces (file:line relative to the reference root); the
 * Python-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross this boundary
 *   - every `const void* / void*` data pointer is DEVICE memory unless the
 *     parameter comment says "host"
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as
 *     void*; NULL = the default stream); no entry point synchronises the device
 *     except hipac_resnet18_pack (one-time upload) and hipac_weights_free
 *   - the caller owns every buffer including the workspace; the library owns
 *     only the packed-weights handle
 *   - return value: 0 on success, otherwise a hipError_t value or one of the
 *     HIPAC_E* codes; nothing throws across the ABI; hipac_last_error() gives
 *     a thread-local message for the last failure
 *   - re-entrant: the library keeps no mutable state between calls except the
 *     thread-local error string; a weights handle is read-only after packing and
 *     may be used from several streams concurrently (large forwards of one handle
 *     share its second launch lane, a stream created at pack time: they stay
 *     correct -- fork / join is by events -- but serialise on that lane)
 *   - a handle belongs to the device that was current in hipac_resnet18_pack;
 *     calling it with another device current returns HIPAC_EINVAL
 *   - developer knobs (HIPAC_SUBBATCH, HIPAC_GROUP, HIPAC_LANES, HIPAC_FUSE_STEM,
 *     HIPAC_STEM_STRIP) are read from the environment on every call; they select
 *     among equivalent schedules and must not change while calls are in flight
 */
#ifndef HIPAC_H_
#define HIPAC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPAC_ABI_VERSION 8  /* 8: HIPAC_PREC_FP16Q8; 2: the native training entry points (round 2); 3: HIPAC_PREC_FP16X3; 4: hipac_train_amp_*; 5: hipac_augment_views; 6: hipac_jpeg_decode_tiles; 7: larger scratch of hipac_cross_entropy_fwd_bwd, workspaces of the fp32 training step */

/* error codes (positive small values are hipError_t) */
#define HIPAC_EINVAL (-1)     /* bad argument (shape, enum, null pointer, alignment) */
#define HIPAC_EWORKSPACE (-2) /* workspace too small */
#define HIPAC_EUNSUPPORTED (-3)

/* arithmetic type of the network's MFMA operands (accumulation is always fp32) */
#define HIPAC_PREC_BF16 0
#define HIPAC_PREC_FP16 1
#define HIPAC_PREC_FP32 2  /* debugging reference: fp32 storage, exact f32 MFMA (1/16 of the bf16 rate); no uint8 input */
#define HIPAC_PREC_FP16X3 3 /* parity mode: every weight and activation is a (hi, lo) pair of fp16 numbers and every
                               product is hi*hi + hi*lo + lo*hi on the fp16 MFMA with fp32 accumulation (~2^-22
                               relative per term): meets the reference's fp32 results (src/main.py:870) to 1e-3.
                               NHWC4_PAD input is float32[B,230,232,4] in this mode. */
#define HIPAC_PREC_FP16Q8 4 /* the faster parity mode: the pair layout of FP16X3, hi*hi on the fp16 MFMA and the two cross
                               products hi*lo + lo*hi of the 3x3 / stride 1 convolutions on the e4m3 MX MFMA with constant
                               scales (2 MFMA time units per term instead of 3): logits within ~5e-5 of the reference's fp32
                               results, labels identical.  Inputs as for FP16X3. */

/* input layouts accepted by hipac_resnet18_forward */
#define HIPAC_IN_NCHW_F32 0   /* float32[B,3,224,224], the reference's layout (src/main.py:870) */
#define HIPAC_IN_NHWC4_PAD 1  /* native: T[B,230,232,4] (T = bf16|fp16 per precision), the
                                 224x224 image at rows 3..226, cols 3..226, channel 3 and the
                                 border all zero.  Written directly by hipac_tile_preprocess. */

#define HIPAC_IN_U8_HWC 2     /* uint8[B,224,224,3] raw RGB patches (e.g. hipac_tile_preprocess's
                                 HIPAC_OUT_U8_HWC): ToTensor + Normalize (src/main.py:815-816) are
                                 applied inside the stem kernel, bit-identical to the other paths */

/* output formats of hipac_tile_preprocess */
#define HIPAC_OUT_NCHW_F32 0     /* float32[n,3,224,224] == Resize->ToTensor->Normalize */
#define HIPAC_OUT_NHWC4_PAD_BF16 1
#define HIPAC_OUT_NHWC4_PAD_FP16 2
#define HIPAC_OUT_U8_HWC 3       /* uint8[n,224,224,3]: the resized pixels before ToTensor */

#define HIPAC_PATCH 224
#define HIPAC_PAD_H 230
#define HIPAC_PAD_W 232

int hipac_abi_version(void);
const char* hipac_last_error(void);

/* ------------------------------------------------------------------------- *
 * ResNet18 weights
 * ------------------------------------------------------------------------- */

/* One conv + its eval-mode BatchNorm, HOST float32 pointers in PyTorch layout. */
typedef struct hipac_convbn {
  const float* conv_w;   /* [Cout, Cin, kh, kw] */
  const float* bn_gamma; /* [Cout] (bn.weight)       */
  const float* bn_beta;  /* [Cout] (bn.bias)         */
  const float* bn_mean;  /* [Cout] (bn.running_mean) */
  const float* bn_var;   /* [Cout] (bn.running_var)  */
} hipac_convbn_t;

/* The tensors of torchvision.models.resnet18 as the reference instantiates it
 * (src/models/resnet.py:25, :45, :63-65; src/models/simclr.py:17).
 * block[2*s + b][c] = layer{s+1}.{b}.conv{c+1}/bn{c+1};  down[s-1] =
 * layer{s+1}.0.downsample.{0,1} for s = 1..3.  fc_w/fc_b may be NULL
 * (fc = Identity, resnet.py:46) and then num_classes must be 0. */
typedef struct hipac_resnet18_params {
  hipac_convbn_t stem;        /* conv1 [64,3,7,7] + bn1 */
  hipac_convbn_t block[8][2];
  hipac_convbn_t down[3];
  const float* fc_w;          /* host float32 [num_classes, 512] or NULL */
  const float* fc_b;          /* host float32 [num_classes] or NULL */
  int32_t num_classes;        /* 0..16 */
  float bn_eps;               /* 1e-5 */
} hipac_resnet18_params_t;

typedef struct hipac_weights hipac_weights_t;

/* Fold BN into the convs (w' = w*gamma/sqrt(var+eps), b' = beta - mean*gamma/
 * sqrt(var+eps)), round to `precision`, repack to the MFMA-friendly
 * [Cout][kh][kw][Cin] layout and upload.  Replaces the per-forward
 * conv->batch_norm pairs torchvision executes (call sites resnet.py:38-40,
 * :54-55, :69-77).  Synchronous; call once per state_dict. */
int hipac_resnet18_pack(const hipac_resnet18_params_t* params, int precision, hipac_weights_t** out);
void hipac_weights_free(hipac_weights_t* w);
int hipac_weights_precision(const hipac_weights_t* w);
int hipac_weights_num_classes(const hipac_weights_t* w);

/* Bytes of caller-provided scratch needed for a forward of `batch` patches. */
size_t hipac_resnet18_workspace_bytes(int batch, int precision);

/* Batched ResNet18 eval forward.  Replaces
 *   ResNet18FeatureExtractor.forward   src/models/resnet.py:38-40  -> feats
 *   UnifiedResNet.forward              src/models/resnet.py:54-55  -> feats | logits
 *   ResNet18Classifier.forward         src/models/resnet.py:69-77  -> logits
 * as called from src/main.py:870 (extract_features), :920, :505/:519 (train /
 * val scoring) and :1009.
 *   x       : `batch` patches in `in_layout`
 *   feats   : float32[batch,512] or NULL
 *   logits  : float32[batch,num_classes] or NULL (requires fc in the handle)
 *   labels  : int64[batch] argmax(logits, dim=1) (src/main.py:510) or NULL
 */
int hipac_resnet18_forward(const hipac_weights_t* w, const void* x, int batch, int in_layout,
                           float* feats, float* logits, int64_t* labels,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Profiling aid: run ops first_op..last_op of the trunk (0 stem conv, 1 max-pool, then
 * per stage conv1(b0) [1x1 proj] conv2(b0) conv1(b1) conv2(b1); 21 ops) on the
 * activations a previous hipac_resnet18_forward(x, in_layout) left in `workspace`
 * (`x` is re-read by op 0 for the uint8 / native layouts).
 * Ops 0..10 (stem, pool, layer1, layer2) act on the first internal sub-batch
 * (min(batch, 512) images), ops 11..20 (layer3, layer4) on all `batch` images, which
 * must fit one internal group (4096).  Used by bench.py to time single kernels with
 * events on the caller's stream; not part of the reference's surface. */
int hipac_resnet18_run_ops(const hipac_weights_t* w, const void* x, int in_layout, void* workspace,
                           size_t workspace_bytes, int batch, int first_op, int last_op, void* stream);
#define HIPAC_NUM_OPS 21

/* Debug / test tap: copy one intermediate activation of the LAST forward run
 * with this workspace to `dst` as float32 NCHW.  `tap`: 0 stem(after relu),
 * 1 maxpool, 2..9 output of block 0..7.  Used only by the layer-wise parity
 * tests.  Tap 9: by default the last conv leaves per-image partial sums of the
 * global average pool, not its fp32 map (round 4; HIPAC_POOL_HEAD=0 restores the
 * map); the tap then re-runs that conv with the map epilogue into the workspace
 * (which it therefore writes, `const` notwithstanding) before exporting it. */
int hipac_resnet18_tap(const hipac_weights_t* w, const void* workspace, int batch, int tap,
                       float* dst, void* stream);

/* ------------------------------------------------------------------------- *
 * Tile / crop / whiteness / resize / normalise
 * ------------------------------------------------------------------------- */

/* Pillow's resampling tables for one window size P -> 224 (HOST int32).
 * bounds[224*2] = (first source index, tap count); kk[224*ksize] = 22-bit
 * fixed-point weights.  Computed in IEEE double exactly like
 * Pillow's precompute_coeffs / normalize_coeffs_8bpc (libImaging/Resample.c),
 * the code torchvision.transforms.Resize reaches for a PIL image
 * (reference call site src/main.py:814).  Returns ksize, or <0 on error.
 * `bounds`/`kk` may be NULL to query ksize only. */
int hipac_resample_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int kk_stride);
/* level base pointer and pitch must be 16-byte aligned when P > 224. */

/* Per-window crop + white border pad + whiteness sum + antialiased bilinear
 * resize to 224x224 + ToTensor/Normalize, for `n` windows of one pyramid
 * level that is resident in HBM.  Replaces, per window,
 *   read_region(...).convert("RGB")          src/main.py:693-697
 *   paste on a white PxP canvas              src/main.py:699-703
 *   np.mean(patch) > 240 -> drop             src/main.py:718-720
 *   Resize((224,224)) / ToTensor / Normalize src/main.py:812-818 (Pillow + torch)
 *
 *   level     : uint8 image, `chans` (3 = RGB, 4 = RGBA, alpha ignored) bytes
 *               per pixel, row pitch `pitch` bytes, W x H pixels
 *   xy        : int32[n,2] window origins in level pixels (0 <= x < W, 0 <= y < H;
 *               for P > 224, x must be a multiple of 16 -- the reference's grid
 *               strides by 224, src/main.py:682)
 *   P         : window size 224*s, s in 1..8 (224, 448, 896, 1792 in the reference)
 *   coeff_bounds, coeff_kk, ksize : DEVICE copies of hipac_resample_coeffs(P, 224,
 *               ..., kk_stride = ksize); may be NULL when P == 224 (identity)
 *   lut       : float32[3,256] = (v/255 - mean_c)/std_c evaluated in fp32
 *   out       : per `out_format`
 *   sums      : uint32[n] sum of the padded PxPx3 window (NULL to skip)
 *   keep      : uint8[n] 1 iff sum <= 240*3*P*P, i.e. not (mean > 240) (NULL to skip)
 */
int hipac_tile_preprocess(const uint8_t* level, int W, int H, int64_t pitch, int chans,
                          const int32_t* xy, int n, int P,
                          const int32_t* coeff_bounds, const int32_t* coeff_kk, int ksize,
                          const float* lut,
                          void* out, int out_format, uint32_t* sums, uint8_t* keep,
                          void* stream);

/* ---- whole-level form ("planes") for windows on the reference's 224-pixel lattice ----
 * With origins on multiples of 224 (src/main.py:682-683) and P = 224*s, s in {2,4,8}, every
 * window's resampled pixels are a gather from ONE resampled image of the level, so each
 * source pixel is read once per level instead of once per overlapping window (up to 64x at
 * level 0).  Same arithmetic as hipac_tile_preprocess, bit for bit.
 *   hipac_level_planes_sizes  : bytes of the three caller-owned scratch buffers
 *   hipac_level_build_planes  : level (RGB, 3 B/px, 4-byte aligned base and pitch) -> himg, dimg,
 *                               cells (sums of the 224x224 source cells, for the whiteness test)
 *   hipac_level_window_stats  : per window sum of the padded PxPx3 pixels and keep flag
 *                               (replaces np.mean(patch) > 240, src/main.py:718-720)
 *   hipac_level_gather        : uint8[n,224,224,3] resized pixels of the listed windows
 *                               (HIPAC_OUT_U8_HWC; feed to hipac_resnet18_forward as HIPAC_IN_U8_HWC)
 * xy: int32[n,2] window origins, multiples of 224, x < W, y < H.  coeff_*: device tables as for
 * hipac_tile_preprocess. */
int hipac_level_planes_sizes(int W, int H, int P, size_t* himg_bytes, size_t* dimg_bytes, size_t* cell_bytes);
int hipac_level_build_planes(const uint8_t* level, int W, int H, int64_t pitch, int P,
                             const int32_t* coeff_bounds, const int32_t* coeff_kk, int ksize,
                             void* himg, void* dimg, uint32_t* cells, void* stream);
int hipac_level_window_stats(const uint32_t* cells, int W, int H, int P, const int32_t* xy, int n,
                             uint32_t* sums, uint8_t* keep, void* stream);
int hipac_level_gather(const void* dimg, int W, int H, int P, const int32_t* xy, int n, uint8_t* out,
                       void* stream);

/* Lattice form of hipac_window_labels (origins on multiples of 224): one pass over the mask
 * into per-cell flags, then s x s cells per window.  Same result as hipac_window_labels. */
int hipac_mask_cells(const uint8_t* mask, int W, int H, int64_t pitch, uint8_t* cellany, void* stream);
int hipac_window_labels_cells(const uint8_t* cellany, int W, int H, int P, const int32_t* xy, int n,
                              uint8_t* labels, void* stream);

/* tumour / normal label per window: 1 iff any mask pixel > 0 inside
 * [x,x+P) x [y,y+P) (pixels outside the mask count as 0).
 * Replaces mask.crop(...) / np.any(... > 0), src/main.py:707-712. */
int hipac_window_labels(const uint8_t* mask, int W, int H, int64_t pitch,
                        const int32_t* xy, int n, int P, uint8_t* labels, void* stream);

/* uint8[n,224,224,3] patches (already 224x224, e.g. decoded level-3 PNGs) ->
 * network input in `out_format` (ToTensor + Normalize only; Resize is the
 * identity at 224, src/main.py:814). */
int hipac_patches_normalize(const uint8_t* patches, int n, const float* lut,
                            void* out, int out_format, void* stream);

/* ---------------------------------------------------------------------------
 * MIL head over bags of patch features: attention / mean / max pooling
 * of each bag followed by the two-layer classifier.
 * Replaces MILAttentionPooling.forward and MILClassifier.forward,
 * src/models/mil_classifier.py:12-18 and :38-45, for MANY bags per call (the
 * reference scores one bag per forward).  All pointers are DEVICE pointers; weight
 * matrices are in the PyTorch Linear layout [out][in], float32.
 * ------------------------------------------------------------------------- */
#define HIPAC_MIL_ATTENTION 0
#define HIPAC_MIL_MEAN 1
#define HIPAC_MIL_MAX 2

typedef struct {
  const float* attn_V_w; /* [attn_dim][feature_dim]   aggregator.attn_V.weight (attention only) */
  const float* attn_V_b; /* [attn_dim]                                                      */
  const float* attn_U_w; /* [1][attn_dim]             aggregator.attn_U.weight              */
  const float* attn_U_b; /* [1]                                                             */
  const float* fc1_w;    /* [hidden_dim][feature_dim] classifier.0.weight                   */
  const float* fc1_b;    /* [hidden_dim]                                                    */
  const float* fc2_w;    /* [num_classes][hidden_dim] classifier.2.weight                   */
  const float* fc2_b;    /* [num_classes]                                                   */
  int32_t feature_dim;   /* multiple of 4, <= 2048 (512 for ResNet18 features)              */
  int32_t attn_dim;      /* <= 256 (128 in the reference)                                   */
  int32_t hidden_dim;    /* <= 256 (128 in the reference)                                   */
  int32_t num_classes;   /* <= 16                                                           */
} hipac_mil_params_t;

/* feats [n][feature_dim] float32, rows of one bag contiguous; bag b = rows
 * bag_offsets[b] .. bag_offsets[b+1]-1 (int32[n_bags+1], non-decreasing, every bag
 * non-empty -- the caller checks).  Outputs: logits [n_bags][num_classes];
 * attn [n] (attention pooling only, may be NULL) = softmax over each bag of
 * attn_U(tanh(attn_V(x))); pooled [n_bags][feature_dim] (may be NULL).
 * scores: scratch float[n] (attention only).  Asynchronous on `stream`. */
int hipac_mil_forward(const hipac_mil_params_t* params, int pooling, const float* feats,
                      const int32_t* bag_offsets, int n, int n_bags, float* logits, float* attn,
                      float* pooled, float* scores, void* stream);

/* ---------------------------------------------------------------------------
 * NT-Xent loss of the SimCLR step, value and gradient in one call (SURVEY a-12).
 * Replaces nt_xent_loss, src/models/simclr.py:31-54, and its autograd backward:
 * z = cat(z_i, z_j) float32 [2n][d] (device), loss float32[1] (device),
 * dz float32 [2n][d] = d loss / d z (NULL: value only).  n <= 16384; the scratch holds the [2n][2n] similarity matrix.
 * scratch: hipac_ntxent_scratch_bytes(n, d) bytes of device memory.
 * ------------------------------------------------------------------------- */
size_t hipac_ntxent_scratch_bytes(int n, int d);
int hipac_ntxent_fwd_bwd(const float* z, int n, int d, float temperature, float* loss, float* dz,
                         void* scratch, size_t scratch_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Native training step (SURVEY a-12 / a-13): the ResNet18 encoder in TRAIN mode, forward and backward,
 * fp32 throughout on the exact f32 MFMA.  Replaces what torch autograd executes for
 *   z_i = model(x_i); z_j = model(x_j); loss.backward()      src/models/simclr.py:88-94
 *   outputs = model(imgs); scaler.scale(loss).backward()      src/main.py:499-506, :578-586
 * Batch-norm uses the statistics of the call's own batch (per replica, as nn.DataParallel does: SURVEY F6)
 * and updates the running statistics with `momentum` (torch: 0.1) and the unbiased variance.
 *
 * Parameters live in ONE flat float32 device buffer in a fixed order: 20 convolutions (0 = stem, then per
 * stage block0.conv1, block0.conv2, [block0.downsample], block1.conv1, block1.conv2), each as
 *   weight [Cout][Cin][kh][kw] (PyTorch layout), bn.weight [Cout], bn.bias [Cout];
 * running statistics in a second flat buffer: per conv running_mean [Cout], running_var [Cout];
 * gradients in a buffer shaped like the parameters.  hipac_train_conv_desc gives geometry and offsets.
 * ------------------------------------------------------------------------- */
int hipac_train_num_convs(void);
int hipac_train_conv_desc(int i, int* cout, int* cin, int* ks, int* stride, int64_t* param_off, int64_t* stat_off);
size_t hipac_train_param_floats(void);
size_t hipac_train_stat_floats(void);
/* Bytes of one forward's workspace (input copy, pre-/post-BN maps of every conv, pool arg-max, batch
 * statistics, scratch): ~31 MB per image.  The SAME workspace must be handed to the matching backward. */
size_t hipac_train_workspace_bytes(int batch);
/* Test tap: byte offset inside the workspace of a map the forward keeps for the backward -- kind 0: output of
 * conv `conv` before BN, 1: after BN (+ residual) (+ ReLU), both float32 NHWC [batch][H][W][Cout]; 2: pooled stem
 * map [batch][56][56][64]; 3: batch mean[Cout] then rstd[Cout] of conv `conv`; 4: the max-pool's arg-max bytes
 * [batch][56][56][64] (0..8 = dy * 3 + dx inside the 3x3 window).  -1 on a bad argument. */
int64_t hipac_train_debug_offset(int batch, int kind, int conv);
/* x: float32[batch,3,224,224] NCHW (ImageNet-normalised) -> feats float32[batch,512] (fc = Identity,
 * src/models/simclr.py:19).  stats (running statistics) may be NULL: not updated.  batch <= 4096. */
int hipac_train_encoder_forward(const float* params, float* stats, const float* x, int batch, float momentum,
                                float eps, float* feats, void* workspace, size_t workspace_bytes, void* stream);
/* dfeats: float32[batch,512] = d loss / d feats of the forward that filled `workspace`.  grads: flat buffer,
 * overwritten (accumulate = 0) or added to (accumulate = 1: second view of a SimCLR step). */
int hipac_train_encoder_backward(const float* params, const float* dfeats, int batch, float* grads, int accumulate,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* Mixed-precision form of the two calls above -- what the reference's fine-tune loops run under
 * torch.cuda.amp.autocast() + GradScaler (src/main.py:499-508, :578-587): fp16 operands on the fp16 MFMA with fp32
 * accumulation, fp16 activations and activation gradients, the SAME fp32 flat parameter / gradient / statistics
 * buffers.  `x`, `feats`, `dfeats`, `grads` stay float32; `dfeats` arrives multiplied by the caller's loss scale and
 * `grads` leaves multiplied by it (hipac_grads_unscale_check divides it out and reports inf / nan, as
 * GradScaler.unscale_ does).  Every reduction is two-stage in a fixed order: the same step run twice gives the same
 * bits.  batch <= 2048.  Its own workspace size; the maps inside are fp16 NHWC (hipac_train_amp_debug_offset). */
size_t hipac_train_amp_workspace_bytes(int batch);
int64_t hipac_train_amp_debug_offset(int batch, int kind, int conv);
int hipac_train_amp_encoder_forward(const float* params, float* stats, const float* x, int batch, float momentum,
                                    float eps, float* feats, void* workspace, size_t workspace_bytes, void* stream);
int hipac_train_amp_encoder_backward(const float* params, const float* dfeats, int batch, float* grads, int accumulate,
                                     void* workspace, size_t workspace_bytes, void* stream);
/* grads[i] *= inv_scale; found_inf[0] (device int32, zeroed by the caller) becomes 1 when a gradient is inf / nan. */
int hipac_grads_unscale_check(float* grads, int64_t n, float inv_scale, int32_t* found_inf, void* stream);

/* Baseline-JPEG tiles of a tiled pyramidal TIFF decoded on the device, straight into a level image in HBM -- what
 * openslide.OpenSlide(path) / read_region do with libjpeg on the host for the reference (src/main.py:650, :693).
 * Huffman decoding runs one lane per TILE, then libjpeg's integer ("ISLOW") IDCT, its h2v2 "fancy" chroma upsampling and
 * its fixed-point YCbCr -> RGB: bit-exact against libjpeg / Pillow.  Supported per tile: baseline sequential, 8 bit, 3
 * components in one interleaved scan, 4:2:0, 4:2:2 or 4:4:4, Huffman table ids 0 / 1, JPEG size == tile size, tile sides multiples
 * of 16; every other tile is left untouched and reported in `status_host` so the caller decodes it on the host.
 *   file_host / file_dev : the file's bytes in HOST memory (headers are parsed there) and the same bytes in DEVICE memory
 *                          (16 readable bytes behind the end)
 *   levels               : HOST hipac_jpeg_level[n_levels]: where the tiles of a level go (DEVICE uint8[H][pitch_bytes] RGB,
 *                          W x H pixels; tiles are clipped to it), its tile size, its JPEGTables tag (347; HOST bytes or NULL)
 *                          and its photometric tag (262: 6 = YCbCr, converted to RGB; 2 = RGB samples in the stream)
 *   tile_off, tile_len   : HOST int64[n_tiles] TileOffsets / TileByteCounts (len 0 = missing tile)
 *   tile_xyl             : HOST int32[n_tiles][3]: destination (x, y) of the tile's top-left pixel, index into `levels` --
 *                          tiles of ALL levels go into one call: a lane's walk through its tile is serial, so the time of a
 *                          call is the time of its slowest tile and the levels should share it
 *   workspace            : DEVICE scratch of hipac_jpeg_workspace_bytes(largest tile_w, tile_h, n_tiles) bytes
 *   status_host          : HOST uint8[n_tiles]: 0 decoded here, 1 not supported (decode on the host), 2 missing tile
 * n_tiles <= 65535 per call, n_levels <= 64.  The call returns when its kernels have finished. */
typedef struct {
  uint8_t* pixels;
  int64_t pitch_bytes;
  int32_t W, H, tile_w, tile_h, photometric, reserved;
  const uint8_t* jpeg_tables;
  int64_t jpeg_tables_len;
} hipac_jpeg_level;
size_t hipac_jpeg_workspace_bytes(int tile_w, int tile_h, int n_tiles);
int hipac_jpeg_decode_tiles(const uint8_t* file_host, const uint8_t* file_dev, int64_t file_bytes, const hipac_jpeg_level* levels,
                            int n_levels, const int64_t* tile_off, const int64_t* tile_len, const int32_t* tile_xyl, int n_tiles,
                            void* workspace, size_t workspace_bytes, uint8_t* status_host, void* stream);

/* Training-view augmentation on patches resident in HBM -- replaces, per view, torchvision's PIL pipelines
 *   geometry 0 (SimCLR): RandomResizedCrop(224) / RandomHorizontalFlip / RandomApply([ColorJitter(.4,.4,.4,.1)], .8) /
 *     RandomGrayscale(.2) / ToTensor / Normalize               src/models/simclr.py:57-66, applied twice per patch by
 *     SimCLRDataset.__getitem__                                src/datasets/simclr_dataset.py:8-12
 *   geometry 1 (classifier loops, tumour patches; P must be 224): RandomHorizontalFlip / RandomVerticalFlip /
 *     RandomRotation(90) / ColorJitter(.2,.2,.2,.1) / Resize((224,224)) / ToTensor / Normalize      src/main.py:417-425
 *     (all-default parameters = the eval transform of :426-430 on a 224-pixel patch)
 * The random draws are the caller's (HOST memory, checked here); the kernels are Pillow's arithmetic (ImagingResample 8-bit
 * passes, affine_fixed nearest-neighbour rotation, ImagingBlend, the L conversion, rgb2hsv / hsv2rgb), bit-exact against Pillow.
 *   pool       : DEVICE uint8[n_pool][P][P][3] decoded patches
 *   params_host: HOST int32[n_views][24] per view:
 *                 [0] patch index  [1] crop top  [2] crop left  [3] crop height  [4] crop width (geometry 0)  [5] horizontal flip
 *                 [6..9] colour operations in application order (0 brightness, 1 contrast, 2 saturation, 3 hue, -1 none)
 *                 [10] grayscale  [11] brightness  [12] contrast  [13] saturation factor (float32 bit patterns)
 *                 [14] hue shift added to Pillow's uint8 H channel (0..255)  [15] reserved (0)  [16] vertical flip (geometry 1)
 *                 [17..22] a0 a1 a2 a3 a4 a5 of affine_fixed (16.16): source x = (a2 + x a0 + y a1) >> 16,
 *                 y = (a5 + x a3 + y a4) >> 16 (geometry 1; identity = 65536 0 0 0 65536 0)  [23] reserved (0)
 *   params_dev : DEVICE int32[n_views][24] scratch (the checked copy the kernels read)
 *   tab_bounds : DEVICE int32[P][224][2], tab_kk DEVICE int32[P][224][ksize]: hipac_resample_coeffs(size, 224) for every
 *                source size 1..P (row size-1), taps beyond a row's count zero (geometry 0; may be NULL for geometry 1)
 *   lut        : DEVICE float32[3][256] = (v/255 - mean_c)/std_c in fp32
 *   tmp        : DEVICE uint8[n_views][P][224][3] (geometry 0), crops: DEVICE uint8[n_views][224][224][3] (scratch)
 *   out        : DEVICE float32[n_views][3][224][224] (NCHW, what the training step takes) or NULL
 *   out_u8     : DEVICE uint8[n_views][224][224][3] augmented image before ToTensor, or NULL (test tap)
 * n_views <= 65535. */
int hipac_augment_views(const uint8_t* pool, int64_t n_pool, int P, int geometry, const int32_t* params_host, int32_t* params_dev,
                        int n_views, const int32_t* tab_bounds, const int32_t* tab_kk, int ksize, const float* lut, uint8_t* tmp,
                        uint8_t* crops, float* out, uint8_t* out_u8, void* stream);

/* nn.Linear forward y = x w^T + b (optional ReLU): projector src/models/simclr.py:20-24, fc resnet.py:66. */
int hipac_linear_forward(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int relu,
                         void* stream);
/* its backward: dy [M][N] (masked by y > 0 when y, the ReLU output, is given; dym = scratch [M][N] then),
 * dx [M][K] or NULL, dw [N][K], db [N] or NULL; accumulate = add to dw / db instead of overwriting. */
int hipac_linear_backward(const float* x, const float* w, const float* dy, const float* y, float* dym, float* dx,
                          float* dw, float* db, int M, int N, int K, int accumulate, void* stream);

/* nn.CrossEntropyLoss(weight = class_w) value and gradient (src/main.py:490, :552-566): logits [M][C],
 * labels int64 [M], class_w [C] or NULL, loss float[1], dlogits [M][C], scratch float[2 + 8 * ceil(M / 256)]
 * (per-wave partial sums, added in a fixed order: the loss is reproducible); all device.
 * A label outside [0, C) (torch raises there) is never dereferenced: the loss and that row's gradient come out NaN. */
int hipac_cross_entropy_fwd_bwd(const float* logits, const int64_t* labels, const float* class_w, int M, int C,
                                float* loss, float* dlogits, float* scratch, void* stream);

/* torch.optim.Adam step `step` (1-based), no weight decay (src/main.py:492, src/models/simclr.py:79). */
int hipac_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1,
                    float beta2, float eps, int step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HIPAC_H_ */
