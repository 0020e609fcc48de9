/* dptx.h -- C ABI of libdptx.so: the MI355X (gfx950) DPT-Hybrid-384 inference engine.
 *
 * This is the drop-in boundary for ONE path of EPFL-VILAB/omnidata: the forward pass of
 *   DPTDepthModel(backbone='vitb_rn50_384', num_channels={3|1})
 * (omnidata_tools/torch/modules/midas/dpt_depth.py:87-107, DPT.forward :67-85) that the
 * reference reaches from demo.py:140 (`model(img_tensor)`) and from the torch.hub entry
 * points `surface_normal_dpt_hybrid_384` / `depth_dpt_hybrid_384` (README.md:23-29).
 * The reference has no FFI of its own (it is pure Python on ATen); every entry point
 * below names the reference interface it replaces.  Plain pointers and sizes only; no
 * torch types, no C++ exceptions across the boundary.  All functions return 0 on success
 * or a negative DPTX_E_* code; dptx_last_error() gives the message.
 *
 * Ownership: the caller owns x / y device buffers (e.g. torch tensors' data_ptr());
 * the engine owns its packed weights and activation arena and never frees caller memory.
 * Threading: one handle per (device, stream); a handle is not re-entrant (the reference
 * is not either: vit.py:158 keeps hook outputs in a module-global dict); independent
 * handles may be used from independent threads.
 */
#ifndef DPTX_H_
#define DPTX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dptx_engine* dptx_handle;

enum {
  DPTX_OK = 0,
  DPTX_E_INVALID = -1,   /* bad argument / wrong call order            */
  DPTX_E_KEY = -2,       /* unknown / missing / mis-shaped tensor key  */
  DPTX_E_HIP = -3,       /* a HIP runtime call failed                  */
  DPTX_E_NODEVICE = -4,  /* compute entry point called on a host-only handle */
  DPTX_E_ALLOC = -5
};

/* arithmetic type of the MFMA operands and of the stored activations.
 * BF16X3 is the high-precision mode: every 16-bit tensor is a pair of bf16 planes (hi, lo = x - hi,
 * 16 significand bits) and every product is 3 MFMAs (lo*hi + hi*lo + hi*hi) with fp32 accumulate:
 * ~3x the MFMA work and 2x the activation bytes, meets 1e-3 abs against the fp32 reference. */
enum { DPTX_DTYPE_BF16 = 0, DPTX_DTYPE_FP16 = 1, DPTX_DTYPE_BF16X3 = 2,
       /* FP16X3: the same hi/lo scheme with fp16 planes (hi = fp16(x) is also a valid single-pass fp16 operand).
        * MIXED : fp16 planes; the layer groups named in dptx_config.x3_groups run with 3 MFMAs per product, the others
        *         single-pass fp16 on the hi plane -- a per-layer precision policy (profiles/r02_precision_frontier.md). */
       DPTX_DTYPE_FP16X3 = 3, DPTX_DTYPE_MIXED = 4,
       /* FP8 (BASELINE.json configs[4] "fp8 MFMA weights"): a bf16 engine whose decoder convolutions -- the RCU 3x3 convs,
        *         out_conv and the first head conv: 31 % of a single-task forward's MACs, 43 % of the dual-task one's --
        *         run on OCP e4m3 operands (weights quantised at load time after a per-layer power-of-two scale, activations
        *         quantised by their producers' epilogues) on v_mfma_scale_f32_32x32x64_f8f6f4 at unit block scale, fp32
        *         accumulate.  Half the operand bytes per flop and twice the MFMA rate of bf16; per-conv error ~3 % rms
        *         (4-bit significand) -- a throughput mode with its own stated tolerance, NOT a parity mode
        *         (tests/test_gpu_fp8.py, profiles/r02_precision_frontier.md). */
       DPTX_DTYPE_FP8 = 5 };
/* layer groups of the forward for dptx_config.x3_groups (dtype = DPTX_DTYPE_MIXED).  A 3-MFMA group may only read
 * tensors produced by 3-MFMA groups (its lo planes must exist): HEAD needs FUSION needs RN needs RESNET and REASSEMBLE;
 * EMBED needs RESNET; the 12 ViT blocks exchange only the fp32 token stream and are free.  dptx_create rejects others. */
enum { DPTX_GROUP_RESNET = 1,      /* stem + ResNetV2 stages (convs and GroupNorms)            */
       DPTX_GROUP_EMBED = 2,       /* patch_embed.proj                                         */
       DPTX_GROUP_VIT = 4,         /* the 12 transformer blocks (LN, qkv, attention, proj, MLP) */
       DPTX_GROUP_REASSEMBLE = 8,  /* ProjectReadout + act_postprocess3/4 convs                */
       DPTX_GROUP_RN = 16,         /* scratch.layerN_rn                                        */
       DPTX_GROUP_FUSION = 32,     /* scratch.refinenet4..1                                    */
       DPTX_GROUP_HEAD = 64,       /* scratch.output_conv                                      */
       DPTX_GROUP_ALL = 127 };
/* backbones behind the same ABI (dpt_depth.py:27-35 `backbone=`) */
enum { DPTX_BACKBONE_VITB_RN50_384 = 0, DPTX_BACKBONE_VITL16_384 = 1 };
/* element type of the caller-side image AND result buffers of one forward call (`x_dtype`): fp32 is the drop-in default
 * (the reference feeds and returns fp32 tensors); with BF16 / FP16 the stem reads and the head writes 16-bit NCHW
 * tensors directly (SURVEY.md 8d config 2 feeds bf16) -- half the bytes at the boundary, no conversion pass. */
enum { DPTX_IO_FP32 = 0, DPTX_IO_BF16 = 1, DPTX_IO_FP16 = 2 };

typedef struct dptx_config {
  int32_t num_channels;  /* 3 = surface normals, 1 = depth (dpt_depth.py:88 num_channels)      */
  int32_t max_batch;     /* arena is sized for this many max_height x max_width images per call (1..48) */
  int32_t dtype;         /* DPTX_DTYPE_*                                                        */
  int32_t device_id;     /* HIP device ordinal; -1 = host-only handle (weight packing only)     */
  int32_t non_negative;  /* final ReLU of the head (dpt_depth.py:88,98 non_negative=True)       */
  int32_t ws_form;       /* 0: (w-mean)/(std+eps) timm 0.4.x;  1: (w-mean)/sqrt(var+eps)        */
  float   ws_eps;        /* StdConv2dSame eps (timm vit_base_r50_s16: 1e-8)                     */
  int32_t max_height;    /* largest input the arena is planned for; 0 = 384. Multiples of 32,   */
  int32_t max_width;     /*   >= 64, and max_batch*max_height*max_width*256 < 2^31 (see dptx_forward_hw) */
  int32_t dual_task;     /* 1: two decoders on one shared encoder (see dptx_forward_dual); needs num_channels = 3 */
  int32_t streams;       /* n = 2..4: a forward of >= 2 images runs as n sub-batches on n internal streams, forked   */
                         /*   from / joined to the caller's stream (same bits); 1: caller's stream only; 0 (default):  */
                         /*   MEASURED choice between 1 and 2 -- one stream until dptx_tune_schedule has timed both    */
  int32_t x3_groups;     /* dtype MIXED: OR of DPTX_GROUP_* that run with 3 MFMAs per product; 0 = the default policy  */
                         /*   (everything except the ViT blocks).  Ignored by the other dtypes.                      */
  int32_t backbone;      /* DPTX_BACKBONE_*: 0 = vitb_rn50_384 (DPT-Hybrid, the default), 1 = vitl16_384 (DPT-Large:  */
                         /*   dpt_depth.py:41-45 hooks [5,11,17,23], blocks.py:12-18, vit.py:176-309; demo.py:81)       */
  int32_t flags;         /* OR of DPTX_FLAG_*: switches for A/B runs of the fused schedules (0 = everything on)  */
  int32_t reserved;      /* must be zero                                                        */
} dptx_config;
/* dptx_config.flags.  NO_LN_FOLD: keep the 24 LayerNorm launches of the ViT blocks instead of folding the LayerNorm into
 * the qkv / fc1 GEMMs (gamma into W, W beta into the bias, (x W' - mu colsum(W')) rstd in the epilogue; row statistics and
 * the 16-bit operand copy of the fp32 token stream come out of the preceding proj / fc2 / patch-embed epilogue).  The fold
 * applies to single-pass ViT blocks only (bf16, fp16, fp8, and mixed policies without DPTX_GROUP_VIT).
 * GROUP_POLICY (dtype MIXED): do not install the default per-LAYER table of the decoder (see dptx_set_layer_precision);
 * every layer then follows its group's bit in x3_groups (round 2's policy: 2.14x the MFMA work of single-pass instead of
 * 1.66x, 1.5-2x smaller deviation from the fp32 forward). */
/* FP32_STREAM: keep the fp32 copy of the ViT token stream in the single-pass dtypes (BF16 / FP16 / FP8).  By default these
 * dtypes, with the LayerNorm fold, carry the residual stream of the 12 blocks only as the 16-bit tensor that the qkv / fc1
 * GEMMs multiply (what the reference itself does under model.half() / .bfloat16()): the proj / fc2 epilogues move 4 instead
 * of 10 bytes per element; the deviation from the fp32 forward grows by 1-5 % of itself (profiles/r03_experiments.md).
 * MIXED and the 3-MFMA dtypes always keep the fp32 stream. */
/* NO_RANGE_CHECK: skip the per-forward range scan of the fp16-plane dtypes (see dptx_range_status). */
/* FP8_ALL (dtype FP8): run all 19 eligible decoder convolutions (14 RCU 3x3, 4 out_conv, output_conv.0) on e4m3 operands --
 * round 3's mode: 7.5 - 9 degrees of mean angular error on the synthetic weight families, a lossy throughput mode.  Without the
 * flag only the six resConfUnit1 convolutions of refinenet1..3 do (oracle/fp8_layers.py: the set that keeps the mode within
 * 2 x the bf16 engine's error on both families). */
/* FP8_VIT (dtype FP8, round 6): qkv / fc1 / fc2 of every transformer block on e4m3 operands too (45 of the 127.6 GMAC; weights per
 * output channel, ONE calibrated power-of-two scale per activation tensor -- the token stream after proj / fc2 and the GELU
 * output get e4m3 copies from the producing epilogues; proj stays bf16).  Needs the LayerNorm fold and the 16-bit token stream
 * (the defaults).  oracle/fp8_vit.py: +2.1-2.7 / +1.2-1.9 degrees of mean angular error on the two synthetic weight families when
 * taken alone; on the GPU, together with the default decoder preset: 4.62 / 1.98 degrees against the bf16 engine's 4.18 / 1.07 --
 * inside the default preset's own bar (<= 2 x bf16 on both families, tests/test_gpu_fp8.py).  Not a parity mode. */
enum { DPTX_FLAG_NO_LN_FOLD = 1, DPTX_FLAG_GROUP_POLICY = 2, DPTX_FLAG_FP32_STREAM = 4, DPTX_FLAG_NO_RANGE_CHECK = 8, DPTX_FLAG_FP8_ALL = 16,
       DPTX_FLAG_FP8_VIT = 32 };

/* Fills *cfg with the reference defaults: C=3, max_batch=32, dtype MIXED (the mode that matches the reference's fp32
 * forward within 1e-3; DPTX_DTYPE_BF16 is the ~1.6x faster throughput mode that does not), device 0, non_negative=1,
 * ws_form=0, ws_eps=1e-8. */
void dptx_default_config(dptx_config* cfg);

/* Replaces the constructor DPTDepthModel(...) (dpt_depth.py:87-104). */
int dptx_create(dptx_handle* out, const dptx_config* cfg);
void dptx_destroy(dptx_handle h);

/* Replaces `model.load_state_dict(state_dict)` (demo.py:72; BaseModel.load base_model.py:4-16),
 * one tensor at a time.  `ref_key` is the reference state_dict key after the Lightning
 * `model.` prefix has been stripped (demo.py:65-70), e.g.
 * "pretrained.model.blocks.0.attn.qkv.weight"; `host_fp32` is a contiguous fp32 host array of
 * the given shape (OIHW for convs, [out,in] for linears).  Keys the forward never reads
 * (pretrained.model.norm.*, pretrained.model.head.*, scratch.refinenet4.resConfUnit1.*) are
 * accepted and ignored.  Unknown keys or wrong shapes -> DPTX_E_KEY. */
int dptx_load_tensor(dptx_handle h, const char* ref_key, const float* host_fp32,
                     const int64_t* shape, int32_t ndim);

/* Strict check + packing: every tensor the forward reads must have been loaded (else DPTX_E_KEY,
 * the message lists the missing keys -- mirrors load_state_dict(strict=True)).  Folds the
 * StdConv2dSame weight standardisation (input independent) into the conv weights, re-lays
 * OIHW -> [O][kh][kw][I], converts GEMM operands to cfg.dtype and builds one contiguous blob.
 * On a device handle the blob is uploaded and the activation arena is allocated. */
int dptx_finalize_weights(dptx_handle h);

/* Size of the packed blob (valid after finalize, or for any handle: depends only on cfg). */
size_t dptx_packed_bytes(dptx_handle h);
/* Copies the packed blob to host memory (works on host-only handles: used by CPU tests). */
int dptx_export_packed_host(dptx_handle h, void* dst_host, size_t bytes);
/* Copies the packed blob to / from a caller DEVICE buffer on `stream`.  This is the multi-GPU
 * start-up path: rank 0 finalizes, exports into a torch uint8 tensor, torch.distributed
 * (RCCL) broadcasts it over xGMI, the other ranks import it instead of packing themselves. */
int dptx_export_packed_device(dptx_handle h, void* dst_dev, size_t bytes, void* stream);
int dptx_import_packed_device(dptx_handle h, const void* src_dev, size_t bytes, void* stream);
/* Round 5: `dst` uses `src`'s packed weights IN PLACE (no copy): both handles must live on the same device and have been created
 * with configurations that pack the same blob (same layout header, same size); `src` must have its weights on the device.  The
 * allocation is reference-counted: the handles may be destroyed in any order (the last one frees it), and a handle that loads or
 * imports weights while others read its blob writes a fresh allocation instead -- sharing is a snapshot, never an alias of a
 * blob that changes under a reader.  A handle is not
 * re-entrant, so several forwards in flight on one GPU need several handles (each with its own activation arena, on its own
 * stream): this lets them read ONE copy of the 244 MB of weights out of L2 / Infinity Cache instead of one copy each
 * (omnidata_amd/pipeline.py ForwardPipeline: two batch-32 forwards in flight, +8 % images/s over the two-halves-of-one-forward
 * schedule -- the ResNetV2 stages of one forward then run under the MFMA-bound ViT / decoder launches of the other).
 * The packed blob is read-only during forwards. */
int dptx_share_packed(dptx_handle dst, dptx_handle src);

/* Bytes of device memory held by the handle (packed weights + activation arena). */
size_t dptx_workspace_bytes(dptx_handle h);

/* Replaces `DPTDepthModel.forward(x)` (dpt_depth.py:106-107 -> DPT.forward :67-85).
 *   x_dev : [batch,3,384,384] contiguous NCHW of x_dtype (DPTX_IO_FP32 / _BF16 / _FP16); normal model expects
 *           values in [0,1], depth model in [-1,1] (omnidata_tools/torch/README.md:46,49).
 *   y_dev : [batch,C,384,384] contiguous NCHW of the same x_dtype (for C=1 this is bit-identical to the
 *           reference's squeezed [batch,384,384]); >= 0 when non_negative, NOT clamped to 1
 *           (callers clamp: demo.py:140).
 * Asynchronous on `stream` (a hipStream_t; NULL = default stream). 1 <= batch <= max_batch. */
int dptx_forward(dptx_handle h, const void* x_dev, int32_t x_dtype, void* y_dev,
                 int32_t batch, void* stream);

/* Same forward at another input size: what the reference does when DPTDepthModel is fed a tensor
 * that is not 384x384 -- forward_flex (vit.py:119-155) resamples pos_embed to the (height/16, width/16)
 * patch grid with _resize_pos_embed (vit.py:102-116, bilinear, align_corners=False) and everything else
 * is convolutional.  height, width: multiples of 32 (the reference's own constraint: the 1/32-scale map is
 * up-sampled x2 and added to the 1/16-scale map), >= 64, height*width <= max_height*max_width of the config.
 *   x_dev [batch,3,height,width] NCHW fp32  ->  y_dev [batch,C,height,width] NCHW fp32.
 * dptx_forward(...) == dptx_forward_hw(..., 384, 384, ...). */
int dptx_forward_hw(dptx_handle h, const void* x_dev, int32_t x_dtype, void* y_dev,
                    int32_t batch, int32_t height, int32_t width, void* stream);

/* Dual-task forward (BASELINE.json configs[4]): ONE encoder pass (`pretrained.*`: ResNetV2 stem and
 * stages, ViT blocks, read-outs, DPT.forward dpt_depth.py:71) feeds TWO decoders -- `scratch.*` (surface normals, 3
 * channels) and `depth.scratch.*` (depth, 1 channel; same layer names as dpt_depth.py:73-83 behind the "depth." prefix).
 * 185.29 GMAC per image instead of 2 x 127.62.  This is a composition the reference does not ship (its two checkpoints
 * are separately fine-tuned full models); parity is defined against the reference forward run twice with `pretrained.*`
 * tied.  Needs a handle created with dual_task = 1, num_channels = 3; both heads see the same input tensor.
 *   y_normal_dev [batch,3,height,width], y_depth_dev [batch,1,height,width], NCHW fp32.
 * dptx_tap after a dual forward: encoder taps as usual, the depth decoder's under "depth.<name>"; the normal decoder's
 * are not available (its buffers were re-used). */
int dptx_forward_dual(dptx_handle h, const void* x_dev, int32_t x_dtype, void* y_normal_dev, void* y_depth_dev,
                      int32_t batch, int32_t height, int32_t width, void* stream);

/* dtype MIXED: per-layer precision inside the decoder (scratch.layerN_rn, refinenet*, output_conv.0 / .2).  `mfmas` = 1: the
 * convolution multiplies the hi planes only (one MFMA per product); 3: hi/lo planes, three MFMAs; 2: hi/lo planes of the weights,
 * hi plane of the activations (a_hi w_hi + a_hi w_lo: the input is rounded to fp16 once, the weights are exact to 22 bits --
 * half of the layer's rounding variance for two thirds of the 3-MFMA cost).  Layers that were never set follow their group's
 * bit in dptx_config.x3_groups.  Tensors between layers carry a lo plane exactly where a 3-MFMA consumer reads it, so every
 * assignment is valid.  With x3_groups = 0 dptx_create installs the default table (oracle/precision_layers.py);
 * DPTX_FLAG_GROUP_POLICY suppresses it.  (Measured option outside the default: "scratch.output_conv.0.weight" = 2 is +3.9 %
 * throughput for a worst-case deviation of 7.6e-4 instead of 6.3e-4 over a 32-image batch.)  May be called at any time before a forward; does not
 * touch the packed weights.  The reference has no counterpart (it computes in fp32 throughout). */
int dptx_set_layer_precision(dptx_handle h, const char* conv_weight_key, int32_t mfmas);

/* fp8 dtype: activation scales.  Every tensor that has an e4m3 copy (the inputs of the decoder's fp8 convolutions) is
 * quantised as e4m3(x * s) with a per-tensor power-of-two scale s; e4m3 covers 2^-9 .. 448, so with s = 1 (a fresh handle)
 * activations beyond 448 saturate and activations below 2^-9 vanish.  dptx_calibrate_fp8 runs ONE forward of the given
 * batch with the decoder convolutions on their bf16 operands, measures max |x| of every such tensor and sets s so that
 * the maximum lands in (112, 224] (one binade of headroom); the results y (and y2 for a dual-task handle, else NULL) are
 * the bf16-decoder results of that batch.  Call it once after loading weights, on a representative batch, before the
 * first dptx_forward.  The reference has no counterpart (it has no fp8 path); weights are quantised per output channel
 * at dptx_finalize_weights. */
int dptx_calibrate_fp8(dptx_handle h, const void* x_dev, int32_t x_dtype, void* y_dev, void* y2_dev,
                       int32_t batch, int32_t height, int32_t width, void* stream);
/* Number of e4m3 tensors of the last forward (<= 128); copies their scales / calibration max |x| (either may be NULL).
 * dptx_fp8_set_calibration installs scales measured elsewhere (rank 0 calibrates, the others receive: positive powers of
 * two, slot order = launch order of the forward). */
int dptx_fp8_get_calibration(dptx_handle h, float* scales, float* amax, int32_t capacity);
int dptx_fp8_set_calibration(dptx_handle h, const float* scales, int32_t n);

/* Range check of the fp16-plane dtypes (FP16, FP16X3, MIXED): fp16 planes cannot hold |x| > 65504 and nothing in the forward
 * clamps.  Every forward of such a handle scans the first head convolution's output -- the tensor that every decoder path
 * and, through them, the ViT blocks reach by residual additions -- for Inf / NaN (one 9.4 MB/image read, ~0.3 % of the
 * forward) and ORs the finding into a STICKY device flag.  dptx_range_status waits for `stream`, stores the flag in
 * *nonfinite (0 / 1), and clears it when reset != 0.  The check reads the activations, not the result: ReLUs behind the
 * scanned tensor turn a NaN into 0, so an overflow can leave a finite, wrong output.  Always 0 for the bf16-plane dtypes
 * (fp32's exponent range) and with DPTX_FLAG_NO_RANGE_CHECK.  omnidata_amd.model reads it after the first forward of a
 * set of weights and periodically afterwards, and falls back to bf16 planes (BF16X3 / BF16) when it is set. */
int dptx_range_status(dptx_handle h, int32_t* nonfinite, int32_t reset, void* stream);

/* Debug hook for stage-level parity (SURVEY.md A.1 tap names: "stem","s0","s1","s2","tok0",
 * "blk0".."blk11","l3","l4","l1_rn".."l4_rn","p4","p3","p2","p1","h0","h1").  Copies the
 * stage activation of the LAST forward, converted to fp32 in the engine's internal layout
 * (NHWC for feature maps, [batch*S,768] for tokens; S = (H/16)*(W/16)+1 = 577 at 384x384), to dst_host.  *shape4
 * receives {batch, H, W, C} (or {batch,S,768,1}).  Returns DPTX_E_KEY for unknown names.  Taps make the forward run
 * as one pass over the whole batch on the caller's stream and keep the three-launch head tail (so "h1" exists). */
int dptx_tap(dptx_handle h, const char* name, float* dst_host, size_t capacity_floats,
             int64_t shape4[4]);
/* The fp32 token stream is updated in place by the 12 blocks; "tok0"/"blkN" taps therefore need
 * copies.  on=1 allocates 13 snapshots and makes dptx_forward record them (debug only). */
int dptx_enable_taps(dptx_handle h, int on);

/* Number of kernel launches issued by one dptx_forward and algorithmic vs executed MACs
 * per image (algorithmic 127.624e9 for C=3; executed differs because out_conv
 * is commuted in front of the x2 upsample). */
int dptx_forward_info(dptx_handle h, int64_t* launches, double* algorithmic_macs,
                      double* executed_macs);

/* Intra-forward schedule of a handle created with streams = 0 (round 6).  Whether two half-batches on two internal streams
 * beat one whole-batch run depends on what the runtime does with the two streams on THIS box in THIS process (hardware-queue
 * sharing, priorities: measured from +6 % to -25 %, profiles/r05_experiments.md, r06_experiments.md) -- so it is measured,
 * not assumed: dptx_tune_schedule runs the forward of this very batch `reps` times per schedule (after one untimed run each),
 * timed with HIP events on `stream`, keeps the two-stream schedule only if it is at least 3 % faster, and blocks until done.
 * The result of either schedule is the same bits.  dptx_schedule_info reports the decision: *split = 1 two half-batches,
 * 0 one stream; *tuned = 0 while nothing has been measured (one stream); ms_* = the measured times (0 before).
 * Handles with streams >= 1 (or DPTX_STREAMS set) keep the schedule they were told: tune returns DPTX_OK without measuring.
 * y_depth_dev: dual-task handles only (else NULL). */
int dptx_tune_schedule(dptx_handle h, const void* x_dev, int32_t x_dtype, void* y_dev, void* y_depth_dev, int32_t batch,
                       int32_t height, int32_t width, int32_t reps, void* stream);
int dptx_schedule_info(dptx_handle h, int32_t* split, int32_t* tuned, float* ms_single, float* ms_split);

/* Do two streams of `device_id` actually run concurrently?  Work on two streams that the runtime has mapped onto ONE hardware
 * queue executes in order (profiles/r05_experiments.md: 1974 instead of 2600 images/s), and nothing in the API says so.  The
 * probe launches a sleeping one-wave kernel of `spin_us` microseconds (0 = 2000) on stream_a alone, then on both streams at
 * once, host-timed: *ratio = t_both / t_alone -- ~1.0 concurrent, ~2.0 serialised.  Blocks (two stream synchronisations per
 * measurement); both streams must be idle-able.  omnidata_amd/pipeline.py checks its slot streams with it. */
int dptx_probe_stream_overlap(int32_t device_id, void* stream_a, void* stream_b, int32_t spin_us, float* ratio);

/* Per-launch timing of the NEXT forwards with one HIP event after every launch on the forward's
 * stream (kernels of one forward are serialized, so consecutive events bracket one kernel plus
 * its launch gap).  dptx_profile_get sums the last forward by category:
 * 0 = implicit-GEMM MFMA kernel, 1 = attention, 2 = LayerNorm/GroupNorm, 3 = other glue. */
int dptx_set_profiling(dptx_handle h, int on);
int dptx_profile_get(dptx_handle h, int32_t category, double* ms, int64_t* launches,
                     double* macs_per_image);
/* Writes "idx,category,name,ms" for every launch of the last profiled forward. */
int dptx_profile_dump(dptx_handle h, const char* path);

const char* dptx_last_error(dptx_handle h);
const char* dptx_version(void);

/* ---- GPU-side pre/post-processing of demo.py (no handle needed; all pointers are DEVICE pointers) ----
 * dptx_preprocess_u8 replaces demo.py:74-76 / 92-95 / 130-138 for RGB (C=3) or greyscale (C=1) uint8 HWC images:
 * Resize(384, BILINEAR, shorter side) with Pillow's antialiased 8-bit two-pass resampler (bit-identical),
 * CenterCrop(384), ToTensor (/255), optional Normalize(0.5,0.5) (depth), 1->3 channel repeat.
 * x_dev: [3,384,384] fp32.  Both image sides must be >= 1 and the resized image >= 384x384 (always true). */
int dptx_preprocess_u8(const void* img_dev, int32_t H, int32_t W, int32_t C, int32_t row_stride_bytes,
                       int32_t depth_normalize, void* x_dev, void* stream);
/* demo.py:140,150: clamp(0,1) -> *255 -> uint8 (truncation), [3,384,384] fp32 -> [384,384,3] uint8. */
int dptx_postprocess_normal_u8(const void* y_dev, void* rgb_u8_dev, void* stream);
/* demo.py:143-145: bicubic 384->512 (align_corners=False), clamp(0,1), 1-x; [384,384] -> [512,512] fp32. */
int dptx_postprocess_depth(const void* y_dev, void* out512_dev, void* stream);
/* Host-side helper (exposed for tests): Pillow's fixed-point bilinear resampling coefficients for one axis.
 * bounds: [out_size][2] (first tap, tap count); kk: [out_size][*ksize] 22-bit fixed-point weights. */
int dptx_resample_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* kk, int32_t kk_capacity,
                         int32_t* ksize);

/* ---- op-level entry points (unit tests + micro-benchmarks of the individual kernels) ----
 * dtype: DPTX_DTYPE_*.  All pointers are device pointers; row-major / NHWC. */

/* For dtype = DPTX_DTYPE_BF16X3 the op entry points read/write hi/lo plane pairs: the lo plane of
 * every activation (weight) tensor lies act_plane_elems (w_plane_elems) 16-bit elements after its
 * hi plane.  Process-global, test use only. */
int dptx_op_set_planes(int64_t act_plane_elems, int64_t w_plane_elems);

/* C[M,N] = act(A[M,K] * W[N,K]^T + bias) (+R); A,W,C,R 16-bit `dtype`; bias fp32 or NULL;
 * act: 0 none, 1 relu, 2 gelu(erf); c_fp32/r_fp32 select fp32 C / R. */
int dptx_op_gemm(int32_t dtype, const void* A, const void* W, const float* bias, const void* R,
                 void* C, int32_t M, int32_t N, int32_t K, int32_t act, int32_t a_fp32,
                 int32_t c_fp32, int32_t r_fp32, void* stream);
/* NHWC conv as implicit GEMM: X[B,H,W,Cin], Wt[Cout][k][k][Cin], Y[B,Ho,Wo,Cout]. */
int dptx_op_conv(int32_t dtype, const void* X, const void* Wt, const float* bias, const void* R,
                 void* Y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                 int32_t ksize, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho,
                 int32_t Wo, int32_t a_relu, int32_t act, void* stream);
/* the same convolution with the plane switches of the MIXED dtype's per-layer policy (kernels.h GemmParams::epi2 /
 * c_hi_only / r1_hi_only / a_hi_only): epi2 = 1 with dtype FP16 multiplies the hi planes only but reads R and writes Y as hi/lo
 * pairs; epi2 = 2 with dtype FP16X3 is the 2-MFMA form (both planes of Wt, the hi plane of X only) */
int dptx_op_conv_planes(int32_t dtype, const void* X, const void* Wt, const float* bias, const void* R, void* Y, int32_t B,
                        int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad_t,
                        int32_t pad_l, int32_t Ho, int32_t Wo, int32_t a_relu, int32_t act, int32_t epi2, int32_t c_hi_only,
                        int32_t r1_hi_only, void* stream);
/* Fused stem: x NCHW fp32 [B,3,H,W] -> y NHWC 16-bit [B,H/2,W/2,64] = conv 7x7 stride 2 with TF-SAME padding;
 * Wt [64][176] 16-bit with k = (c*7 + ky)*8 + kx (kx = 7 and k >= 168 are zero).  H % 8 == 0, W % 128 == 0. */
int dptx_op_stem_conv(int32_t dtype, const float* x, const void* Wt, void* y, int32_t B, int32_t H, int32_t W,
                      void* stream);
/* qkv[B*S,3*H*64] packed (which, head, dim) -> out[B*S,H*64]; softmax(q k^T / 8) v. */
int dptx_op_attention(int32_t dtype, const void* qkv, void* out, int32_t B, int32_t S,
                      int32_t heads, void* stream);
/* y16[M,768] = LayerNorm(x32[M,768]; gamma, beta, eps) */
int dptx_op_layernorm(int32_t dtype, const float* x, const float* gamma, const float* beta,
                      void* y, int32_t M, int32_t C, float eps, void* stream);
/* GroupNorm(32) (+ optional residual R, + optional ReLU) on NHWC 16-bit, out of place.  scratch_f32 receives the
 * per-block partial sums: B * ceil(HW / pix) * 64 floats with pix = clamp(16384 / C, 16, 256). */
int dptx_op_groupnorm(int32_t dtype, const void* X, const float* gamma, const float* beta,
                      const void* R, void* Y, int32_t B, int32_t HW, int32_t C, int32_t relu,
                      float eps, void* scratch_f32, void* stream);
/* Debug: s_memtime stamps of the GEMM k-loop (block 0, lane 0 of each wave; [wave][64 k-tiles][4 phases] int64) into a
 * device buffer of 8*64*4 int64 for every following GEMM launch; NULL switches it off (tools/gpu/gemm_trace.py). */
int dptx_debug_set_trace(void* dev_buf);
/* Debug / tests (tests/test_gpu_poison.py): the activation arena of a handle.  A forward must not read an arena byte it
 * has not written itself -- dptx_debug_arena_fill sets every byte of the arena (all planes) to byte_value (0xFF: NaN in
 * every element type used) after a device synchronisation; the next forward's result must not change.
 * dptx_debug_arena_read copies a byte range to the host; dptx_debug_arena_layout writes "key value" / "buf name off bytes
 * off2" lines (off: whole-batch plan, off2: inside a sub-batch region of the multi-stream plan) and returns the size needed. */
int dptx_debug_arena_fill(dptx_handle h, int32_t byte_value);
int dptx_debug_arena_read(dptx_handle h, void* dst_host, size_t offset, size_t bytes);
int dptx_debug_arena_layout(dptx_handle h, char* dst, size_t capacity);
/* One 64-bit word sum per arena buffer, sub-batch region and plane of the layout the LAST forward used, computed on `stream`
 * behind that forward: out_dev[(plane * regions + region) * nbuf + buf] (uint64, device memory), buffers in the order of
 * dptx_debug_arena_layout.  Returns the number of sums; with out_dev = NULL the capacity needed.  Two forwards of one input
 * must give the same vector; the first entry that differs names the tensor (tools/gpu/r4_hunt.py). */
int dptx_debug_arena_checksums(dptx_handle h, void* out_dev, int32_t capacity, void* stream);
/* Word sums of the ViT buffers {lnst, Hn, QKV, AO, F1} after every launch of the ViT blocks of the following single-stream
 * forwards: dev_buf[launch * 5 + buffer] (uint64 device memory, capacity entries; launch 0 = after the cls rows, then qkv /
 * attention / proj / fc1 / fc2 per block).  NULL switches it off. */
int dptx_debug_set_launch_sums(dptx_handle h, void* dev_buf, int32_t capacity);
/* Debug / tests: switches (per calling host thread) of the 256x256 GEMM kernel's launch form -- 1: staged epilogue instead of the
 * register-direct one, 2: one block per tile instead of the persistent tile loop, 4: the lockstep loop instead of the ping-pong
 * schedule in the two-plane 128x128 kernel.  Results do not depend on them. */
int dptx_debug_set_gemm_flags(int32_t flags);
/* NHWC conv on OCP e4m3 operands (the fp8 dtype's convolution): X8[B,H,W,Cin] and Wt8[Cout][k][k][Cin] are e4m3 bytes
 * (Cin % 128 == 0), fp32 accumulate on the block-scaled fp8 MFMA at unit scale; Y = act(out_scale * conv + bias) (+R) in
 * bf16; Y8 (optional) receives the e4m3 copy of Y (ReLU'd first when q_relu). */
int dptx_op_conv_fp8(const void* X8, const void* Wt8, const float* bias, const void* R, void* Y, void* Y8, int32_t B,
                     int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad_t,
                     int32_t pad_l, int32_t Ho, int32_t Wo, int32_t act, int32_t q_relu, float out_scale, void* stream);
/* Bias-free convolution followed by GroupNorm(32) (+ residual R, + ReLU) the way the ResNetV2 stages run it: the
 * statistics come out of the conv's GEMM epilogue (fp32 accumulators, one record per 32-row block and group, fixed
 * order), the apply pass normalises the stored 16-bit map.  Yraw receives the conv output, Y the normalised result
 * (Y == Yraw allowed).  Ho*Wo % 32 == 0, Cout % 64 == 0; scratch_f32: B * (Ho*Wo/32) * 64 floats. */
int dptx_op_conv_groupnorm(int32_t dtype, const void* X, const void* Wt, void* Yraw, const float* gamma, const float* beta,
                           const void* R, void* Y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                           int32_t ksize, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo,
                           int32_t relu, float eps, void* scratch_f32, void* stream);
/* bilinear x2, align_corners=True, NHWC 16-bit. */
int dptx_op_upsample2x(int32_t dtype, const void* X, void* Y, int32_t B, int32_t H, int32_t W,
                       int32_t C, void* stream);
/* Dense GEMM with the consumer epilogue of the LayerNorm fold (what the qkv / fc1 launches run): C[M,N] (16-bit) =
 * act((A[M,K] W[N,K]^T - mu colsum) rstd + bias), with (mu, rstd) of row m combined from the (sum, sum of squares) records
 * ln_stats[m][0 .. ln_nblk) (float2, row stride 8 records; ln_nblk = K / 128 = 6 or 8) and ln_colsum[n] = sum_k W[n][k]. */
int dptx_op_gemm_ln(int32_t dtype, const void* A, const void* W, const float* bias, void* C, int32_t M, int32_t N, int32_t K,
                    int32_t act, const float* ln_stats, const float* ln_colsum, int32_t ln_nblk, float ln_eps, void* stream);
/* Dense GEMM with the PRODUCER epilogue of the LayerNorm fold on the 16-bit token stream (what the proj / fc2 launches run,
 * vit.py:150-151 x = x + attn(...) / x + mlp(...)): C[M,N] (16-bit) <- C + A[M,K] W[N,K]^T + bias in place, and the (sum, sum of
 * squares) of every new row per 128-column block as float2 records row_stats[m][0 .. N / 128) (row stride 8 records; N a
 * multiple of 128, <= 1024). */
int dptx_op_gemm_stream(int32_t dtype, const void* A, const void* W, const float* bias, void* C, float* row_stats, int32_t M,
                        int32_t N, int32_t K, void* stream);
/* The same on the fp32 token stream (the parity mode's ViT blocks): X[M,N] (fp32) <- X + A W^T + bias in place, the 16-bit image
 * of the new rows into C16 (what the next qkv / fc1 GEMM multiplies), records as above. */
int dptx_op_gemm_stream32(int32_t dtype, const void* A, const void* W, const float* bias, float* X, void* C16, float* row_stats,
                          int32_t M, int32_t N, int32_t K, void* stream);
/* Fused tail of the head (dpt_depth.py:93-98): Interpolate(x2, bilinear, align_corners=True) -> Conv2d(128,32,3,pad 1)
 * -> ReLU -> Conv2d(32,C,1) -> ReLU(if relu_out).  H0 NHWC 16-bit [B,Hs,Ws,128]; W2 16-bit [32][3][3][128]
 * (O,kh,kw,I); b2 fp32[32]; w4 fp32 [C][32]; b4 fp32[C]; y NCHW fp32 [B,C,2Hs,2Ws].  BF16 / FP16 only; C <= 3. */
int dptx_op_head_tail(int32_t dtype, const void* H0, const void* W2, const float* b2, const float* w4,
                      const float* b4, float* y, int32_t B, int32_t Hs, int32_t Ws, int32_t C,
                      int32_t relu_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DPTX_H_ */
