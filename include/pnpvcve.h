/* libpnpvcve_hip.so -- C ABI of the MI355X-native PnP-VCVE BAE/CAA forward hot path.
 *
 * The reference (ZeldaM1/PnP-VCVE) is pure Python on PyTorch/mmcv and has no FFI of its own;
 * each entry point below names the reference function it replaces (paths relative to
 * /root/reference).  All pointers named *_dev are device (HBM) addresses owned by the
 * caller; the library allocates no device memory, launches asynchronously on the supplied
 * stream (a hipStream_t passed as void*) and returns 0 on success, a hipError_t value, or
 * one of the PNP_ERR_* codes.  fp32 throughout unless PNP_PREC_F16 is selected.
 * State: all mutable state lives in the pnp_generator handle (precision, options, profiling events, side streams);
 * a handle must not be used from two threads at once.  The only process-wide state is a per-DEVICE cache of one-time
 * kernel attributes (dynamic-LDS opt-in, CU count), keyed by the current device and mutex-protected, so one process may
 * drive several GPUs.  No environment variables are read.
 */
#ifndef PNPVCVE_H
#define PNPVCVE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PNP_ERR_BAD_ARG 1001
#define PNP_ERR_UNSUPPORTED 1002
#define PNP_ERR_WORKSPACE 1003
#define PNP_ERR_SIZE_ASSERT 1004 /* reference: AssertionError, h/w < 64 (iconvsr_ipb_par.py:51) */
#define PNP_ERR_SIZE_VALUE 1005  /* reference: ValueError from flow_warp.py:27-29 (h/w % 4 != 0) */

int pnp_abi_version(void); /* 5: PNP_OPT_WINOGRAD, pnp_wino_* / pnp_conv3x3_wino_f32.  4: pnp_generator_cfg grew num_group / flow_inter / blocktype (3: the never-implemented fused-block option / query of v2
                              removed, PNP_OPT_* renumbered, PNP_OPT_SPARSE_EVAL, PNP_OPT_F16_MIRRORS) */

/* ------------------------------------------------------------------ generator (a1/a2)
 * Constructor kwargs of IconVSR_restore_wo_refill_mv_ipb_fast_domain_dynamic_with_par
 * (mmedit/models/backbones/sr_backbones/iconvsr_ipb_par.py:18-41 and parents
 *  iconvsr_ipb.py:16-31, iconvsr.py:346-369).  Booleans are 0/1. */
typedef struct pnp_generator_cfg {
    int mid_channels; /* only 64 */
    int num_blocks;
    int num_experts;
    int with_cat, use_base_qp, expert_softmax, with_bias, with_se;
    int one_layer, channel_first, align_key, vsr;
    int deform; /* 0 'vos' (MV bilinear warp, iconvsr_mv.py:12-18); 1 'basic' (:52-84), 2 'fvc' (:21-41):
                   flow-guided modulated deformable conv, deform_groups 16 (mmcv semantics restated) */
    int sparse_val; /* eval-time sparse evaluation of the 1x1 partition branches (basicvsr_net.py:456-476,511-514,
                       sr_backbone_utils.py:262-302): branch j where plane j != 0, later planes overwrite, result / 255.
                       Identical to the dense path for one-hot maps in {0, 1/255}; one clip at a time (n == 1): the
                       reference indexes sample 0 only */
    int num_group;  /* groups of every conv of a BAE block (sr_backbone_utils.py:285-289): a divisor of 64 (0 = 1); packed as the dense
                       conv it equals.  Not with sparse_val (the reference's sparse_conv multiplies a (64, 64/groups) weight
                       with 64-channel columns and raises) */
    int flow_inter; /* 0 'bilinear', 1 'nearest' (flow_warp.py:8,47 -> F.grid_sample mode): the MV alignment and the warp inside
                       the 'basic' aligner */
    int blocktype;  /* 0 'drt'; 1 'drt_woqp' (ResidualBlockNoBNDynamic_drt_wo_qp, sr_backbone_utils.py:336-384): both 3x3 convs
                       are plain convs, no expert mix and no gain; needs one_layer = 1 (with Dynamic_conv2d_se convs the
                       reference indexes a tensor with 'x' and raises) */
} pnp_generator_cfg;

typedef struct pnp_generator pnp_generator;

int pnp_generator_create(const pnp_generator_cfg* cfg, pnp_generator** out);
void pnp_generator_destroy(pnp_generator* g);

/* Parameter schema = the reference state-dict: name, shape and the
 * float offset of each tensor inside the flat parameter buffer the caller fills. */
int pnp_generator_num_params(const pnp_generator* g);
const char* pnp_generator_param_name(const pnp_generator* g, int i);
int pnp_generator_param_ndim(const pnp_generator* g, int i);
int64_t pnp_generator_param_dim(const pnp_generator* g, int i, int d);
int64_t pnp_generator_param_offset(const pnp_generator* g, int i);
int64_t pnp_generator_flat_floats(const pnp_generator* g);
int64_t pnp_generator_packed_floats(const pnp_generator* g);

/* Arithmetic of the 64-channel convs (BASELINE configs[4]; mmcv's wrap_fp16_model / fp16_enabled switch,
 * mmedit/models/restorers/basic_restorer.py:64 @auto_fp16).  PNP_PREC_F32 (default): exact fp32 MFMA.
 * PNP_PREC_F16: activations and weights rounded to fp16 as MFMA operands, fp32 accumulation, fp32 feature
 * maps in HBM; conv_last and an RGB-only input conv stay fp32.
 * PNP_PREC_F16X3: split fp16 -- every activation and weight of the NHWC64 64-channel convs is carried as
 * hi + lo/2048 (two fp16 numbers, 22 significand bits) and each product is three fp16 MFMAs with fp32
 * accumulation; results agree with PNP_PREC_F32 to ~1e-6 relative per conv (inside north_star's 1e-3 gate,
 * which PNP_PREC_F16 is not) at about a third of the fp16 matrix rate.  Feature maps stay fp32; the RGB
 * frame, the RGB head (conv_last) and the deformable alignment stay on the exact fp32 kernels.  Range: the high
 * parts are fp16, so activations and weights beyond +-65504 saturate (the exact fp32 path has no such limit).
 * Set it BEFORE sizing/packing: it changes pnp_generator_packed_floats and pnp_generator_workspace_bytes. */
#define PNP_PREC_F32 0
#define PNP_PREC_F16 1
#define PNP_PREC_F16X3 2
int pnp_generator_set_precision(pnp_generator* g, int precision);
int pnp_generator_get_precision(const pnp_generator* g);

/* Per-generator execution switches (A/B and diagnostic; results are bit-identical either way unless stated).
 * value 0/1 (PNP_OPT_WINOGRAD: 0/1/2); all default to 1 except PNP_OPT_F16_CHAIN_MIRRORS.  State lives in the handle. */
#define PNP_OPT_F16_MAPS 0       /* PNP_PREC_F16: the map between the two halves of a BAE block / behind conv_hr is stored fp16 */
#define PNP_OPT_PAR_SKIP 1       /* skip 1x1 partition branches whose plane is all zero on a tile (exact zeros); on frames whose every 8x8
                                    quadrant is zero or one constant plane on its pixels inside the image (any frame size) also launch the
                                    front halves in a cheaper form behind a device-side gate on the frame's partition word (bit-identical
                                    results) */
#define PNP_OPT_CONV_LAST_VALU 2 /* conv_last on the vector ALUs (<= 2e-6 from the MFMA kernel: other summation order) */
#define PNP_OPT_PERSIST 3        /* persistent strip kernel for the 64->64 convs on frames with >= 1024 tiles */
#define PNP_OPT_SMALL_F16 4      /* PNP_PREC_F16: tile-per-block fp16 kernel for the 64->64 convs on frames with < 1024 tiles */
#define PNP_OPT_SPARSE_EVAL 5    /* cfg.sparse_val only: 1 = eval mode (sparse semantics), 0 = training mode -- the reference takes the
                                    sparse branch only when `self.sparse_val and not self.training` (sr_backbone_utils.py:308,322,
                                    basicvsr_net.py:511) and the dense par * conv1x1 formula otherwise (NOT bit-identical: the two
                                    differ on non-one-hot maps) */
#define PNP_OPT_F16_MIRRORS 6    /* PNP_PREC_F16 (with PNP_OPT_F16_MAPS, deform 'vos'): every 64-channel map that is only read as an MFMA A
                                    operand (the running map of a branch, the frame slots, the MV-aligned key frame) gets an fp16 copy
                                    from its producer, and the input conv of a branch runs as ONE launch over those copies instead of
                                    a chain of single-source launches through fp32 partial sums.  Bit-identical: same rounding points */
#define PNP_OPT_F16_CHAIN_MIRRORS 7 /* with PNP_OPT_F16_MIRRORS: also mirror the running map x inside a branch (default 0: measured
                                    neutral at 720p -- the back half writes 128 B per pixel more for what the front half reads less) */
#define PNP_OPT_TILE_QUEUE 8     /* PNP_PREC_F16X3: the split conv kernel's blocks draw their tiles from a per-XCD queue in the workspace instead of
                                    walking a static share (tiles differ in cost and the two blocks of a CU in speed); results identical */
#define PNP_OPT_WINOGRAD 9       /* PNP_PREC_F32: the single-source 64 -> 64 convs (both halves of a BAE block, conv_hr) and the input convs over wide sources in
                                    Winograd F(2x2,3x3) form (conv_wino.hip): 2.25x fewer matrix FLOPs, still fp32 products and sums, NOT bit-identical
                                    to the direct kernels (summation order + the +-1 input transform: ~1e-6 per conv on unit-scale maps; whole-clip
                                    gates in tests/test_gpu_wino.py).  0 off | 1 (default): frames of up to PNP_WINO_UNITS_MAX_TILES 16x16 tiles
                                    one block per 8x8 quadrant unit, larger ones the persistent tile kernels (same values bit for bit) |
                                    2: the tile kernels at every frame size.  This is the ONE statement of the rule */
#define PNP_WINO_UNITS_MAX_TILES 128   /* ceil(h/16) * ceil(w/16) up to which PNP_OPT_WINOGRAD = 1 takes the quadrant-unit kernels */
#define PNP_OPT_COUNT 10
int pnp_generator_set_option(pnp_generator* g, int option, int value);
int pnp_generator_get_option(const pnp_generator* g, int option);

/* flat (reference layouts) -> packed (MFMA B images); replaces nothing in the reference,
 * it is the checkpoint-load-time half of mmcv load_checkpoint (iconvsr.py:510-523). */
int pnp_generator_pack(const pnp_generator* g, const float* flat_dev, float* packed_dev, void* stream);

/* Bytes of ONE workspace context (one clip in flight).  pnp_generator_forward accepts a workspace of k such
 * contexts (k <= PNP_MAX_CONTEXTS) and then runs up to k samples of the batch concurrently on internal streams
 * forked from / joined to the caller's stream -- worth it for small frames, which cannot fill 256 CUs alone. */
#define PNP_MAX_CONTEXTS 8
int64_t pnp_generator_workspace_bytes(const pnp_generator* g, int t, int h, int w);

/* generator.forward(lrs, QPs, slices, mvs, base_QPs, par_map)  iconvsr_ipb_par.py:44-149.
 *   lrs_dev (n,t,3,h,w)  mvs_dev (n,t,4,h,w)  par_dev (n,t,3,h,w)      NCHW, contiguous
 *   slices/qps/base_qps: HOST arrays of n*t floats (the (n,t,1,1,1) tensors, flattened);
 *   the reference reads the same values host-side (int(torch.where(...)), :81,:116).
 *   out_dev (n,t,3,h,w), or (n,t,3,4h,4w) when cfg.vsr. */
int pnp_generator_forward(const pnp_generator* g, const float* flat_dev, const float* packed_dev,
                          const float* lrs_dev, const float* mvs_dev, const float* par_dev,
                          const float* slices_host, const float* qps_host, const float* base_qps_host,
                          float* out_dev, void* workspace_dev, int64_t workspace_bytes,
                          int n, int t, int h, int w, void* stream);

/* Optional per-launch timing with HIP events recorded on the caller's stream around every
 * kernel of pnp_generator_forward (measurement aid for bench.py; replaces the reference's
 * wall-clock print, mmedit/models/restorers/basicvsr.py:176-182).  Enable, run forwards,
 * then read per kind: total device ms, launch count and algorithmic work (FLOPs for the
 * conv kinds, bytes for PNP_PROF_WARP).  Reading waits for the recorded events. */
#define PNP_PROF_CONV_BLOCK 0 /* the 64->64 BAE convs (K = 576 or 768) */
#define PNP_PROF_CONV_INPUT 1 /* input convs over the virtual concat */
#define PNP_PROF_CONV_HEAD 2  /* conv_hr is BLOCK-shaped but conv_last / upsample convs land here */
#define PNP_PROF_WARP 3       /* MV-guided bilinear alignment, 520 B per pixel */
#define PNP_PROF_DCN 4        /* modulated deformable alignment (deform = basic|fvc), 2240 B per pixel */
int pnp_generator_profile(pnp_generator* g, int enable);
int pnp_generator_profile_read(pnp_generator* g, int kind, double* total_ms, int64_t* launches, double* work);

/* ------------------------------------------------------------------ single ops
 * flow_warp(x, flow, 'bilinear', 'zeros', align_corners=True)  mmedit/models/common/flow_warp.py:6-50
 *   x (n,c,h,w) NCHW ; flow (n,h,w,2) pixels (dx,dy) ; out (n,c,h,w). */
int pnp_flow_warp_nchw_f32(const float* x_dev, const float* flow_dev, float* out_dev,
                           int n, int c, int h, int w, void* stream);
/* The same gather in the pixel-major layout the fused path uses (VOSAlignment.forward,
 * iconvsr_mv.py:17-18): feat/out (h,w,c) ; flow_x/flow_y (h,w) planes. */
int pnp_mv_warp_nhwc_f32(const float* feat_dev, const float* flow_x_dev, const float* flow_y_dev,
                         float* out_dev, int h, int w, int c, void* stream);
/* Both with flow_warp's `interpolation` argument (flow_warp.py:8,47 -> F.grid_sample mode): mode 0 'bilinear', 1 'nearest' (the
 * pixel at nearbyint of the sampling position, ties to even, 0 outside the image -- ATen's grid_sampler_2d). */
int pnp_flow_warp_nchw_mode_f32(const float* x_dev, const float* flow_dev, float* out_dev, int n, int c, int h, int w, int mode,
                                void* stream);
int pnp_mv_warp_nhwc_mode_f32(const float* feat_dev, const float* flow_x_dev, const float* flow_y_dev, float* out_dev, int h,
                              int w, int c, int mode, void* stream);

int pnp_nchw_to_nhwc_f32(const float* in_dev, float* out_dev, int n, int c, int h, int w, void* stream);
int pnp_nhwc_to_nchw_f32(const float* in_dev, float* out_dev, int n, int c, int h, int w, void* stream);

/* Base_Predictor / SEModule  (domain_aware.py:172-183, 210-222); host QP values in,
 * ew_dev (count, E) and gamma_dev (count, 64) out.  v1/v2 may be NULL (gamma = 1). */
int pnp_caa_predict_f32(const float* q_ew_host, const float* q_gamma_host, int count, int num_experts,
                        int softmax, const float* w1_dev, const float* b1_dev, const float* w2_dev,
                        const float* b2_dev, const float* v1_dev, const float* v2_dev,
                        float* ew_dev, float* gamma_dev, void* stream);

/* Weight packing for pnp_conv3x3_f32: OIHW (cout, cin_total, 3, 3) -> B image of input
 * channels [cbase, cbase+csrc) (csrc = 64 -> 9 chunks, csrc = 3 -> 1 chunk); with
 * num_experts > 1, w is (E, cout, cin_total, 3, 3) and ew_dev (E) mixes the experts first
 * (Dynamic_conv2d_se.forward, sr_backbone_utils.py:198-199).  cout <= 64. */
int64_t pnp_packed_conv_floats(int csrc);
int pnp_pack_conv3x3_f32(const float* w_dev, const float* ew_dev, int num_experts, int cout, int cin_total,
                         int cbase, int csrc, float* dst_dev, void* stream);
/* 1x1 (64,64,1,1) -> 1 chunk */
int pnp_pack_conv1x1_f32(const float* w_dev, float* dst_dev, void* stream);

/* Generic fused 3x3 conv over a virtual concat of up to 4 pixel-major sources
 * (64 channels each, or the 4-channel RGB0 frame):
 *   v = conv3x3(cat(srcs)) ; v = (v + bias) * gamma ; v += sum_j par_j * conv1x1_j(src0)
 *   v = act(v) ; v += residual ; out (h,w,64)
 * This one op covers input_conv+LeakyReLU (basicvsr_net.py:484,515), both halves of
 * ResidualBlockNoBNDynamic_drt.forward (sr_backbone_utils.py:304-333) and conv_hr
 * (iconvsr.py:365).  NULL for unused bias/gamma/par/w1x1/residual.  act: 0 none, 1 relu,
 * 2 leaky-relu(0.1).  par_dev: 3 NCHW planes (3,h,w). */
int pnp_conv3x3_f32(int nsrc, const float* const* srcs_dev, const int* src_channels,
                    const float* const* packed_w_dev, const float* bias_dev, const float* gamma_dev,
                    const float* packed_w1x1_dev, const float* par_dev, const float* residual_dev,
                    int act, float* out_dev, int h, int w, void* stream);

/* One BAE block, ResidualBlockNoBNDynamic_drt.forward in the shipped layout (channel_first, one_layer;
 * sr_backbone_utils.py:305-313,329):  out = x + conv1(relu(gamma * (conv2_mix(x) + b2) + sum_j par_j * conv1x1_j(x))) + b1.
 * w2_packed: the expert-mixed dynamic conv (pnp_pack_conv3x3_f32 with num_experts > 1 = Dynamic_conv2d_se's
 * mm(attention, weight), :198-199); w1_packed: the static conv1; w1x1_packed / par may both be NULL.
 * scratch_dev: h*w*64 floats for the intermediate; out_dev may alias x_dev. */
int pnp_bae_block_f32(const float* x_dev, const float* w2_packed_dev, const float* b2_dev, const float* gamma_dev,
                      const float* w1x1_packed_dev, const float* par_dev, const float* w1_packed_dev,
                      const float* b1_dev, float* scratch_dev, float* out_dev, int h, int w, void* stream);

/* PixelShufflePack (mmedit/models/common/upsample.py:8-51): conv3x3 64 -> 256 followed by F.pixel_shuffle(2).
 * w (256,64,3,3), b (256) -> packed image (pnp_packed_pixel_shuffle_floats floats); x (h,w,64) -> out (2h,2w,64),
 * act as in pnp_conv3x3_f32 (the x4 head applies leaky-relu, iconvsr_ipb_par.py:136-137). */
int64_t pnp_packed_pixel_shuffle_floats(void);
int pnp_pack_pixel_shuffle_f32(const float* w_dev, const float* b_dev, float* dst_dev, void* stream);
int pnp_pixel_shuffle_conv_f32(const float* x_dev, const float* packed_dev, int act, float* out_dev, int h, int w,
                               void* stream);

/* The single-source 64 -> 64 form of pnp_conv3x3_f32 as Winograd F(2x2,3x3) (sr_backbone_utils.py:304-333 block halves,
 * iconvsr_ipb_par.py:144 conv_hr): act(gamma * (conv3x3(x; W) + bias) + Sum_j par_j * conv1x1_j(x)) + residual with
 * x (h,w,64).  wino_w_dev = pnp_wino_image_from_packed_f32 of the packed direct-conv image WITH the same gamma: the gain of the conv
 * term lives in the transformed weights, and gamma_dev HERE SCALES ONLY THE BIAS -- an image built without it (or with another one)
 * gives gamma * bias + conv instead of gamma * (conv + bias), silently; the C ABI cannot tell (the Python op checks the pairing);
 * wino_w1x1_dev = pnp_wino_par_image_from_packed_f32 of the
 * packed 1x1 images or NULL (then par_dev / par_flags_dev are ignored); par_flags_dev as pnp_par_tile_flags_f32 writes
 * them, or NULL.  fp32 arithmetic; differs from pnp_conv3x3_f32 by summation order (~1e-6 on unit-scale maps).  With residual_dev the
 * tile form takes act = 0 only (what the reference's blocks do; PNP_ERR_UNSUPPORTED otherwise), the unit form any act. */
int64_t pnp_wino_image_floats(void);
int64_t pnp_wino_par_image_floats(void);
int pnp_wino_image_from_packed_f32(const float* packed_w_dev, const float* gamma_dev, float* dst_dev, void* stream);
int pnp_wino_par_image_from_packed_f32(const float* packed_w1x1_dev, float* dst_dev, void* stream);
/* The input conv (basicvsr_net.py:484 over the virtual concat of iconvsr_ipb_par.py:90,125) in the same form: srcs_dev[0] the frame
 * as (h,w,4) RGB0, srcs_dev[1..nsrc-1] one to three (h,w,64) maps; wino_w_dev[0] = pnp_wino_rgb_image_from_packed_f32 of the frame's
 * packed chunk (pnp_pack_conv3x3_f32 with csrc 3), wino_w_dev[s] = pnp_wino_image_from_packed_f32 (no gamma) of source s's image.
 * The images of the 64-channel sources must lie within 4 GiB of each other.  out = act(sum_s conv3x3(src_s) + bias). */
int64_t pnp_wino_rgb_image_floats(void);
int pnp_wino_rgb_image_from_packed_f32(const float* packed_rgb_chunk_dev, float* dst_dev, void* stream);
int pnp_conv3x3_wino_ms_f32(int nsrc, const float* const* srcs_dev, const float* const* wino_w_dev, const float* bias_dev,
                            int act, float* out_dev, int h, int w, void* stream);
int pnp_conv3x3_wino_ms_units_f32(int nsrc, const float* const* srcs_dev, const float* const* wino_w_dev, const float* bias_dev,
                                  int act, float* out_dev, int h, int w, void* stream);   /* one block per quadrant unit (small frames) */
int pnp_conv3x3_wino_f32(const float* src_dev, const float* wino_w_dev, const float* bias_dev, const float* gamma_dev,
                         const float* wino_w1x1_dev, const float* par_dev, const int* par_flags_dev,
                         const float* residual_dev, int act, float* out_dev, int h, int w, void* stream);
/* The same conv with one block per 8x8 QUADRANT of a 16x16 tile (four waves = four 16-channel slices of the output): what
 * pnp_generator_forward uses on frames too small to fill the chip with whole tiles (<= PNP_WINO_UNITS_MAX_TILES of them; PNP_OPT_WINOGRAD = 1).  Same
 * arguments, same values bit for bit. */
int pnp_conv3x3_wino_units_f32(const float* src_dev, const float* wino_w_dev, const float* bias_dev, const float* gamma_dev,
                               const float* wino_w1x1_dev, const float* par_dev, const int* par_flags_dev,
                               const float* residual_dev, int act, float* out_dev, int h, int w, void* stream);

/* Which of the three 1x1 partition branches (sr_backbone_utils.py:310-311, Sum_j par_j * conv1x1_j(x)) an 8x16 pixel tile
 * needs at all: par_dev (3,h,w) -> flags_dev[((w+15)/16) * ((h+7)/8)] ints, bit j set iff plane j is nonzero somewhere in
 * the tile.  pnp_generator_forward computes these once per frame; its persistent conv kernel skips a branch on tiles
 * where the plane is zero (exact zeros: bit-identical result).  Codec partition maps are one-hot per >= 8x8 block.
 * Bit 3 + j: every value of plane j in the tile is 0 or exactly 1/255 (what loading_ipb.py writes: uint8 one-hot / 255.) -- the
 * split-fp16 kernel then contracts a masked operand with weights scaled at pack time; consumers of bits 0..2 mask with 7. */
int pnp_par_tile_flags_f32(const float* par_dev, int* flags_dev, int h, int w, void* stream);

/* The same op with fp16 MFMA operands (fp32 sources, accumulation and output): packed_w_f16 are fp16 images
 * made by pnp_f16_image_from_f32 from the fp32 images above (nchunks = 9 per 64-channel source, 1 per RGB0
 * source, 3 for the 1x1 branches; same element count, so nchunks * 8192 bytes).  At least one source must
 * have 64 channels; several 64-channel sources exclude gamma / residual; otherwise PNP_ERR_UNSUPPORTED. */
int pnp_f16_image_from_f32(const float* packed_w_dev, void* dst_dev, int nchunks, void* stream);
int pnp_conv3x3_f16(int nsrc, const float* const* srcs_dev, const int* src_channels,
                    const void* const* packed_w_f16_dev, const float* bias_dev, const float* gamma_dev,
                    const void* packed_w1x1_f16_dev, const float* par_dev, const float* residual_dev,
                    int act, float* out_dev, int h, int w, void* stream);

/* The same op in split fp16 (PNP_PREC_F16X3): every activation x and weight w is carried as hi + lo/2048 with
 * hi = fp16(x), lo = fp16((x - hi) * 2048), a product is three fp16 MFMAs (hi*hi, lo*hi, hi*lo) accumulated in
 * fp32 -- fp32-level results (~1e-6 relative to pnp_conv3x3_f32) from the fp16 matrix pipe.  fp32 sources and
 * output.  packed_w_f32 are the fp32 images (read only for a 4-channel RGB0 source, which runs on the exact fp32
 * kernel; may be NULL for 64-channel sources); packed_w_x3 = pnp_f16x3_image_from_f32 of the same fp32 images
 * (64-channel sources and the 1x1 branches; hi and lo halves interleaved: nchunks * 16384 bytes, nchunks as for
 * pnp_f16_image_from_f32).  par_flags_dev: optional pnp_par_tile_flags_f32 output.  Restrictions as for
 * pnp_conv3x3_f16. */
int pnp_f16x3_image_from_f32(const float* packed_w_dev, void* dst_dev, int nchunks, void* stream);
int pnp_conv3x3_f16x3(int nsrc, const float* const* srcs_dev, const int* src_channels,
                      const float* const* packed_w_f32_dev, const void* const* packed_w_x3_dev,
                      const float* bias_dev, const float* gamma_dev, const void* packed_w1x1_x3_dev,
                      const float* par_dev, const int* par_flags_dev, const float* residual_dev, int act,
                      float* out_dev, int h, int w, void* stream);

/* Per-frame sum of squared differences of the uint8-rounded frames (the statistic behind
 * psnr(tensor2img(a), tensor2img(b)), mmedit/core/misc.py:51-71 + core/evaluation/metrics.py:200-215):
 * a, b (frames, c, h, w) fp32 in HBM -> sse_dev (frames) uint64, exact.  PSNR = 20 log10(255 / sqrt(sse / N)),
 * N = c * (h - 2 crop) * (w - 2 crop). */
int pnp_psnr_sse_f32(const float* a_dev, const float* b_dev, unsigned long long* sse_dev, int frames,
                     int c, int h, int w, int crop_border, void* stream);

/* tensor2img for the PNG write-back (mmedit/core/misc.py:51-71, basicvsr.py:205-231): frames (n,3,h,w) fp32 ->
 * out (n,h,w,3) uint8 RGB = round_half_even(clamp(x,0,1) * 255); a quarter of the D2H bytes of the fp32 frames. */
int pnp_frames_to_rgb8(const float* frames_dev, unsigned char* out_dev, int nframes, int h, int w, void* stream);

/* SSIM statistic (mmedit/core/evaluation/metrics.py:266-355: per channel, 11x11 Gaussian sigma 1.5, 'valid'
 * window, fp64, on the uint8-rounded frames).  Writes one partial sum of the SSIM map per 16x32 tile:
 * partials_dev [frames*c][pnp_ssim_blocks(h,w,crop)] doubles; SSIM(frame) = mean over channels of
 * sum(partials) / ((h-2crop-10)(w-2crop-10)). */
int pnp_ssim_blocks(int h, int w, int crop_border);
int pnp_ssim_partials_f32(const float* a_dev, const float* b_dev, double* partials_dev, int frames, int c, int h,
                          int w, int crop_border, void* stream);

/* Modulated deformable 3x3 conv, 64 -> 64 channels, deform_groups 16 (mmcv.ops.modulated_deform_conv2d as
 * called at mmedit/models/backbones/sr_backbones/iconvsr_mv.py:38-41,81-84; semantics restated, mmcv is not
 * vendored).  x_dev (h,w,64) pixel-major; om_dev (h,w,448): conv_offset[2] output (pre-sigmoid masks) in the
 * channel order given by pnp_dcn_ref_channel (packed channel -> reference channel of the 432, -1 = padding);
 * flow_x/flow_y (h,w) planes added to every (dx, dy) offset or NULL; w_packed = pnp_pack_conv3x3_f32 image. */
int pnp_dcn_nhwc_f32(const float* x_dev, const float* om_dev, const float* flow_x_dev, const float* flow_y_dev,
                     const float* w_packed_dev, const float* bias_dev, float* out_dev, int h, int w, void* stream);
int pnp_dcn_ref_channel(int packed_channel);
/* The same op with fp16 MFMA operands (PNP_PREC_F16: samples and weights rounded to fp16, fp32 accumulation):
 * w_f16_dev = pnp_dcn_f16_image_from_f32(w_packed) (9 * 4096 halfs = 73,728 bytes). */
int pnp_dcn_f16_image_from_f32(const float* w_packed_dev, void* dst_dev, void* stream);
int pnp_dcn_nhwc_f16(const float* x_dev, const float* om_dev, const float* flow_x_dev, const float* flow_y_dev,
                     const void* w_f16_dev, const float* bias_dev, float* out_dev, int h, int w, void* stream);

/* MV / partition records -> dense maps: the inner loop of LoadImageFromFileList_ipb.__call__
 * (mmedit/datasets/pipelines/loading_ipb.py:328-369) + RescaleToZeroOne(partitions) + HWC->CHW.
 *   records_dev (R,10) fp32 rows (direction, w, h, x_w, y_w, x, y, motion_x, motion_y, scale) in file
 *   order over the whole clip; rec_frame_dev (R) int32 frame of each row; slices_host (t) = ord('I'|'P'|'B').
 *   mvs_dev (t,4,h,w) and par_dev (t,3,h,w) are fully written; scratch_dev = t*2*h*w int32.  t <= 256. */
int pnp_rasterise_side_info_f32(const float* records_dev, const int* rec_frame_dev, long num_records,
                                const float* slices_host, int t, int h, int w, float* mvs_dev, float* par_dev,
                                int* scratch_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
