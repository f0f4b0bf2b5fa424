/* libpnpvcve_hip.so -- diagnostic entry points (kernel-variant selection and in-kernel timelines).
 *
 * Declared separately from pnpvcve.h because nothing in the reference corresponds to them: they exist so that
 * tests can compare two kernels of this build bit for bit and so that tools/trace_*.py can read shader-clock
 * timelines.  Same conventions as pnpvcve.h (device pointers, void* stream, int return); no global state -- the
 * variant and the trace buffer are arguments of the call.
 */
#ifndef PNPVCVE_DEBUG_H
#define PNPVCVE_DEBUG_H
#include "pnpvcve.h"
#ifdef __cplusplus
extern "C" {
#endif

#define PNP_CONV_AUTO 0     /* what pnp_conv3x3_f32 does: persistent strips from 1024 tiles, else tile per block */
#define PNP_CONV_TILE 1     /* tile-per-block kernel, tile size picked from the frame size */
#define PNP_CONV_TILE_BIG 2 /* tile-per-block kernel, 8x16 tiles */

/* The fused conv of pnpvcve.h (sr_backbone_utils.py:304-333 halves, basicvsr_net.py:484, iconvsr.py:365) with an explicit kernel
 * variant, optional per-tile partition flags (pnp_par_tile_flags_f32; NULL = no branch skipped) and an optional
 * timeline buffer (8-16 u64 per block, device memory, NULL = none). */
int pnp_conv3x3_f32_ex(int nsrc, const float* const* srcs_dev, const int* src_channels,
                       const float* const* packed_w_dev, const float* bias_dev, const float* gamma_dev,
                       const float* packed_w1x1_dev, const float* par_dev, const float* residual_dev,
                       int act, float* out_dev, int h, int w, int variant, const int* par_flags_dev,
                       void* trace_dev, void* stream);

/* pnp_conv3x3_wino_f32 of pnpvcve.h with a timeline buffer (16 u64 per block: [0] start tick, [1] K-loop ticks, [2] epilogue ticks,
 * [3] end tick, [7] tiles walked, [13] / [14] the 100 MHz counter at start / end; NULL = none). */
int pnp_conv3x3_wino_f32_ex(const float* src_dev, const float* wino_w_dev, const float* bias_dev, const float* gamma_dev,
                            const float* wino_w1x1_dev, const float* par_dev, const int* par_flags_dev,
                            const float* residual_dev, int act, float* out_dev, int h, int w, void* trace_dev, void* stream);
/* The frame's partition word (pnp_generator_forward computes it per frame: bit 3 = every 8x8 quadrant is all zero or one constant
 * plane) for the NEXT pnp_conv3x3_wino_f32_ex calls with branches: they then take the one gated launch the generator uses (fold-only
 * body when bit 3 is set).  A device int; NULL = back to the ungated branch kernel.  Process-wide, not thread-safe: a tracing hook. */
int pnp_debug_wino_gate_word(const int* gate_word_dev);

/* The fp16-operand conv of pnpvcve.h with an optional timeline buffer (16 u64 per 4-wave group). */
int pnp_conv3x3_f16_ex(int nsrc, const float* const* srcs_dev, const int* src_channels,
                       const void* const* packed_w_f16_dev, const float* bias_dev, const float* gamma_dev,
                       const void* packed_w1x1_f16_dev, const float* par_dev, const float* residual_dev,
                       int act, float* out_dev, int h, int w, void* trace_dev, void* stream);

/* The modulated deformable conv of pnpvcve.h with an optional timeline buffer: 8 u64 per wave (8 waves per block,
 * one block per CU): shader-clock sums of window fill, prologue, gather, MFMA, barrier + weight hand-over, epilogue,
 * total, tiles.  The buffer must hold pnp_dcn_trace_u64s() elements (the launch has one block per CU of the current device). */
int pnp_dcn_trace_u64s(void);
int pnp_dcn_nhwc_f32_ex(const float* x_dev, const float* om_dev, const float* flow_x_dev, const float* flow_y_dev,
                        const float* w_packed_dev, const float* bias_dev, float* out_dev, int h, int w, void* trace_dev,
                        void* stream);

/* The fp16-operand conv with explicit fp16 maps, as pnp_generator_forward chains its launches under PNP_OPT_F16_MAPS /
 * PNP_OPT_F16_MIRRORS.  Bit s of src_f16_mask: srcs_dev[s] IS an fp16 NHWC64 map (else fp32); out_f16: out_dev is an fp16 map
 * (single source, no residual); out16_dev (optional, only with an fp32 out_dev): an fp16 copy of the output written in the same
 * pass; par_flags_dev: pnp_par_tile_flags_f32 output or NULL.  Several 64-channel sources run as ONE launch when all of them are
 * fp16 maps (chain = 0), or -- fp32 sources only -- as the chain of single-source launches through fp32 partial sums (chain = 1;
 * bit-identical). */
int pnp_conv3x3_f16_maps(int nsrc, const void* const* srcs_dev, const int* src_channels, int src_f16_mask,
                         const void* const* packed_w_f16_dev, const float* bias_dev, const float* gamma_dev,
                         const void* packed_w1x1_f16_dev, const float* par_dev, const int* par_flags_dev,
                         const float* residual_dev, int act, void* out_dev, int out_f16, void* out16_dev, int h, int w,
                         int chain, void* trace_dev, void* stream);

/* pnp_conv3x3_f16x3 with the in-kernel timeline: trace_dev = 8 u64 per persistent block (512 at most): s_memtime at kernel
 * start; cycles spent waiting for / splitting the halo into the A tiles; cycles in the K loops; s_memtime at the end; cycles in the
 * epilogues; tiles done; block lifetime in 100 MHz s_memrealtime ticks; HW_REG_LDS_ALLOC (low byte 0: the first block on its CU). */
int pnp_conv3x3_f16x3_ex(int nsrc, const float* const* srcs_dev, const int* src_channels,
                         const float* const* packed_w_f32_dev, const void* const* packed_w_x3_dev, const float* bias_dev,
                         const float* gamma_dev, const void* packed_w1x1_x3_dev, const float* par_dev,
                         const int* par_flags_dev, const float* residual_dev, int act, float* out_dev, int h, int w,
                         int w1x1_scaled, int* tile_queue_dev, void* trace_dev, void* stream);
/* tile_queue_dev: NULL (every block walks a static share of the tiles) or 16 ints, zero on entry and zero again when the launch
 * has finished: the blocks draw their tiles per XCD from it, as pnp_generator_forward's launches do from its workspace
 * (PNP_OPT_TILE_QUEUE).
 * w1x1_scaled != 0: packed_w1x1_x3_dev holds 12 split chunks -- the three branch images and then the same three scaled by 1/255
 * (what pnp_generator_pack lays out) -- and tiles whose partition values are all 0 or exactly 1/255 (par_flags bits 3..5) contract
 * the branches with the scaled images and a masked A operand instead of re-splitting par_j(pixel) * x.  trace_dev may be NULL. */

/* pnp_mv_warp_nhwc_f32 writing its result as an fp16 (h,w,c) map (saturating round-to-nearest-even of the fp32 value). */
int pnp_mv_warp_nhwc_f16out(const float* feat_dev, const float* flow_x_dev, const float* flow_y_dev, void* out16_dev, int h,
                            int w, int c, void* stream);

#ifdef __cplusplus
}
#endif
#endif
