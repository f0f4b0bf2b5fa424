// 3x3 (+ fused 1x1 partition branches) implicit-GEMM convolution on the fp32
// MFMA pipe of gfx950.  See conv_mfma.hip for the design notes.
#pragma once
#include "common.h"

struct ConvArgs {
    const float* src[4];    // NHWC sources (virtual concat, never materialised)
    const float* wsrc[4];   // packed chunk images per source (9 chunks for C=64, 1 for C=4)
    int src_c[4];           // 64 or 4
    int nsrc;
    const float* wpar;      // 3 chunks (conv16x16, conv16x8, conv8x8) or nullptr
    const float* wwino;     // Winograd F(2x2,3x3) image of wsrc[0] (16 chunks of 4096 floats, gamma folded in: launch_wino_images), or nullptr:
                            // with it a single-source 64 -> 64 conv (prec 0, out_mode 0) runs on conv_wino.hip
    const float* wwino_par; // ... and of wpar (launch_wino_par_image; 12288 floats); required with wwino when wpar is set
    const float* wwino_src[4]; // the input conv in Winograd form (source 0 the RGB frame, then 1..3 64-channel sources): the image of
    const float* wwino_rgb;    // wsrc[s] for s >= 1 (launch_wino_images) and of the frame's chunk wsrc[0] (launch_wino_rgb_image)
    const int* par_any;        // with wwino + wpar (tile kernel): the frame's partition word (launch_par_frame_any; bit 3 = every 8x8 quadrant of
                               // the frame is all zero or carries one constant plane).  The conv is then launched twice behind a device-side
                               // gate: the fold-only kernel (runs iff bit 3 is set) and the branch kernel (iff not); any frame size
    int wino_units;            // with wwino: one block per 8x8 quadrant unit (conv3x3_wino_quad_kernel: frames too small to fill the chip with 16x16 tiles)
    const float* wvalu;     // conv_last only: [9][64][4] weights for the vector-ALU kernel (conv_last.hip), or nullptr
    const void* wsrc_h[4];  // prec == 1: fp16 twins of wsrc / wpar (conv_f16.hip); prec == 2: their split images (hi and lo
                            // interleaved, twice the halfs; conv_f16x3.hip launch_f16x3_image)
    const void* wpar_h;
    int wpar_h_scaled;      // prec == 2: wpar_h holds 12 split chunks -- the three branch images, then the same three scaled by PNP_PAR_UNIT
    int* tile_queue;        // prec == 2: 16 zeroed ints the split kernel hands its tiles out from (conv_f16x3.hip, X3Args::queue), or nullptr
    int prec;               // 0 fp32 MFMA | 1 fp16 operands, fp32 accumulate (where conv_f16_eligible)
                            // 2 split fp16 (hi + lo / 2048, three MFMAs per product, fp32-level results; where conv_f16x3_eligible)
    int src_f16, out_f16;   // prec 1, out_mode 0: bit s of src_f16: src[s] IS an fp16 NHWC64 map (the mirror its producer
                            // wrote); out_f16: out is one (single source, no residual)
    void* out16;            // prec 1, out_mode 0, !out_f16: additionally write an fp16 NHWC64 mirror of the fp32 output
    int no_multi16;         // 1: several fp16 sources are refused instead of running the one-launch input-conv kernel
    const float* par;       // 3 NCHW planes of the partition map, nullptr if wpar == nullptr
    const int* par_flags;   // optional, one int per 8x16 tile (row-major; consumers read bits 0..2 through `& 7`): bit j set <=> plane j has a nonzero value in
                            // the tile (launch_par_tile_flags).  The persistent kernel skips the 1x1 branches whose
                            // plane is zero over its whole tile -- exact zeros, bit-identical result for finite activations (a skipped
                            // 0 * inf would have been NaN).  nullptr: none skipped
    long par_plane;         // floats between planes
    const float* bias;      // [N] or nullptr
    const float* gamma;     // [N] channel gain applied to (conv + bias) BEFORE the 1x1 branches, or nullptr
    const float* residual;  // NHWC64, added after the activation, or nullptr
    float* out;
    const float* lr;        // out_mode 2/3: RGB frame, 3 NCHW planes
    long lr_plane;
    long w_ystride;         // floats between the weight images of consecutive blockIdx.y
    int bias_ystride;
    int H, W;               // spatial size of the sources
    int act;                // 0 none, 1 relu, 2 leaky-relu(0.1)
    unsigned long long* dbg; // diagnostic timeline buffer (8 u64 per block) or nullptr
    int no_small16;         // 1: never the small-frame fp16 kernel (PNP_OPT_SMALL_F16 0): persistent fp16 kernel at every size
    int no_persist;         // 1: never the persistent kernel (pnp_generator_set_option PNP_OPT_PERSIST 0 / pnp_conv3x3_f32_ex)
    int out_cstride;        // out_mode 4: channels per pixel of the output buffer
    int out_mode;           // 0 NHWC64 | 1 NHWC64 pixel-shuffle(2), sub-pixel = blockIdx.y
                            // 2 RGB NCHW + lr | 3 RGB NCHW + bilinear x4 upsample of lr (H/4 x W/4)
                            // 4 NHWC with out_cstride channels per pixel, channel block 64*blockIdx.y
};

enum { CONV_CFG_BIG = 0,    // 8x16 pixel tile, 64 output channels per block
       CONV_CFG_SMALL = 1,  // 4x16 pixel tile, 64 output channels per block (fills the chip on small frames)
       CONV_CFG_RGB = 2 };  // 8x16 pixel tile, <=32 output channels (conv_last)

// grid_y > 1 only for out_mode 1 (4 sub-pixel images).
int launch_conv3x3(const ConvArgs& a, int cfg, int grid_y, hipStream_t stream);
int conv_pick_cfg(int H, int W);

// fp16-operand variant (conv_f16.hip): fp32 B image -> fp16 image (same element count, chunk by chunk)
int launch_f16_image(const float* src, void* dst, int nchunks, int ntb, hipStream_t stream);
// which launches the fp16-operand kernels take (a pure function of the arguments: shared with the host-only scheduler test build)
static inline bool conv_f16_eligible(const ConvArgs& a, int cfg, int grid_y) {
    const bool has_par = a.wpar || a.wpar_h;
    if (cfg == CONV_CFG_RGB)     // conv_last: one 64-channel source, 3 NCHW planes + the low-quality frame
        return (a.out_mode == 2 || a.out_mode == 3) && a.nsrc == 1 && a.src_c[0] == 64 && a.wsrc_h[0] && grid_y == 1 &&
               !has_par && !a.residual && !a.gamma && !a.out_f16 && a.lr;
    if (a.out_mode != 0 && a.out_mode != 1 && a.out_mode != 4) return false;
    int nwide = 0;
    for (int s = 0; s < a.nsrc; ++s) {
        if (a.src_c[s] == 64) ++nwide;
        if (!a.wsrc_h[s]) return false;
    }
    if (nwide == 0) return false;                        // an RGB-only input conv stays on the fp32 kernel
    if (nwide > 1 && (a.residual || a.gamma || grid_y != 1 || a.out_mode != 0)) return false;
    if (has_par && (a.nsrc != 1 || !a.wpar_h || !a.par || grid_y != 1)) return false;
    // fp16 maps (a source read through its fp16 mirror, an fp16 output, an fp16 mirror of the fp32 output)
    // ... in the NHWC64 modes: plain (0) and, for a source / output map (no mirror), the pixel-shuffle conv (1, four sub-pixel images)
    if ((a.src_f16 || a.out_f16) && !(a.out_mode == 0 && grid_y == 1) && !(a.out_mode == 1 && a.nsrc == 1 && !a.out16)) return false;
    if (a.out16 && (grid_y != 1 || a.out_mode != 0)) return false;
    if (a.out_f16 && (a.residual || a.out16 || a.nsrc != 1)) return false;
    if (a.src_f16 && a.nsrc != 1) {          // several sources: the one-launch kernel, which reads ALL wide sources as fp16
        for (int s = 0; s < a.nsrc; ++s)
            if (a.src_c[s] == 64 && !((a.src_f16 >> s) & 1)) return false;
        if (a.residual || a.gamma || has_par) return false;
    }
    return true;
}
int launch_conv3x3_f16(const ConvArgs& a, int grid_y, hipStream_t stream);

// split-fp16 variant (conv_f16x3.hip): fp32 maps in and out, three fp16 MFMAs per product.  Takes the NHWC64 convs of 64-channel
// sources (+ optionally the RGB frame, which runs on the exact fp32 kernel as the first link of the source chain) and the pixel-
// shuffle and DCN offset convs (one launch per 64-channel output block); the RGB heads stay on the fp32 kernels.
int launch_f16x3_image(const float* src, void* dst, int nchunks, hipStream_t stream);      // nchunks * 16 KiB
static inline bool conv_f16x3_eligible(const ConvArgs& a, int cfg, int grid_y) {
    if (cfg != CONV_CFG_RGB && a.out_mode == 1 && grid_y == 4)       // pixel-shuffle conv: one 64-channel source, four sub-pixel images
        return a.nsrc == 1 && a.src_c[0] == 64 && a.wsrc_h[0] && !a.wpar && !a.wpar_h && !a.residual && !a.gamma && !a.src_f16 &&
               !a.out_f16 && !a.out16 && (long)a.H * a.W * 1024 < ((long)1 << 32);
    if (cfg != CONV_CFG_RGB && a.out_mode == 4 && grid_y * 64 == a.out_cstride)     // DCN offset conv: grid_y blocks of 64 channels
        return a.nsrc == 1 && a.src_c[0] == 64 && a.wsrc_h[0] && !a.wpar && !a.wpar_h && !a.residual && !a.gamma && !a.src_f16 &&
               !a.out_f16 && !a.out16 && (long)a.H * a.W * a.out_cstride * 4 < ((long)1 << 32);
    if (cfg == CONV_CFG_RGB || a.out_mode != 0 || grid_y != 1) return false;
    int nwide = 0;
    for (int s = 0; s < a.nsrc; ++s)
        if (a.src_c[s] == 64) {
            ++nwide;
            if (!a.wsrc_h[s]) return false;
        }
    if (nwide == 0) return false;
    const bool has_par = a.wpar || a.wpar_h;
    if (has_par && (a.nsrc != 1 || !a.wpar_h || !a.par || a.residual)) return false;   // a branch conv with a residual: fp32 kernels
    if (a.nsrc > 1 && (a.residual || a.gamma)) return false;
    return !a.src_f16 && !a.out_f16 && !a.out16;
}
int launch_conv3x3_f16x3(const ConvArgs& a, int cfg, hipStream_t stream);

// the partition value the reference's loader produces for an active plane: float32(1) / float32(255) (loading_ipb.py)
#define PNP_PAR_UNIT (1.0f / 255.0f)
// per-tile summary of a partition map for ConvArgs::par_flags (conv_persist.hip); flags: ((W+15)/16) * ((H+7)/8) ints;
// bits 0..2: plane j has a nonzero value in the tile; bits 3..5: every value of plane j in the tile is 0 or PNP_PAR_UNIT
// frames consecutive (3, H, W) maps -> frames consecutive flag arrays
int launch_par_tile_flags(const float* par, long par_plane, int* flags, int frames, int H, int W, hipStream_t stream);
int launch_par_frame_any(const int* flags, int* any, int frames, int H, int W, hipStream_t stream);

// conv_last on the vector ALUs (conv_last.hip): OIHW (3,64,3,3) -> [9][64][4]; 2304 floats
int launch_pack_last_valu(const float* w_oihw, float* dst, hipStream_t stream);
bool conv_last_valu_eligible(const ConvArgs& a, int cfg, int grid_y);
int launch_conv_last_valu(const ConvArgs& a, hipStream_t stream);

// Winograd F(2x2,3x3) variant of the single-source 64 -> 64 conv (conv_wino.hip): 2.25x fewer matrix FLOPs, fp32 arithmetic, results
// within ~1e-6 (unit-scale data) of the direct kernels.  Weight images: src[i] = a packed direct-conv image (9 chunks), dst[i] = 16
// chunks (65536 floats); gamma (64 floats or nullptr) is multiplied into every image (a dynamic conv's channel gain applies to
// conv + bias but not to the partition branches, which share the accumulators here -- so it lives in the weights; the kernel
// scales the bias).  n <= 16 images per launch.
int launch_wino_images(const float* const* src, float* const* dst, int n, const float* gamma, hipStream_t stream);
int launch_wino_par_image(const float* src_3chunks, float* dst_12288, hipStream_t stream);
int launch_wino_rgb_image(const float* src_chunk, float* dst_4096, hipStream_t stream);
bool conv_wino_eligible(const ConvArgs& a, int cfg, int grid_y);
bool conv_wino_ms_eligible(const ConvArgs& a, int cfg, int grid_y);
int launch_conv3x3_wino(const ConvArgs& a, hipStream_t stream);
#define PNP_WINO_IMG_FLOATS 65536
#define PNP_WINO_PAR_FLOATS 12288
#define PNP_WINO_RGB_FLOATS 4096

// persistent single-source variant (conv_persist.hip)
bool conv_persist_eligible(const ConvArgs& a, int cfg, int grid_y);
int launch_conv3x3_persist(const ConvArgs& a, hipStream_t stream);
