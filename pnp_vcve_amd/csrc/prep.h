// Layout, weight-packing, expert-mixing and CAA hyper-network kernels.
#pragma once
#include "common.h"

// dst[0 .. n_words) = 0 (32-bit words; dst 4-byte aligned) by a KERNEL.  Used instead of hipMemsetAsync wherever the call could end up
// inside a captured graph: on ROCm 7.2 a memset NODE replays pointer-like garbage from the second hipGraphLaunch on
// (tools/repro/graph_memset_node.py, profiles/r04_graph_memset_node_report.txt).
int launch_zero_words(void* dst, long n_words, hipStream_t stream);
// (T,3,H,W) NCHW frames -> (T,H,W,4) pixel-major, 4th channel zero
int launch_pack_lr(const float* lrs, float* lr4, int T, int H, int W, hipStream_t stream);
// (T,3,H,W) partition maps -> the dense equivalent of the reference's sparse_val evaluation (prep.hip)
int launch_par_sparse(const float* par, float* out, int T, int H, int W, hipStream_t stream);
// DCN conv_offset output channel order used by this build (dcn.hip): packed channel c' -> reference channel of
// conv_offset[2] (offsets (g*9+k)*2+{dy,dx}, masks 288+g*9+k; g = deform group, k = tap), -1 = padding.
// Packed order: a lane of the DCN kernel is (pixel, k-half h) and needs groups g = 2q+h, q = 0..7, of one tap at a
// time, so offsets are [tap k][h][q][dy,dx] (c' = 32k + 16h + 2q + e) and mask logits [tap k][h][q] (288 + 16k + 8h + q):
// 64 + 32 contiguous bytes per lane and tap.
static inline __host__ __device__ int pnp_dcn_ref_channel_impl(int c) {
    if (c < 288) {
        const int k = c >> 5, r = c & 31, g = 2 * ((r >> 1) & 7) + (r >> 4);
        return ((g * 9 + k) << 1) + (r & 1);
    }
    if (c < 432) {
        const int k = (c - 288) >> 4, r = (c - 288) & 15, g = 2 * (r & 7) + (r >> 3);
        return 288 + g * 9 + k;
    }
    return -1;
}
// two (H,W) planes -> (H,W,4) pixel-major (x, y, 0, 0): the flow as a conv source
int launch_pack_flow4(const float* fx, const float* fy, float* out4, int H, int W, hipStream_t stream);
// generic layout converters (op-level tests / boundary glue)
int launch_nchw_to_nhwc(const float* in, float* out, int N, int C, int H, int W, hipStream_t stream);
int launch_nhwc_to_nchw(const float* in, float* out, int N, int C, int H, int W, hipStream_t stream);

enum { PACK_WIDE = 0, PACK_RGB4 = 1, PACK_1X1 = 2 };
struct PackArgs {
    const float* w;     // [E][cout_total][cin_total][ktaps]
    const float* ew;    // [E] mixing weights (nullptr: E must be 1)
    int E;
    long e_stride;      // floats between experts
    int cin_total, ktaps;
    int group_cin;      // 0: dense.  > 0: a grouped 64 -> 64 conv (cbase 0) whose weight is [cout][group_cin][ktaps]: packed as the dense conv it
                        // equals, zeros outside the diagonal blocks (nn.Conv2d(groups=64 / group_cin), sr_backbone_utils.py:285-289)
    int co_mul, co_add; // reference output channel = co * co_mul + co_add
    int n_valid;        // packed output channels >= n_valid are zero
    int co_mode;        // 0: affine (co_mul, co_add); 1: DCN offset/mask permutation of packed channel 64*blockIdx.y + co
    int cvalid;         // PACK_RGB4: valid channels of the 4-channel source (3 RGB frame, 2 flow)
    int kind;           // PACK_*
    int cbase;          // first input channel of this source inside the virtual concat
    int ntb;            // N tiles of 32 in the image (2 -> 64 channels, 1 -> 32)
    float scale;        // every packed value is multiplied by this (1: plain; PNP_PAR_UNIT: the 1x1 branch images of the split-fp16 fast path)
    float* dst;         // 9 chunks (PACK_WIDE) or 1 chunk
    long w_ystride;     // blockIdx.y batching: floats between consecutive convs in w / dst
    long dst_ystride;
};
int launch_pack_weights(const PackArgs& a, int grid_y, hipStream_t stream);
// bias_out[y][c] = sum_e ew[e] * b[y][e][c]   (Dynamic_conv2d_se aggregate_bias)
int launch_mix_bias(const float* b, const float* ew, float* out, int E, int C, int nconv, hipStream_t stream);

// CAA hyper-network (Base_Predictor + SEModule) for up to 32 frames per launch.
struct CaaArgs {
    float q_ew[32];     // base_QPs (or QPs) per frame
    float q_g[32];      // QPs per frame
    int count, t0, E, softmax, with_se;
    const float *w1, *b1, *w2, *b2;   // BasePredictor.BaseNet.{0,2}
    const float *v1, *v2;             // BiasePredictor.fc.{0,2} (with_se)
    float* ew;          // [T][E]
    float* gamma;       // [T][64]
};
int launch_caa_predict(const CaaArgs& a, hipStream_t stream);
