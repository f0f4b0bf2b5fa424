// Layout, weight-packing, expert-mixing and CAA hyper-network kernels.
//
// * pack_weights: reference OIHW conv weights -> the MFMA "B image" of common.h, optionally
//   mixing E expert weight sets first.  With mixing this is Dynamic_conv2d_se's
//   aggregate_weight = mm(softmax_attention, weight)  (mmedit/models/common/sr_backbone_utils.py:198-199),
//   hoisted out of the per-block/per-frame path: the routing weights depend only on the
//   clip's base QP, so it runs once per distinct value per clip instead of 16*2*T times.
// * caa_predict: Base_Predictor / SEModule+Hsigmoid
//   (mmedit/models/backbones/sr_backbones/domain_aware.py:172-183, 201-222).
#include "prep.h"

namespace {

__global__ __launch_bounds__(256) void pack_lr_kernel(const float* __restrict__ lrs, float* __restrict__ lr4,
                                                      long hw, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long t = i / hw, p = i - t * hw;
    const float* s = lrs + t * 3 * hw + p;
    f32x4 v = {s[0], s[hw], s[2 * hw], 0.f};
    reinterpret_cast<f32x4*>(lr4)[i] = v;
}

// sparse_val (eval): basicvsr_net.py:511-514 generate_indices(par_j) + sr_backbone_utils.py:294-302 sparse_conv:
// the 1x1 branch j is evaluated where plane j is NONZERO (whatever its value), later planes overwrite earlier ones
// (mask_roi_back assigns), and the result is divided by 255.  As a dense map: plane j = 1/255 where par_j != 0 and no
// later plane is nonzero, else 0.
__global__ __launch_bounds__(256) void par_sparse_kernel(const float* __restrict__ par, float* __restrict__ out, long hw,
                                                         long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long t = i / hw, p = i - t * hw;
    const float* s = par + t * 3 * hw + p;
    float* d = out + t * 3 * hw + p;
    const bool n0 = s[0] != 0.f, n1 = s[hw] != 0.f, n2 = s[2 * hw] != 0.f;
    const float v = 1.0f / 255.0f;
    d[0] = (n0 && !n1 && !n2) ? v : 0.f;
    d[hw] = (n1 && !n2) ? v : 0.f;
    d[2 * hw] = n2 ? v : 0.f;
}

__global__ __launch_bounds__(256) void pack_flow4_kernel(const float* __restrict__ fx, const float* __restrict__ fy,
                                                         float* __restrict__ out4, long hw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hw) return;
    f32x4 v = {fx[i], fy[i], 0.f, 0.f};
    reinterpret_cast<f32x4*>(out4)[i] = v;
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int C, long hw, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over N*HW*C, c fastest
    if (i >= total) return;
    const int c = (int)(i % C);
    const long np = i / C;
    const long n = np / hw, p = np - n * hw;
    out[i] = in[(n * C + c) * hw + p];
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int C, long hw, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over N*C*HW, p fastest
    if (i >= total) return;
    const long p = i % hw;
    const long nc = i / hw;
    const long n = nc / C;
    const int c = (int)(nc - n * C);
    out[i] = in[(n * hw + p) * C + c];
}

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs a) {
    const int chunk_floats = PNP_CHUNK_Q * a.ntb * 256;
    const int nchunk = (a.kind == PACK_WIDE) ? 9 : 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nchunk * chunk_floats) return;
    const int chunk = idx / chunk_floats;
    const int rem = idx - chunk * chunk_floats;
    const int q = rem / (a.ntb * 256);
    const int nt = (rem >> 8) % a.ntb;
    const int lane = (rem >> 2) & 63;
    const int j = rem & 3;
    const int n = lane & 31, h = lane >> 5;
    const int co = nt * 32 + n;
    int ci, tap;
    bool valid = co < a.n_valid;
    if (a.kind == PACK_WIDE) {
        tap = chunk;
        ci = a.cbase + 8 * q + 4 * h + j;
    } else if (a.kind == PACK_RGB4) {
        tap = 2 * q + h;
        ci = a.cbase + j;
        valid = valid && (q < 5) && (tap <= 8) && (j < a.cvalid);
    } else {
        tap = 0;
        ci = a.cbase + 8 * q + 4 * h + j;
    }
    float v = 0.f;
    int co_ref = co * a.co_mul + a.co_add;
    if (a.co_mode == 1) {
        co_ref = pnp_dcn_ref_channel_impl(blockIdx.y * 64 + co);
        valid = valid && co_ref >= 0;
    }
    int cin_w = a.cin_total;
    if (a.group_cin > 0) {                  // grouped conv: only the diagonal blocks exist
        valid = valid && (ci / a.group_cin == co_ref / a.group_cin);
        ci %= a.group_cin;
        cin_w = a.group_cin;
    }
    if (valid) {
        const float* w = a.w + (long)blockIdx.y * a.w_ystride + ((long)co_ref * cin_w + ci) * a.ktaps + tap;
        if (a.ew) {
            for (int e = 0; e < a.E; ++e) v += a.ew[e] * w[(long)e * a.e_stride];
        } else {
            v = w[0];
        }
    }
    a.dst[(long)blockIdx.y * a.dst_ystride + idx] = v * a.scale;        // (x 1.0f is exact)
}

__global__ void mix_bias_kernel(const float* __restrict__ b, const float* __restrict__ ew, float* __restrict__ out,
                                int E, int C) {
    const int c = threadIdx.x, y = blockIdx.x;
    if (c >= C) return;
    float v = 0.f;
    for (int e = 0; e < E; ++e) v += ew[e] * b[((long)y * E + e) * C + c];
    out[(long)y * C + c] = v;
}

__global__ __launch_bounds__(64) void caa_predict_kernel(const CaaArgs a) {
    __shared__ float hid[64];
    __shared__ float logit[64];
    const int i = blockIdx.x, c = threadIdx.x;
    const float q = a.q_ew[i];
    // Base_Predictor: Linear(1,64) -> ReLU -> Linear(64,E) [-> Softmax]
    hid[c] = fmaxf(q * a.w1[c] + a.b1[c], 0.f);
    __syncthreads();
    if (c < a.E) {
        float s = 0.f;
        for (int k = 0; k < 64; ++k) s += hid[k] * a.w2[c * 64 + k];
        logit[c] = s + a.b2[c];
    }
    __syncthreads();
    if (c < a.E) {
        float v = logit[c];
        if (a.softmax) {
            float mx = logit[0];
            for (int k = 1; k < a.E; ++k) mx = fmaxf(mx, logit[k]);
            float den = 0.f;
            for (int k = 0; k < a.E; ++k) den += expf(logit[k] - mx);
            v = expf(v - mx) / den;
        }
        a.ew[(long)(a.t0 + i) * a.E + c] = v;
    }
    // SEModule: Linear(1,4,no bias) -> ReLU -> Linear(4,64,no bias) -> relu6(x+3)/3
    float g = 1.f;
    if (a.with_se) {
        const float qg = a.q_g[i];
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) s += fmaxf(qg * a.v1[r], 0.f) * a.v2[c * 4 + r];
        g = fminf(fmaxf(s + 3.f, 0.f), 6.f) / 3.f;
    }
    a.gamma[(long)(a.t0 + i) * 64 + c] = g;
}

}  // namespace

int launch_pack_lr(const float* lrs, float* lr4, int T, int H, int W, hipStream_t stream) {
    const long hw = (long)H * W, total = hw * T;
    hipLaunchKernelGGL(pack_lr_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, lrs, lr4, hw,
                       total);
    return (int)hipGetLastError();
}

int launch_par_sparse(const float* par, float* out, int T, int H, int W, hipStream_t stream) {
    const long hw = (long)H * W, total = hw * T;
    hipLaunchKernelGGL(par_sparse_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, par, out, hw,
                       total);
    return (int)hipGetLastError();
}

int launch_pack_flow4(const float* fx, const float* fy, float* out4, int H, int W, hipStream_t stream) {
    const long hw = (long)H * W;
    hipLaunchKernelGGL(pack_flow4_kernel, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, stream, fx, fy, out4, hw);
    return (int)hipGetLastError();
}

int launch_nchw_to_nhwc(const float* in, float* out, int N, int C, int H, int W, hipStream_t stream) {
    const long hw = (long)H * W, total = hw * N * C;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, out, C,
                       hw, total);
    return (int)hipGetLastError();
}

int launch_nhwc_to_nchw(const float* in, float* out, int N, int C, int H, int W, hipStream_t stream) {
    const long hw = (long)H * W, total = hw * N * C;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, out, C,
                       hw, total);
    return (int)hipGetLastError();
}

int launch_pack_weights(const PackArgs& a, int grid_y, hipStream_t stream) {
    if (a.ntb != 1 && a.ntb != 2) return PNP_ERR_BAD_ARG;
    if (!a.ew && a.E != 1) return PNP_ERR_BAD_ARG;
    const int total = ((a.kind == PACK_WIDE) ? 9 : 1) * pnp_chunk_floats(a.ntb);
    hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256, grid_y), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}

int launch_mix_bias(const float* b, const float* ew, float* out, int E, int C, int nconv, hipStream_t stream) {
    if (C > 64) return PNP_ERR_BAD_ARG;
    hipLaunchKernelGGL(mix_bias_kernel, dim3(nconv), dim3(64), 0, stream, b, ew, out, E, C);
    return (int)hipGetLastError();
}

int launch_caa_predict(const CaaArgs& a, hipStream_t stream) {
    if (a.count < 1 || a.count > 32 || a.E > 64) return PNP_ERR_BAD_ARG;
    hipLaunchKernelGGL(caa_predict_kernel, dim3(a.count), dim3(64), 0, stream, a);
    return (int)hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = 0u;
}
}  // namespace

int launch_zero_words(void* dst, long n_words, hipStream_t stream) {
    if (n_words <= 0) return PNP_OK;
    long blocks = (n_words + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<unsigned*>(dst), n_words);
    return (int)hipGetLastError();
}
