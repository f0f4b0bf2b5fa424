// Split-fp16 ("f16x3") variant of the fused 3x3 conv: fp32-LEVEL results from the fp16 matrix pipe.
//
// Every operand is split into two fp16 numbers, x = hi + lo / 2048 with hi = fp16(x), lo = fp16((x - hi) * 2048), and a product
// is contracted as three MFMAs -- hi*hi into one fp32 accumulator, lo*hi + hi*lo into a second one that is scaled by 1/2048 at
// the end (the scaling keeps the low parts in fp16's normal range; the dropped lo*lo term is 2^-22 relative).  hi + lo carries
// 22 of fp32's 24 significand bits and every fp16 x fp16 product is exact in fp32, so a conv differs from the exact-fp32 MFMA
// kernels by ~1e-6 relative -- inside north_star's 1e-3 gate, which the plain fp16 operands (PNP_PREC_F16: 2e-2 on a clip)
// are not -- at 3 MFMAs of v_mfma_f32_32x32x16_f16 per 16-deep k-step against 8 of v_mfma_f32_32x32x2_f32 at 1/16 of the
// rate: ~5x fewer matrix-pipe cycles than the fp32 path.  Feature maps stay fp32 in HBM (reader and writer both): the mode
// changes no data layout, only how a conv multiplies.  Opt-in (PNP_PREC_F16X3); the default and the headline stay exact fp32.
//
// Kernel: the small-frame fp16 kernel's structure (conv_f16.hip) for every frame size -- one 8x16 tile per 4-wave block, two
// fp16 A tiles (hi, lo) converted from the fp32 halo, weight chunks streamed from L2 through a 3-slot ring: per 3x3 tap (or 1x1
// partition branch) one chunk of hi weights (4 k-steps x {hi*hi, lo*hi} x 2 N tiles) and one of lo weights (4 k-steps x hi*lo
// x 2 N tiles).  A partition branch is summed on its own and scaled by par_j(pixel) on the output side, in fp32.  80.9 KiB of LDS -> two blocks per CU.  LDS reads are 1 KiB per MFMA (6 fragments per 6 MFMAs).
#include "conv_mfma.h"
#include "f16_util.h"

namespace {

constexpr int X3_RING = 3;
constexpr int X3_CHUNK = 8 * UNIT;                            // 4 k-steps x 2 N tiles
constexpr int X3_LDS = 2 * A_BYTES + X3_RING * X3_CHUNK;      // 80,896
static_assert(2 * X3_LDS <= 160 * 1024, "two blocks per CU");
constexpr int X3_AIT = (NPIX * 16 + 255) / 256;               // 12 16-byte halo loads per thread (fp32 source)
constexpr float X3_SCALE = 2048.f, X3_INV = 1.f / 2048.f;

struct X3Args {
    const float* src;            // NHWC64 fp32
    const _Float16 *w_hi, *w_lo; // 72 units each: fp16(w) and fp16((w - hi) * 2048) in the fp16 image layout
    const _Float16 *wpar_hi, *wpar_lo;   // 24 units each or nullptr
    const float* par;
    long par_plane;
    const int* par_flags;
    const float *bias, *gamma, *residual;
    float* out;
    int res_pre;                 // residual is added BEFORE the activation (partial sum of a source chain)
    int H, W, act;
};

template <bool PAR>
__global__ __launch_bounds__(256, 2) void conv3x3_f16x3_kernel(const X3Args a) {
    constexpr int NC = 18 + (PAR ? 6 : 0);                   // chunks: (hi, lo) per tap, then (hi, lo) per partition branch
    constexpr int WPT = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m = lane & 31, h = lane >> 5, my = m >> 4, mx = m & 15;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TW - 1) / TW;
    const int ntiles = tiles_x * ((H + TH - 1) / TH);
    int tile;
    {   // XCD-aware remap: each XCD walks a contiguous band of tiles (halo rows meet in its L2)
        const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7;
        const int q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    char* const sAh = smem;
    char* const sAl = smem + A_BYTES;
    char* const sR = smem + 2 * A_BYTES;

    const unsigned map_bytes = (unsigned)H * (unsigned)W * 256u;
    const __amdgpu_buffer_rsrc_t r_src = make_rsrc(a.src, map_bytes);
    const __amdgpu_buffer_rsrc_t r_res = make_rsrc(a.residual ? (const void*)a.residual : (const void*)a.src, a.residual ? map_bytes : 0);
    const __amdgpu_buffer_rsrc_t r_par = make_rsrc(PAR ? (const void*)a.par : (const void*)a.src, PAR ? (unsigned)(3 * a.par_plane * 4) : 0);
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(a.out, map_bytes);
    const __amdgpu_buffer_rsrc_t r_flags = make_rsrc((PAR && a.par_flags) ? (const void*)a.par_flags : (const void*)a.src,
                                                     (PAR && a.par_flags) ? (unsigned)ntiles * 4u : 0);
    const int pfl_v = PAR ? __builtin_bit_cast(int, buf_load1(r_flags, (unsigned)tile * 4u)) : 0;

    // ---- requests: fp32 halo, the first weight chunks, residual rows / partition values
    f32x4 areg[X3_AIT];
    const unsigned hbase = (unsigned)((ty0 - 1) * W + (tx0 - 1)) * 256u;
#pragma unroll
    for (int k = 0; k < X3_AIT; ++k) {
        const int i = t + 256 * k;
        const int pix = i >> 4, cs = i & 15;
        const int ry = pix / PW, rx = pix - ry * PW;
        const bool ok = (pix < NPIX) & ((unsigned)(tx0 - 1 + rx) < (unsigned)W);  // rows outside the image leave the descriptor by themselves
        areg[k] = buf_load4(r_src, ok ? hbase + (unsigned)(ry * W + rx) * 256u + (unsigned)cs * 16u : OOB);
    }
    const f32x4* whi = reinterpret_cast<const f32x4*>(a.w_hi);
    const f32x4* wlo = reinterpret_cast<const f32x4*>(a.w_lo);
    const f32x4* phi = reinterpret_cast<const f32x4*>(a.wpar_hi);
    const f32x4* plo = reinterpret_cast<const f32x4*>(a.wpar_lo);
    int ncr = NC, bs0 = 0, bs1 = 1, bs2 = 2;
    auto bsel = [&](int j) { return j == 0 ? bs0 : (j == 1 ? bs1 : bs2); };
    auto chunk_ptr = [&](int c) -> const f32x4* {          // 512 float4 per chunk; c < 18 compile-time, branch chunks via bsel
        if (c < 18) return ((c & 1) ? wlo : whi) + (c >> 1) * 512;
        return ((c & 1) ? plo : phi) + bsel((c - 18) >> 1) * 512;
    };
    f32x4 wreg[2][WPT];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const f32x4* g = chunk_ptr(c);
        f32x4 v[WPT];
#pragma unroll
        for (int i = 0; i < WPT; ++i) v[i] = g[t + 256 * i];
#pragma unroll
        for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(sR + c * X3_CHUNK + (t + 256 * i) * 16) = v[i];
    }
    {
        const f32x4* g = chunk_ptr(2);
#pragma unroll
        for (int i = 0; i < WPT; ++i) wreg[0][i] = g[t + 256 * i];
    }
    constexpr int EIT = 8;
    const int ec = lane & 15, ep = lane >> 4, n0 = lane & 31;
    const float neg_slope = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.1f);
    const float k_pre = a.res_pre ? 1.f : 0.f, k_post = 1.f - k_pre;
    float bco[2], gco[2], pv[3] = {0.f, 0.f, 0.f};
    f32x4 res4[EIT];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bco[j] = a.bias ? a.bias[j * 32 + n0] : 0.f;
        gco[j] = a.gamma ? a.gamma[j * 32 + n0] : 1.f;
    }
    const unsigned row_bytes = (unsigned)W * 256u;
    {
        const unsigned rbase = ((unsigned)((ty0 + 2 * wave) * W + tx0 + ep) * 64u + (unsigned)ec * 4u) * 4u;
#pragma unroll
        for (int i = 0; i < EIT; ++i) {
            const bool ok = tx0 + ep + 4 * (i & 3) < W;
            res4[i] = buf_load4(r_res, ok ? rbase + (unsigned)(i >> 2) * row_bytes + (unsigned)(i & 3) * 1024u : OOB);
        }
        if (PAR) {
            const int gy = ty0 + 2 * wave + my, gx = tx0 + mx;
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
                pv[jj] = buf_load1(r_par, ((gy < H) & (gx < W)) ? (unsigned)(jj * a.par_plane + (long)gy * W + gx) * 4u : OOB);
        }
    }
    // ---- fp32 halo -> the two fp16 A tiles: hi = fp16(x) (saturating), lo = fp16((x - hi) * 2048)
#pragma unroll
    for (int k = 0; k < X3_AIT; ++k) {
        const int i = t + 256 * k;
        const int pix = i >> 4, cs = i & 15;
        const int ry = pix / PW, rx = pix - ry * PW;
        if (pix < NPIX) {
            const h4 hi = to_h4(areg[k]);
            const f32x4 rem = (areg[k] - __builtin_convertvector(hi, f32x4)) * X3_SCALE;
            const int o = ry * RSB + rx * PSB + cs * 8;
            *reinterpret_cast<h4*>(sAh + o) = hi;
            *reinterpret_cast<h4*>(sAl + o) = to_h4(rem);
        }
    }
    lds_barrier();

    // ---- K loop: chunk c from ring slot c % 3; even chunks hold hi weights (hi*hi -> acc_hi, lo*hi -> acc_lo), odd ones lo
    //      weights (hi*lo -> acc_lo)
    const int a_off = (2 * wave + my) * RSB + mx * PSB + 16 * h;
    f32x16 acc_hi[2], acc_lo[2], br_hi[2];           // br_hi: the hi*hi sum of ONE partition branch (PAR only)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc_hi[j][r] = 0.f;
            acc_lo[j][r] = 0.f;
            br_hi[j][r] = 0.f;
        }
    auto fold = [&](bool with_bias) {          // acc_hi <- ((acc_hi + acc_lo / 2048) [+ bias]) [* gamma]; acc_lo <- 0
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc_hi[j][r] + acc_lo[j][r] * X3_INV;
                acc_hi[j][r] = with_bias ? (v + bco[j]) * gco[j] : v;
                acc_lo[j][r] = 0.f;
            }
    };
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (PAR && c == 15 && a.par_flags) {         // first use of the flags: chunk 18 is requested below
            const int f0 = __builtin_amdgcn_readfirstlane(pfl_v) & 7;
            const int f1 = f0 & (f0 - 1), f2 = f1 & (f1 - 1);
            ncr = 18 + 2 * __builtin_popcount(f0);
            bs0 = f0 ? __builtin_ctz(f0) : 0;
            bs1 = f1 ? __builtin_ctz(f1) : 0;
            bs2 = f2 ? __builtin_ctz(f2) : 0;
        }
        if (PAR && c >= 18 && c >= ncr) break;
        if (c + 3 < NC && (!PAR || c + 3 < ncr)) {
            const f32x4* g = chunk_ptr(c + 3);
#pragma unroll
            for (int i = 0; i < WPT; ++i) wreg[(c + 1) & 1][i] = g[t + 256 * i];
        }
        const char* b_lane = sR + (c % X3_RING) * X3_CHUNK + lane * 16;
        const int tap = c >> 1, dy = c < 18 ? tap / 3 : 1, dx = c < 18 ? tap % 3 : 1;
        const bool lo_w = c & 1;
        if (PAR && c == 18) fold(true);                // (conv + bias) * gamma BEFORE the 1x1 partition branches
#pragma unroll
        for (int sk = 0; sk < 4; ++sk) {
            const int o = a_off + dy * RSB + dx * PSB + 32 * sk;
            const h8 ah = *reinterpret_cast<const h8*>(sAh + o);
            const h8 b0 = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 0) * UNIT);
            const h8 b1 = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 1) * UNIT);
            if (!lo_w) {
                const h8 al = *reinterpret_cast<const h8*>(sAl + o);
                if (!PAR || c < 18) {
                    acc_hi[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0, acc_hi[0], 0, 0, 0);
                    acc_hi[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1, acc_hi[1], 0, 0, 0);
                } else {
                    br_hi[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0, br_hi[0], 0, 0, 0);
                    br_hi[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1, br_hi[1], 0, 0, 0);
                }
                acc_lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b0, acc_lo[0], 0, 0, 0);
                acc_lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1, acc_lo[1], 0, 0, 0);
            } else {
                acc_lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0, acc_lo[0], 0, 0, 0);
                acc_lo[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1, acc_lo[1], 0, 0, 0);
            }
        }
        if (PAR && c >= 18 && lo_w) {                  // one branch done: out += par_j(pixel) * conv1x1_j(x), scaled on the OUTPUT
            const int bi = bsel((c - 18) >> 1);        // side (sr_backbone_utils.py:310-311 multiplies after the conv, too)
            const float pmine = bi == 0 ? pv[0] : (bi == 1 ? pv[1] : pv[2]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {             // accumulator register r holds pixel row (r&3) + 8*(r>>2) + 4*h: lane `row` has its value
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float pr = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(row * 4, __builtin_bit_cast(int, pmine)));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc_hi[j][r] += pr * (br_hi[j][r] + acc_lo[j][r] * X3_INV);
                    br_hi[j][r] = 0.f;
                    acc_lo[j][r] = 0.f;
                }
            }
        }
        if (c + 2 < NC && (!PAR || c + 2 < ncr)) {
            char* d = sR + ((c + 2) % X3_RING) * X3_CHUNK;     // slot of chunk c - 1: every wave left it at the previous barrier
#pragma unroll
            for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(d + (t + 256 * i) * 16) = wreg[c & 1][i];
        }
        lds_barrier();
    }
    fold(!PAR || ncr == 18);         // PAR with branches: bias / gamma went in before them; otherwise here

    // ---- epilogue: transpose through the dead LDS, [+ partial sum], activation, [+ residual], whole pixel rows to HBM
    float* sT = reinterpret_cast<float*>(smem + wave * 8192);
    const f32x4* sT4 = reinterpret_cast<const f32x4*>(sT);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + j * 32 + n0] = acc_hi[j][r];
    asm volatile("" ::: "memory");
    f32x4 rows[EIT];
#pragma unroll
    for (int i = 0; i < EIT; ++i) rows[i] = sT4[(ep + 4 * i) * 16 + ec];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < EIT; ++i) {
        f32x4 v = rows[i] + k_pre * res4[i];
        v = __builtin_elementwise_max(v, (f32x4)(0.f)) + neg_slope * __builtin_elementwise_min(v, (f32x4)(0.f));
        v += k_post * res4[i];
        const int gx = tx0 + ep + 4 * (i & 3);
        const unsigned o = ((unsigned)(ty0 + 2 * wave + (i >> 2)) * (unsigned)W + (unsigned)gx) * 256u + (unsigned)ec * 16u;
        buf_store4(r_out, gx < W ? o : OOB, v);
    }
}

// fp32 B image -> the LOW fp16 image of the split: fp16((w - fp16(w)) * 2048), same element order as f16_image_kernel
__global__ __launch_bounds__(256) void f16_lo_image_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int ntb, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int per_chunk = PNP_CHUNK_Q * ntb * 256;
    const long chunk = idx / per_chunk;
    const int rem = (int)(idx - chunk * per_chunk);
    const int j = rem & 7, lane = (rem >> 3) & 63, nt = (rem >> 9) % ntb, s = rem / (512 * ntb);
    const int n = lane & 31, hh = lane >> 5;
    const int k = 16 * s + 8 * hh + j;
    const float v = src[chunk * per_chunk + (((k >> 3) * ntb + nt) * 64 + ((k >> 2) & 1) * 32 + n) * 4 + (k & 3)];
    const _Float16 hi = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
    dst[idx] = (_Float16)fminf(fmaxf((v - (float)hi) * X3_SCALE, -65504.f), 65504.f);
}

template <bool PAR>
int launch_x3(const X3Args& xa, hipStream_t stream) {
    auto kern = conv3x3_f16x3_kernel<PAR>;
    static PnpPerDevice once;
    const hipError_t attr_err = once.run([&](int, int&) {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
    });
    if (attr_err != hipSuccess) return (int)attr_err;
    const int tiles = ((xa.W + TW - 1) / TW) * ((xa.H + TH - 1) / TH);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), X3_LDS, stream, xa);
    return (int)hipGetLastError();
}

}  // namespace

int launch_f16_lo_image(const float* src, void* dst, int nchunks, int ntb, hipStream_t stream) {
    if (nchunks < 1 || (ntb != 1 && ntb != 2)) return PNP_ERR_BAD_ARG;
    const long total = (long)nchunks * pnp_chunk_floats(ntb);
    hipLaunchKernelGGL(f16_lo_image_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src,
                       reinterpret_cast<_Float16*>(dst), ntb, total);
    return (int)hipGetLastError();
}

// A conv over several sources runs as a chain that accumulates through `out` (fp32 partial sums, added before the activation
// of the last link): the RGB frame first, on the exact fp32 kernel, then one split launch per 64-channel source.
int launch_conv3x3_f16x3(const ConvArgs& a, int cfg, hipStream_t stream) {
    int lr_idx = -1, wide[4], nwide = 0;
    for (int s = 0; s < a.nsrc; ++s) {
        if (a.src_c[s] == 4) lr_idx = s;
        else wide[nwide++] = s;
    }
    bool have_partial = false;
    if (lr_idx >= 0) {
        ConvArgs r = a;
        r.prec = 0;
        r.nsrc = 1;
        r.src[0] = a.src[lr_idx];
        r.src_c[0] = 4;
        r.wsrc[0] = a.wsrc[lr_idx];
        r.act = 0;
        r.residual = nullptr;
        const int rc = launch_conv3x3(r, cfg, 1, stream);
        if (rc) return rc;
        have_partial = true;
    }
    for (int k = 0; k < nwide; ++k) {
        const bool last = k == nwide - 1;
        X3Args x;
        x.src = a.src[wide[k]];
        x.w_hi = reinterpret_cast<const _Float16*>(a.wsrc_h[wide[k]]);
        x.w_lo = reinterpret_cast<const _Float16*>(a.wsrc_l[wide[k]]);
        x.wpar_hi = reinterpret_cast<const _Float16*>(a.wpar_h);
        x.wpar_lo = reinterpret_cast<const _Float16*>(a.wpar_l);
        x.par = a.par;
        x.par_plane = a.par_plane;
        x.par_flags = a.par_flags;
        x.bias = have_partial ? nullptr : a.bias;
        x.gamma = a.gamma;
        x.residual = have_partial ? a.out : (nwide == 1 ? a.residual : nullptr);
        x.res_pre = have_partial ? 1 : 0;
        x.out = a.out;
        x.H = a.H;
        x.W = a.W;
        x.act = last ? a.act : 0;
        const int rc = x.wpar_hi ? launch_x3<true>(x, stream) : launch_x3<false>(x, stream);
        if (rc) return rc;
        have_partial = true;
    }
    return PNP_OK;
}
