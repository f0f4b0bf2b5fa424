// NOT BUILT, NOT SHIPPED.  Kept as the record of a measured experiment (last bullet): it was wired into
// launch_conv3x3_f16 (conv_f16.hip) for single-source launches that write an fp16 map, passed bit-identity tests against the
// resident-weight kernel on 256x512 / 264x520 / 248x1064 / 720x1280 with and without partition branches and fp16 sources, and
// gained 0.5 % of the fp16 720p step (first version: fragments one k-step ahead; this version: three k-steps ahead, ring write at the
// top of the chunk, counted barrier wait -- same timings: 135.6 vs 147.0 us from an fp32 source, 104.1 vs 106.2 from an fp16 one,
// conv_hr 113.5 / 90.7 vs 108.9 / 85.2; K loop 1074 cycles per 16-MFMA chunk either way, see profiles/r03_ub_mfma_issue.txt).  To try it again: add it to build_native.SOURCES, declare launch_conv3x3_f16_wide in
// conv_mfma.h and call it from launch_conv3x3_f16 where `om == 1 && nwide == 1 && lr_idx < 0 && !f.residual`.
//
// fp16-operand 64 -> 64 conv on 8x32-pixel tiles: a wave owns 64 pixels x 64 channels (2 x 2 MFMA tiles), so a k-step is four
// fragment reads (2 A, 2 B) for four MFMAs -- 1 KiB of LDS reads per MFMA instead of the 1.5 KiB of the 32-pixel wave tile of
// conv_f16.hip, whose matrix phase the LDS bounds at 42-60 cycles per MFMA (profiles/r03_ub_lds.txt).
//
// Covers the launches that write an fp16 map (PNP_OPT_F16_MAPS): the front half of a BAE block
//   o = relu(gamma * (conv3x3(x) + b) + sum_j par_j * conv1x1_j(x))          sr_backbone_utils.py:310-311
// and conv_hr (iconvsr_ipb_par.py:144), from an fp32 or an fp16 source map, on frames of >= 1024 8x16 tiles.  Same operands,
// same k order (9 taps x 4 k-steps, then the needed branches in plane order), same rounding points as conv_f16.hip's kernels:
// BIT-IDENTICAL output (tests/test_gpu_fp16.py pins and cross-kernel tests).
//
// One 8x32 tile per 4-wave block, 75.8 KiB of LDS (fp16 A tile 10 x 34 pixels + the 3-slot 8 KiB weight ring) -> two blocks per
// CU: one block's halo wait / epilogue under the other's K loop.  Weight chunks come through a scalar-offset buffer descriptor,
// the fragments of the next k-step are issued before this one's MFMAs (pinned), across the chunk barrier too.
#include "conv_mfma.h"
#include "f16_util.h"

namespace {

constexpr int WTW = 32, WPW = WTW + 2;                       // tile and halo width in pixels
constexpr int WRSB = 5120;                                   // A-tile row stride: 34 x 144 = 4896 -> 20 x 256
constexpr int WA_BYTES = (TH + 2) * WRSB;                    // 51,200
constexpr int W_CHUNK = 8 * UNIT, W_RING = 3;
constexpr int W_LDS = WA_BYTES + W_RING * W_CHUNK;           // 75,776
static_assert(2 * W_LDS <= 160 * 1024, "two blocks per CU");
static_assert(WPW * PSB <= WRSB, "row fits its stride");

struct WideArgs {
    const void* src;             // NHWC64 fp32 or (SRC16) fp16
    const _Float16* w;           // 72 units
    const _Float16* wpar;        // 24 units or nullptr
    const float* par;
    long par_plane;
    const int* par_flags;        // per 8x16 tile (launch_par_tile_flags) or nullptr
    const float *bias, *gamma;
    void* out;                   // fp16 NHWC64
    int H, W, act;
    unsigned long long* dbg;     // 8 u64 per block or nullptr
};

template <bool PAR, bool SRC16>
__global__ __launch_bounds__(256, 2) void conv3x3_f16_wide_kernel(const WideArgs a) {
    constexpr int NC = 9 + (PAR ? 3 : 0);
    constexpr int WPT = 2;
    constexpr int NRQ = SRC16 ? 11 : 22;                     // halo requests of 16 B per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned long long d_t0 = 0, d_t1 = 0, d_t2 = 0;
    if (a.dbg) d_t0 = __builtin_amdgcn_s_memtime();
    const int m = lane & 31, h = lane >> 5, my = m >> 4, mx = m & 15;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + WTW - 1) / WTW;
    int tile;
    {   // XCD-aware remap: each XCD walks a contiguous band of tiles (halo rows meet in its L2)
        const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7;
        const int q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * WTW;
    char* const sR = smem + WA_BYTES;

    constexpr unsigned SPB = SRC16 ? 128u : 256u;            // source bytes per pixel
    const unsigned src_bytes = (unsigned)H * (unsigned)W * SPB;
    const __amdgpu_buffer_rsrc_t r_src = make_rsrc(a.src, src_bytes);
    const __amdgpu_buffer_rsrc_t r_par = make_rsrc(PAR ? (const void*)a.par : a.src, PAR ? (unsigned)(3 * a.par_plane * 4) : 0);
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(a.out, (unsigned)H * (unsigned)W * 128u);
    const int ftx = (W + TW - 1) / TW;                       // the flags are per 8x16 tile: this tile covers two of them
    const __amdgpu_buffer_rsrc_t r_flags = make_rsrc((PAR && a.par_flags) ? (const void*)a.par_flags : a.src,
                                                     (PAR && a.par_flags) ? (unsigned)(ftx * ((H + TH - 1) / TH)) * 4u : 0);
    const __amdgpu_buffer_rsrc_t r_w = make_rsrc(a.w, 9u * W_CHUNK);
    const __amdgpu_buffer_rsrc_t r_wp = make_rsrc(PAR ? (const void*)a.wpar : (const void*)a.w, 3u * W_CHUNK);
    int pfl_a = 0, pfl_b = 0;
    if (PAR) {
        const int fi = (ty0 / TH) * ftx + 2 * (tx0 / WTW);
        pfl_a = __builtin_bit_cast(int, buf_load1(r_flags, (unsigned)fi * 4u));
        pfl_b = __builtin_bit_cast(int, buf_load1(r_flags, 2 * (tx0 / WTW) + 1 < ftx ? (unsigned)(fi + 1) * 4u : OOB));
    }

    // ---- halo requests.  Row-affine mapping: fp32 source: requests 2r, 2r + 1 = row r, pixels 0..15 / 16..31 (thread t: pixel
    //      t >> 4, float4 t & 15), requests 20, 21 the two right-hand columns (item j = t + 256 (k - 20): row j >> 5, pixel
    //      32 + ((j >> 4) & 1)); fp16 source: request r = row r, pixels 0..31 (pixel t >> 3, 16-byte unit t & 7), request 10 the
    //      right-hand columns (row t >> 4, pixel 32 + ((t >> 3) & 1), t < 160)
    f32x4 areg[NRQ];
    const unsigned row_b = (unsigned)W * SPB;
    const unsigned hbase = (unsigned)((ty0 - 1) * W + (tx0 - 1)) * SPB;
    constexpr int UPP = SRC16 ? 8 : 16;                      // 16-byte units per pixel
    const int hp = t / UPP, hu = t % UPP;
    if (SRC16) {
        const bool okm = (unsigned)(tx0 - 1 + hp) < (unsigned)W;
#pragma unroll
        for (int k = 0; k < 10; ++k) areg[k] = buf_load4(r_src, okm ? hbase + (unsigned)k * row_b + (unsigned)hp * 128u + (unsigned)hu * 16u : OOB);
        const int sr = t >> 4, sx = 32 + ((t >> 3) & 1);
        const bool oks = (t < 160) & ((unsigned)(tx0 - 1 + sx) < (unsigned)W);
        areg[10] = buf_load4(r_src, oks ? hbase + (unsigned)sr * row_b + (unsigned)sx * 128u + (unsigned)hu * 16u : OOB);
    } else {
#pragma unroll
        for (int k = 0; k < 20; ++k) {
            const int px = (k & 1) * 16 + hp;
            const bool ok = (unsigned)(tx0 - 1 + px) < (unsigned)W;
            areg[k] = buf_load4(r_src, ok ? hbase + (unsigned)(k >> 1) * row_b + (unsigned)px * 256u + (unsigned)hu * 16u : OOB);
        }
#pragma unroll
        for (int k = 20; k < 22; ++k) {
            const int j = t + 256 * (k - 20);
            const int sr = j >> 5, sx = 32 + ((j >> 4) & 1);
            const bool ok = (sr < TH + 2) & ((unsigned)(tx0 - 1 + sx) < (unsigned)W);
            areg[k] = buf_load4(r_src, ok ? hbase + (unsigned)sr * row_b + (unsigned)sx * 256u + (unsigned)hu * 16u : OOB);
        }
    }
    // ---- weight chunks: chunk c (tap c, or the (c - 9)-th needed branch) in ring slot c % 3
    int ncr = NC, bs0 = 0, bs1 = 1, bs2 = 2;
    auto bsel = [&](int j) { return j == 0 ? bs0 : (j == 1 ? bs1 : bs2); };
    f32x4 wreg[3][WPT];                                  // chunk c travels in set c % 3
    auto request_chunk = [&](int c, int set) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int so = (c < 9 ? c : bsel(c - 9)) * W_CHUNK + i * 4096;
            wreg[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c < 9 ? r_w : r_wp, t * 16, so, 0));
        }
    };
#pragma unroll
    for (int c = 0; c < 3; ++c) request_chunk(c, c);
    // partition values of the lane's two pixels (M tile j: columns 16 j + mx)
    float pv[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (PAR) {
        const int gy = ty0 + 2 * wave + my;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gx = tx0 + 16 * j + mx;
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
                pv[j][jj] = buf_load1(r_par, ((gy < H) & (gx < W)) ? (unsigned)(jj * a.par_plane + (long)gy * W + gx) * 4u : OOB);
        }
    }
    const int n0 = lane & 31;
    const float neg_slope = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.1f);
    float bco[2], gco[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bco[j] = a.bias ? a.bias[j * 32 + n0] : 0.f;
        gco[j] = a.gamma ? a.gamma[j * 32 + n0] : 1.f;
    }
    // ---- halo -> fp16 A tile (pixel stride 144 B)
    if (SRC16) {
#pragma unroll
        for (int k = 0; k < 10; ++k) *reinterpret_cast<f32x4*>(smem + k * WRSB + hp * PSB + hu * 16) = areg[k];
        const int sr = t >> 4, sx = 32 + ((t >> 3) & 1);
        if (t < 160) *reinterpret_cast<f32x4*>(smem + sr * WRSB + sx * PSB + hu * 16) = areg[10];
    } else {
#pragma unroll
        for (int k = 0; k < 20; ++k)
            *reinterpret_cast<h4*>(smem + (k >> 1) * WRSB + ((k & 1) * 16 + hp) * PSB + hu * 8) = to_h4(areg[k]);
#pragma unroll
        for (int k = 20; k < 22; ++k) {
            const int j = t + 256 * (k - 20);
            const int sr = j >> 5, sx = 32 + ((j >> 4) & 1);
            if (sr < TH + 2) *reinterpret_cast<h4*>(smem + sr * WRSB + sx * PSB + hu * 8) = to_h4(areg[k]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(sR + c * W_CHUNK + (t + 256 * i) * 16) = wreg[c][i];
    request_chunk(3, 0);
    lds_barrier();
    if (a.dbg) d_t1 = __builtin_amdgcn_s_memtime();

    // ---- K loop
    const int a_off = (2 * wave + my) * WRSB + mx * PSB + 16 * h;        // M tile j: + 16 j pixels
    f32x16 acc[2][2];                                                     // [M tile][N tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto bias_gamma = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] + bco[j]) * gco[j];
    };
    // One continuous stream of k-steps (4 per chunk; chunk c in ring slot c % 3).  Fragments are fetched DEPTH k-steps ahead of
    // the MFMAs that use them, across chunk boundaries (chunk c + 1 has been in the ring since the barrier that ended chunk
    // c - 1): 4 DEPTH = 12 reads in flight per wave, i.e. 12 MFMAs = 384 cycles of LDS latency covered by a wave on its own.
    // Per chunk: at its top the ring write of chunk c + 2 (into the slot of chunk c - 1, which every wave left before the barrier
    // just passed; requested two chunks ago) and the request for chunk c + 4; at its end one barrier that waits for nothing: LDS
    // operations complete in order, the write is older than the 16 reads the chunk issued, of which at most 12 are outstanding.
    constexpr int DEPTH = 3;
    h8 fa0[DEPTH + 1], fa1[DEPTH + 1], fb0[DEPTH + 1], fb1[DEPTH + 1];      // slot = step % (DEPTH + 1)
    auto fetch = [&](int step) {                   // compile-time step; a branch chunk reads the centre tap
        const int c = step >> 2, sk = step & 3, sl = step % (DEPTH + 1);
        const int dy = c < 9 ? c / 3 : 1, dx = c < 9 ? c % 3 : 1;
        const int o = a_off + dy * WRSB + dx * PSB + 32 * sk;
        const char* b_lane = sR + (c % W_RING) * W_CHUNK + lane * 16;
        fa0[sl] = *reinterpret_cast<const h8*>(smem + o);
        fa1[sl] = *reinterpret_cast<const h8*>(smem + o + 16 * PSB);
        fb0[sl] = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 0) * UNIT);
        fb1[sl] = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 1) * UNIT);
    };
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) fetch(k);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (PAR && c == 5 && a.par_flags) {          // first use of the flags: chunk 9 is requested below
            const int f0 = (__builtin_amdgcn_readfirstlane(pfl_a) | __builtin_amdgcn_readfirstlane(pfl_b)) & 7;
            const int f1 = f0 & (f0 - 1), f2 = f1 & (f1 - 1);
            ncr = 9 + __builtin_popcount(f0);
            bs0 = f0 ? __builtin_ctz(f0) : 0;
            bs1 = f1 ? __builtin_ctz(f1) : 0;
            bs2 = f2 ? __builtin_ctz(f2) : 0;
        }
        if (PAR && c >= 9 && c >= ncr) break;
        if (c + 2 < NC && (!PAR || c + 2 < ncr) && c + 2 >= 3) {     // chunks 0..2 went in before the loop
            char* d = sR + ((c + 2) % W_RING) * W_CHUNK;
#pragma unroll
            for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(d + (t + 256 * i) * 16) = wreg[(c + 2) % 3][i];
        }
        if (c + 4 < NC && (!PAR || c + 4 < ncr)) request_chunk(c + 4, (c + 4) % 3);
        _Float16 pj0 = (_Float16)1.f, pj1 = (_Float16)1.f;
        if (PAR && c >= 9) {
            if (c == 9) bias_gamma();                  // (conv + bias) * gamma BEFORE the 1x1 partition branches
            const int bi = bsel(c - 9);
            pj0 = (_Float16)(bi == 0 ? pv[0][0] : (bi == 1 ? pv[0][1] : pv[0][2]));
            pj1 = (_Float16)(bi == 0 ? pv[1][0] : (bi == 1 ? pv[1][1] : pv[1][2]));
        }
#pragma unroll
        for (int sk = 0; sk < 4; ++sk) {
            const int step = c * 4 + sk, sl = step % (DEPTH + 1);
            if (step + DEPTH < NC * 4) fetch(step + DEPTH);       // (past the tile's last chunk: unused stale bytes)
            __builtin_amdgcn_sched_barrier(0);
            h8 a0 = fa0[sl], a1 = fa1[sl];
            const h8 b0 = fb0[sl], b1 = fb1[sl];
            if (PAR && c >= 9) {
                a0 *= pj0;
                a1 *= pj1;
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(12)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // nobody still reads the LDS the epilogue overwrites
    if (!PAR || ncr == 9) bias_gamma();
    if (a.dbg) d_t2 = __builtin_amdgcn_s_memtime();

    // ---- epilogue, one M tile at a time: transpose through the dead LDS, activation, fp16, 16-byte stores
    float* sT = reinterpret_cast<float*>(smem + wave * 8192);
    const f32x4* sT4 = reinterpret_cast<const f32x4*>(sT);
    const int ec8 = lane & 7, ep8 = lane >> 3;
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sT[((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + j * 32 + n0] = acc[i2][j][r];
        asm volatile("" ::: "memory");
        f32x4 lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = sT4[(ep8 + 8 * i) * 16 + 2 * ec8];
            hi[i] = sT4[(ep8 + 8 * i) * 16 + 2 * ec8 + 1];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 u = lo[i], v = hi[i];
            u = __builtin_elementwise_max(u, (f32x4)(0.f)) + neg_slope * __builtin_elementwise_min(u, (f32x4)(0.f));
            v = __builtin_elementwise_max(v, (f32x4)(0.f)) + neg_slope * __builtin_elementwise_min(v, (f32x4)(0.f));
            const h4 uh = to_h4(u), vh = to_h4(v);
            const h8 pk = __builtin_shufflevector(uh, vh, 0, 1, 2, 3, 4, 5, 6, 7);
            const int gx = tx0 + 16 * i2 + ep8 + 8 * (i & 1);
            const unsigned o = ((unsigned)(ty0 + 2 * wave + (i >> 1)) * (unsigned)W + (unsigned)gx) * 128u + (unsigned)ec8 * 16u;
            buf_store4(r_out, gx < W ? o : OOB, __builtin_bit_cast(f32x4, pk));
        }
    }
    if (a.dbg && t == 0) {
        unsigned long long* d = a.dbg + (size_t)blockIdx.x * 8;
        d[0] = d_t0;
        d[1] = d_t1;
        d[2] = d_t2;
        d[3] = __builtin_amdgcn_s_memtime();
        d[6] = ncr;
    }
}

template <bool PAR, bool SRC16>
int launch_wide(const WideArgs& wa, hipStream_t stream) {
    auto kern = conv3x3_f16_wide_kernel<PAR, SRC16>;
    static PnpPerDevice once;
    const hipError_t attr_err = once.run([&](int, int&) {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
    });
    if (attr_err != hipSuccess) return (int)attr_err;
    const int tiles = ((wa.W + WTW - 1) / WTW) * ((wa.H + TH - 1) / TH);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), W_LDS, stream, wa);
    return (int)hipGetLastError();
}

}  // namespace

int launch_conv3x3_f16_wide(const ConvArgs& a, hipStream_t stream) {
    WideArgs w;
    w.src = a.src[0];
    w.w = reinterpret_cast<const _Float16*>(a.wsrc_h[0]);
    w.wpar = reinterpret_cast<const _Float16*>(a.wpar_h);
    w.par = a.par;
    w.par_plane = a.par_plane;
    w.par_flags = a.par_flags;
    w.bias = a.bias;
    w.gamma = a.gamma;
    w.out = a.out;
    w.H = a.H;
    w.W = a.W;
    w.act = a.act;
    w.dbg = a.dbg;
    const bool s16 = a.src_f16 & 1;
    if (w.wpar) return s16 ? launch_wide<true, true>(w, stream) : launch_wide<true, false>(w, stream);
    return s16 ? launch_wide<false, true>(w, stream) : launch_wide<false, false>(w, stream);
}
