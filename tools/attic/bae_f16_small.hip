// NOT BUILT, NOT SHIPPED.  Record of a measured experiment ("measured and not kept" (c)): both halves of a BAE block
// in one launch on frames below 1024 tiles, wired into run_branch (generator.hip) behind an option.  It reproduced the SHA-256 pins of
// the round-2 build and every fp16 test (94 passed), i.e. the arithmetic is the two launches', but it was SLOWER where it was meant
// to help -- 180x320 one clip 2274 vs 2520 frames/s, three clips 2859 vs 3082, 128x128 one clip 3209 vs 4067; only eight concurrent
// 128x128 clips gained (7131 vs 6531) -- because 60 KiB of LDS allow two blocks per CU instead of three, the six front-half M tiles
// load waves 0 and 1 twice as long as waves 2 and 3, and o costs 64 ds_write_b16 per wave.  With three concurrent clips its output
// also differed from the two-launch path in one A/B run (not chased once the timing was in): treat it as unverified under
// concurrency.  struct BaeSmallArgs lived in conv_mfma.h.
//
// One BAE block in ONE launch on small frames (fp16-operand path, PNP_PREC_F16; ResidualBlockNoBNDynamic_drt.forward,
// sr_backbone_utils.py:305-313,329, channel_first):
//     o  = relu(gamma * (conv3x3(x; W2) + b2) + sum_j par_j * conv1x1_j(x))          front half
//     x' = x + g1 * (conv3x3(o; W1) + b1)                                             back half
// Below 1024 tiles a launch is latency-, not throughput-bound (460 blocks of ~8 us each at 180x320), so the two
// launches of a block cost two dispatch + halo + drain latencies and a round trip of `o` through L2.  Here a 4-wave block owns one
// 8x16 tile of x': it contracts the front half on the tile's 10x18 halo region (180 pixels = six 32-pixel M tiles: waves 0 and 1
// take two, waves 2 and 3 one) from a 12x20 fp16 A tile of x, writes o -- rounded to fp16, zero outside the image, exactly what
// the back half of the two-launch path reads -- over the dead x tile in LDS, and contracts the back half from there.  Every
// output pixel sees the same operands in the same k order with the same rounding points as the two launches: BIT-IDENTICAL
// (tests/test_gpu_fp16.py: pins of the round-2 build, PNP_OPT_FUSED_SMALL 0 / 1).  1.4x the front-half MFMAs (halo) for one
// launch, one halo, no `o` map: 59.9 KiB of LDS, two blocks per CU.
#include "conv_mfma.h"
#include "f16_util.h"

namespace {

constexpr int BX_ROWS = TH + 4, BX_PW = TW + 4;              // 12 x 20 pixels of x
constexpr int BX_RSB = 2944;                                 // row stride of the x tile: 20 x 144 = 2880 + 64
constexpr int BX_BYTES = BX_ROWS * BX_RSB;                   // 35,328 >= the o tile (A_BYTES 28,160) and the 32 KiB transposition
constexpr int B_CHUNK = 8 * UNIT, B_RING = 3;
constexpr int B_LDS = BX_BYTES + B_RING * B_CHUNK;           // 59,904
constexpr int B_AIT = (BX_ROWS * BX_PW * 16 + 255) / 256;    // 15 16-byte halo loads per thread
constexpr int ON = (TH + 2) * PW;                            // 180 pixels of o
static_assert(BX_BYTES >= A_BYTES && BX_BYTES >= 4 * 8192 && 2 * (B_LDS + 64) <= 160 * 1024, "LDS plan");

template <int OUT>
__global__ __launch_bounds__(256, 2) void bae_block_f16_small_kernel(const BaeSmallArgs a) {
    constexpr int WPT = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_flags[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m = lane & 31, h = lane >> 5, my = m >> 4, mx = m & 15, n0 = lane & 31;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TW - 1) / TW;
    int tile;
    {   // XCD-aware remap: each XCD walks a contiguous band of tiles (halo rows meet in its L2)
        const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7;
        const int q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    char* const sX = smem;                                   // x tile, then o tile, then the epilogue's transposition slices
    char* const sR = smem + BX_BYTES;

    const unsigned map_bytes = (unsigned)H * (unsigned)W * 256u;
    const unsigned row_bytes = (unsigned)W * 256u;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(a.x, map_bytes);
    const __amdgpu_buffer_rsrc_t r_par = make_rsrc(a.par, (unsigned)(3 * a.par_plane * 4));
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(a.out, map_bytes);
    const __amdgpu_buffer_rsrc_t r_out16 = make_rsrc(OUT == 2 ? a.out16 : (void*)a.out, OUT == 2 ? map_bytes / 2 : 0);
    const __amdgpu_buffer_rsrc_t r_w2 = make_rsrc(a.w2, 9u * B_CHUNK);      // void* images: raw bytes through descriptors
    const __amdgpu_buffer_rsrc_t r_wp = make_rsrc(a.wpar, 3u * B_CHUNK);
    const __amdgpu_buffer_rsrc_t r_w1 = make_rsrc(a.w1, 9u * B_CHUNK);

    // ---- requests: the 12 x 20 halo of x (row-major items: pixel i >> 4, float4 i & 15), the first front-half weight chunks
    f32x4 areg[B_AIT];
    const unsigned hbase = (unsigned)((ty0 - 2) * W + (tx0 - 2)) * 256u;
#pragma unroll
    for (int k = 0; k < B_AIT; ++k) {
        const int i = t + 256 * k;
        const int pix = i >> 4, cs = i & 15;
        const int ry = pix / BX_PW, rx = pix - ry * BX_PW;
        const bool ok = (unsigned)(tx0 - 2 + rx) < (unsigned)W;      // rows outside the image leave the descriptor by themselves
        areg[k] = buf_load4(r_x, ok ? hbase + (unsigned)(ry * W + rx) * 256u + (unsigned)cs * 16u : OOB);
    }
    f32x4 wreg[2][WPT];
    auto request = [&](__amdgpu_buffer_rsrc_t r, int chunk, int set) {
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            wreg[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, t * 16, chunk * B_CHUNK + i * 4096, 0));
    };
    auto ring_write = [&](int slot, int set) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(sR + slot * B_CHUNK + (t + 256 * i) * 16) = wreg[set][i];
    };
    request(r_w2, 0, 0);
    request(r_w2, 1, 1);

    // ---- the lane's front-half pixels: M tile slot s of wave w is region tile w + 4 s (s = 1 only for waves 0, 1); MFMA row m is
    //      region pixel 32 tile + m (row-major over the 10 x 18 region, clamped to the last one beyond 180)
    const bool two = wave < 2;
    int a_front[2];
    float pv[2][3];
    int wmask = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int p = 32 * (wave + 4 * s) + m;
        p = p < ON ? p : ON - 1;
        const int rr = p / PW, cc = p - rr * PW;
        a_front[s] = rr * BX_RSB + cc * PSB + 16 * h;
        const int gy = ty0 - 1 + rr, gx = tx0 - 1 + cc;
        const bool in = ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W) & (s == 0 || two);
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            pv[s][jj] = buf_load1(r_par, in ? (unsigned)(jj * a.par_plane + (long)gy * W + gx) * 4u : OOB);
        }
    }
    float bco2[2], gco2[2], bco1[2], gco1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bco2[j] = a.b2 ? a.b2[j * 32 + n0] : 0.f;
        gco2[j] = a.gamma ? a.gamma[j * 32 + n0] : 1.f;
        bco1[j] = a.b1 ? a.b1[j * 32 + n0] : 0.f;
        gco1[j] = a.g1 ? a.g1[j * 32 + n0] : 1.f;
    }
    // which 1x1 branches the block needs: a plane that is zero on all 180 pixels contributes exact zeros (skipped, value-identical)
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const bool nz = (pv[0][jj] != 0.f) | (pv[1][jj] != 0.f);
        if (__builtin_amdgcn_ballot_w64(nz) != 0) wmask |= 1 << jj;
    }
    if (lane == 0) s_flags[wave] = wmask;

    // ---- x halo -> fp16 A tile (pixel stride 144 B); chunks 0, 1 -> ring
#pragma unroll
    for (int k = 0; k < B_AIT; ++k) {
        const int i = t + 256 * k;
        const int pix = i >> 4, cs = i & 15;
        const int ry = pix / BX_PW, rx = pix - ry * BX_PW;
        *reinterpret_cast<h4*>(sX + ry * BX_RSB + rx * PSB + cs * 8) = to_h4(areg[k]);
    }
    ring_write(0, 0);
    ring_write(1, 1);
    request(r_w2, 2, 0);
    lds_barrier();
    const int f0 = __builtin_amdgcn_readfirstlane((s_flags[0] | s_flags[1] | s_flags[2] | s_flags[3]) & 7);     // block-uniform
    const int f1 = f0 & (f0 - 1), f2 = f1 & (f1 - 1);
    const int nbr = __builtin_popcount(f0);
    const int bs0 = f0 ? __builtin_ctz(f0) : 0, bs1 = f1 ? __builtin_ctz(f1) : 0, bs2 = f2 ? __builtin_ctz(f2) : 0;
    auto bsel = [&](int j) { return j == 0 ? bs0 : (j == 1 ? bs1 : bs2); };
    const int ncf = __builtin_amdgcn_readfirstlane(9 + nbr);            // front-half chunks

    // ---- front half: chunk c (tap c, or the (c - 9)-th needed branch) from ring slot c % 3
    f32x16 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][j][r] = 0.f;
    auto bias_gamma2 = [&]() {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][j][r] = (acc[s][j][r] + bco2[j]) * gco2[j];
    };
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        if (c >= 9 && c >= ncf) break;
        if (c + 3 < 12 && c + 3 < ncf) {
            if (c + 3 < 9) request(r_w2, c + 3, (c + 1) & 1);
            else request(r_wp, bsel(c + 3 - 9), (c + 1) & 1);
        }
        const int dy = c < 9 ? c / 3 : 1, dx = c < 9 ? c % 3 : 1;
        _Float16 pj0 = (_Float16)1.f, pj1 = (_Float16)1.f;
        if (c >= 9) {
            if (c == 9) bias_gamma2();                 // (conv + bias) * gamma BEFORE the 1x1 partition branches
            const int bi = bsel(c - 9);
            pj0 = (_Float16)(bi == 0 ? pv[0][0] : (bi == 1 ? pv[0][1] : pv[0][2]));
            pj1 = (_Float16)(bi == 0 ? pv[1][0] : (bi == 1 ? pv[1][1] : pv[1][2]));
        }
        const char* b_lane = sR + (c % B_RING) * B_CHUNK + lane * 16;
#pragma unroll
        for (int sk = 0; sk < 4; ++sk) {
            const int o = dy * BX_RSB + dx * PSB + 32 * sk;
            h8 a0 = *reinterpret_cast<const h8*>(sX + a_front[0] + o);
            const h8 b0 = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 0) * UNIT);
            const h8 b1 = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 1) * UNIT);
            if (c >= 9) a0 *= pj0;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
            if (two) {
                h8 a1 = *reinterpret_cast<const h8*>(sX + a_front[1] + o);
                if (c >= 9) a1 *= pj1;
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if (c + 2 < 12 && c + 2 < ncf) ring_write((c + 2) % B_RING, c & 1);      // slot of chunk c - 1: every wave left it at the previous barrier
        lds_barrier();
    }
    if (ncf == 9) bias_gamma2();
    // the back half's first weight chunks travel while o is written (every register set is free, the ring is dead after the barrier)
    request(r_w1, 0, 0);
    request(r_w1, 1, 1);

    // ---- o = fp16(relu(.)), zero outside the image, over the dead x tile in the back half's A-tile layout (10 rows x 18 pixels)
    const float neg_slope = 0.f;                             // the reference's ReLU, written like the kernels' generic activation
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (s == 1 && !two) break;
        const int tl = wave + 4 * s;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = 32 * tl + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int rr = p / PW, cc = p - rr * PW;
            const bool in = (p < ON) & ((unsigned)(ty0 - 1 + rr) < (unsigned)H) & ((unsigned)(tx0 - 1 + cc) < (unsigned)W);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v = acc[s][j][r];
                v = fmaxf(v, 0.f) + neg_slope * fminf(v, 0.f);
                v = fminf(fmaxf(v, -65504.f), 65504.f);
                const _Float16 hv = in ? (_Float16)v : (_Float16)0.f;
                if (p < ON) *reinterpret_cast<_Float16*>(sX + rr * RSB + cc * PSB + (j * 32 + n0) * 2) = hv;
            }
        }
    }
    ring_write(0, 0);
    ring_write(1, 1);
    request(r_w1, 2, 0);
    // residual rows of the tile (x itself, fp32) for the epilogue
    const int ec = lane & 15, ep = lane >> 4;
    f32x4 res4[8];
    {
        const unsigned rbase = ((unsigned)((ty0 + 2 * wave) * W + tx0 + ep) * 64u + (unsigned)ec * 4u) * 4u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = tx0 + ep + 4 * (i & 3) < W;
            res4[i] = buf_load4(r_x, ok ? rbase + (unsigned)(i >> 2) * row_bytes + (unsigned)(i & 3) * 1024u : OOB);
        }
    }
    lds_barrier();

    // ---- back half: the small-frame kernel's K loop on the o tile
    const char* a_lane = sX + (2 * wave + my) * RSB + mx * PSB + 16 * h;
    f32x16 bacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) bacc[j][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        if (c + 3 < 9) request(r_w1, c + 3, (c + 1) & 1);
        const int dy = c / 3, dx = c % 3;
        const char* b_lane = sR + (c % B_RING) * B_CHUNK + lane * 16;
#pragma unroll
        for (int sk = 0; sk < 4; ++sk) {
            const h8 av = *reinterpret_cast<const h8*>(a_lane + dy * RSB + dx * PSB + 32 * sk);
            const h8 b0 = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 0) * UNIT);
            const h8 b1 = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 1) * UNIT);
            bacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b0, bacc[0], 0, 0, 0);
            bacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b1, bacc[1], 0, 0, 0);
        }
        if (c + 2 < 9) ring_write((c + 2) % B_RING, c & 1);
        lds_barrier();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) bacc[j][r] = (bacc[j][r] + bco1[j]) * gco1[j];

    // ---- epilogue (the small-frame kernel's): transpose through the dead LDS, (no) activation, + x, whole pixel rows to HBM
    const float id_slope = 1.f;
    float* sT = reinterpret_cast<float*>(smem + wave * 8192);
    const f32x4* sT4 = reinterpret_cast<const f32x4*>(sT);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + j * 32 + n0] = bacc[j][r];
    asm volatile("" ::: "memory");
    f32x4 rows[8];
    h4 hv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rows[i] = sT4[(ep + 4 * i) * 16 + ec];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x4 v = rows[i];
        v = __builtin_elementwise_max(v, (f32x4)(0.f)) + id_slope * __builtin_elementwise_min(v, (f32x4)(0.f));
        v += res4[i];
        const int gx = tx0 + ep + 4 * (i & 3);
        const unsigned o = ((unsigned)(ty0 + 2 * wave + (i >> 2)) * (unsigned)W + (unsigned)gx) * 256u + (unsigned)ec * 16u;
        buf_store4(r_out, gx < W ? o : OOB, v);
        if (OUT == 2) hv[i] = to_h4(v);
    }
    if (OUT == 2) {      // fp16 mirror of the wave's 2 x 16 pixel slice: lanes ec / ec ^ 1 swap one value per pixel pair (DPP quad_perm
        typedef int i32x2 __attribute__((ext_vector_type(2)));         // [1,0,3,2]) so that a lane stores 16 B (conv_f16.hip store_mirror16)
        const bool odd = ec & 1;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const i32x2 send = __builtin_bit_cast(i32x2, odd ? hv[i] : hv[i + 1]);
            i32x2 got;
            got[0] = __builtin_amdgcn_mov_dpp(send[0], 0xB1, 0xF, 0xF, true);
            got[1] = __builtin_amdgcn_mov_dpp(send[1], 0xB1, 0xF, 0xF, true);
            const h4 recv = __builtin_bit_cast(h4, got);
            const h4 lo = odd ? recv : hv[i], hi = odd ? hv[i + 1] : recv;
            const h8 pk = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            const int ii = i + (odd ? 1 : 0);
            const int gx = tx0 + ep + 4 * (ii & 3);
            const unsigned o = ((unsigned)(ty0 + 2 * wave + (ii >> 2)) * (unsigned)W + (unsigned)gx) * 128u + (unsigned)(ec >> 1) * 16u;
            buf_store4(r_out16, gx < W ? o : OOB, __builtin_bit_cast(f32x4, pk));
        }
    }
}

template <int OUT>
int launch_bae(const BaeSmallArgs& a, hipStream_t stream) {
    auto kern = bae_block_f16_small_kernel<OUT>;
    static PnpPerDevice once;
    const hipError_t attr_err = once.run([&](int, int&) {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS);
    });
    if (attr_err != hipSuccess) return (int)attr_err;
    const int tiles = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), B_LDS, stream, a);
    return (int)hipGetLastError();
}

}  // namespace

int launch_bae_block_f16_small(const BaeSmallArgs& a, hipStream_t stream) {
    if (!a.x || !a.w2 || !a.wpar || !a.w1 || !a.par || !a.out || a.H < 1 || a.W < 1) return PNP_ERR_BAD_ARG;
    return a.out16 ? launch_bae<2>(a, stream) : launch_bae<0>(a, stream);
}
