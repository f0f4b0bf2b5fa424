// 3x3 convolution (+ fused per-pixel-weighted 1x1 branches) as an implicit GEMM on
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD on gfx950).
//
// Restates, as ONE kernel family, the reference's
//   * input_conv + LeakyReLU           mmedit/models/backbones/sr_backbones/basicvsr_net.py:484,515
//     over the *virtual* concat [lr, key_warp, neighbor(, bwd)]  iconvsr_ipb_par.py:90,125
//   * BAE block front half             mmedit/models/common/sr_backbone_utils.py:310-311
//       relu( gamma * (conv3x3(x; Wagg) + bagg) + sum_j par_j * conv1x1_j(x) )
//   * BAE block back half              sr_backbone_utils.py:313,329   x + conv1(o)
//   * conv_hr / conv_last heads        iconvsr_ipb_par.py:144-146
//   * PixelShufflePack convs           mmedit/models/common/upsample.py:49-50
//
// GEMM view:  M = pixels, N = output channels, K = (source, tap, input channel).
// Feature maps are pixel-major (NHWC, 64 fp32 = 256 B per pixel) so that a halo tile is a
// handful of fully coalesced row reads and an A fragment is one ds_read_b128.
//
// Block = 256 threads = 4 waves.  Each wave owns one 32-pixel M tile (2 rows x 16 columns)
// and NT 32-channel N tiles.  Per q-step a lane reads ONE float4 of A (4 consecutive input
// channels of its pixel) and NT float4 of B and issues 4*NT MFMAs: the K order inside a
// q-step is permuted identically for A and B (common.h) so no shuffles are needed.
//
// LDS:  A tile  (TH+2) x 18 pixels x (256 + 16) B: the 16-byte pad makes 16 consecutive columns hit
//       16 disjoint 4-bank groups (conflict-free ds_read_b128) AND keeps every fragment address of
//       the K loop a per-lane base + compile-time immediate (no address VALU between MFMAs);
//       B chunks 2 x (8 q-steps x NTB x 1 KiB), double-buffered through registers.
//       8x16 tile: 46080 + 32768 = 78848 B  -> two blocks per CU (2 waves per SIMD), so one
//       block's halo fill / epilogue overlaps the other's MFMA stream.
//
// What the r01 timeline (tools/trace_conv.py) showed and this structure answers:
//   * a CU retires roughly one wave-level VMEM instruction per ~130 cycles whatever its
//     width, so every global access here is 16 B per lane: the accumulator tile is
//     transposed through the (by then free) A region of LDS and stored / residual-added as
//     whole 256-B pixel rows (8 instead of 32 stores per wave);
//   * fragment reads run one q-step ahead of the MFMAs and continue ACROSS the per-chunk
//     barrier: chunk c+1's B image is written to the other buffer mid-chunk, the single
//     barrier sits at the top of the last q-step (after this chunk's last LDS read, before
//     the first read of the next buffer), so no LDS latency is exposed at chunk seams.
#include "conv_mfma.h"
#include <mutex>
#include <cstdlib>
#include <type_traits>

namespace {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int N>
using T = std::integral_constant<int, N>;

template <int WM, int WN, int NT, int NTB>
struct ConvCfg {
    static constexpr int TH = 2 * WM;
    static constexpr int TW = 16;
    static constexpr int PW = TW + 2;
    static constexpr int PIX = (TH + 2) * PW;
    static constexpr int CH4 = PNP_CHUNK_Q * NTB * 64;      // float4 per chunk
    static constexpr int PSTR = 17;                         // float4 per LDS pixel: 256 B + 16 B pad
    static constexpr int LDS_BYTES = (PIX * PSTR + 2 * CH4) * 16;
};

template <int WM, int WN, int NT, int NTB>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(const ConvArgs a) {
    using C = ConvCfg<WM, WN, NT, NTB>;
    constexpr int TH = C::TH, TW = C::TW, PW = C::PW, PIX = C::PIX, CH4 = C::CH4, PSTR = C::PSTR;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(NT * WN == NTB, "N tiles");
    static_assert(4 * 32 * NT * 32 * 4 <= PIX * 256, "transpose region must fit the A tile");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* sA = reinterpret_cast<f32x4*>(smem_raw);
    f32x4* sB = sA + PIX * PSTR;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // MFMA row m of the wave's 2 x 16 pixel slice: row 1 is rotated by two pixels (m = 16 + i is pixel (i - 2) mod 16) so
    // that the two rows of a ds_read_b128 lane group fall on disjoint banks (see conv_persist.hip)
    const int m = lane & 31, h = lane >> 5, my = m >> 4, mx = my ? ((m + 14) & 15) : (m & 15);
    const int H = a.H, W = a.W;

    // ---- tile id, remapped so that each XCD (blocks b, b+8, ...) walks a contiguous band
    const int tiles_x = (W + TW - 1) / TW;
    int tile;
    {
        const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7;
        const int q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    const int yimg = blockIdx.y;
    const long yoff = (long)yimg * a.w_ystride;

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // diagnostic timeline (a.dbg != nullptr only from tools/trace_conv.py): shader-clock stamps
    unsigned long long dbg_t0 = 0, dbg_t1 = 0, dbg_t2 = 0, dbg_e1 = 0, dbg_r0 = 0;
    if (a.dbg) {
        dbg_t0 = __builtin_amdgcn_s_memtime();
        dbg_r0 = __builtin_amdgcn_s_memrealtime();
    }
    // memory phases (halo fill, epilogue) outrank the co-resident block's MFMA stream for issue slots
    __builtin_amdgcn_s_setprio(3);

    // ---- row-wise epilogue geometry: a lane owns 16 B (4 channels) of one pixel per iteration
    constexpr int CW = NT * 8;        // float4 per pixel inside this wave's N range
    constexpr int PPI = 64 / CW;      // pixels per wave instruction
    constexpr int EIT = 32 / PPI;     // iterations to cover the wave's 32 pixels
    const int ec = lane % CW, ep = lane / CW;
    const int n0 = lane & 31;
    const bool rowwise = a.out_mode < 2 || a.out_mode == 4;
    // activation as ONE slope for negative values (a per-element runtime switch compiles to scalar branches)
    const float neg_slope = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.1f);

    // ---- epilogue operands are fetched EARLY (right behind the first halo-tile loads) so their
    //      latency hides under the K loop; the residual may alias `out` element-for-element.
    float bco[NT], gco[NT], pv[3] = {0.f, 0.f, 0.f};
    f32x4 res4[EIT];
    auto prefetch_epilogue_operands = [&]() {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = (wn * NT + j) * 32 + n0;
            bco[j] = a.bias ? a.bias[yimg * a.bias_ystride + co] : 0.f;
            gco[j] = a.gamma ? a.gamma[co] : 1.f;
        }
#pragma unroll
        for (int i = 0; i < EIT; ++i) {
            const int p = ep + i * PPI;
            const int gy = ty0 + 2 * wm + (p >> 4), gx = tx0 + (p & 15);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (a.residual && gy < H && gx < W)
                v = *reinterpret_cast<const f32x4*>(a.residual + ((long)gy * W + gx) * 64 + wn * NT * 32 + ec * 4);
            res4[i] = v;
        }
        if (a.wpar) {
            const int gy = ty0 + 2 * wm + my, gx = tx0 + mx;
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
                pv[jj] = (gy < H && gx < W) ? a.par[jj * a.par_plane + (long)gy * W + gx] : 0.f;
        }
    };

    constexpr int BPT = CH4 / 256;
    f32x4 breg[BPT];
    auto load_b = [&](const f32x4* g) {
#pragma unroll
        for (int i = 0; i < BPT; ++i) breg[i] = g[t + 256 * i];
    };
    auto store_b = [&](int buf) {
        f32x4* d = sB + buf * CH4;
#pragma unroll
        for (int i = 0; i < BPT; ++i) d[t + 256 * i] = breg[i];
    };
    const int bofs = (wn * NT) * 64 + lane;     // this lane's float4 inside a q-step of a B chunk

    // ---- halo tile staging (global -> registers -> swizzled LDS), split so that the loads of the
    //      first tile, of weight chunk 0 and of the epilogue operands are all in flight together
    constexpr int AIT = (PIX * 16 + 255) / 256;
    f32x4 areg[AIT];
    auto stage_wide_load = [&](const float* sp) {
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int i = t + 256 * k;
            const int pix = i >> 4, c16 = i & 15;
            const int ry = pix / PW, rx = pix - ry * PW;
            const int gy = ty0 - 1 + ry, gx = tx0 - 1 + rx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < PIX * 16 && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const f32x4*>(sp + ((long)gy * W + gx) * 64 + c16 * 4);
            areg[k] = v;
        }
    };
    auto stage_wide_store = [&]() {
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int i = t + 256 * k;
            const int pix = i >> 4, c16 = i & 15;
            if (i < PIX * 16) sA[pix * PSTR + c16] = areg[k];
        }
    };
    auto stage_rgb = [&](const float* sp) {
        for (int i = t; i < PIX; i += 256) {
            const int ry = i / PW, rx = i - ry * PW;
            const int gy = ty0 - 1 + ry, gx = tx0 - 1 + rx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const f32x4*>(sp + ((long)gy * W + gx) * 4);
            sA[i] = v;
        }
    };

    // A-fragment addresses: lane base + immediates (tap and q are compile-time everywhere)
    const f32x4* a_lane = sA + ((2 * wm + my) * PW + mx) * PSTR + h;
    const f32x4* r_lane = sA + (2 * wm + my) * PW + mx;          // 4-channel frame: one float4 per pixel
    auto a_wide = [&](int tap, int q) -> f32x4 {
        const int dy = tap / 3, dx = tap - dy * 3;
        return a_lane[(dy * PW + dx) * PSTR + 2 * q];
    };
    auto a_rgb = [&](int q) -> f32x4 {
        const int tp0 = 2 * q, tp1 = (2 * q + 1 > 8) ? 8 : 2 * q + 1;
        const int o0 = (tp0 / 3) * PW + tp0 % 3, o1 = (tp1 / 3) * PW + tp1 % 3;
        return r_lane[h ? o1 : o0];
    };

    // ---- one chunk: NQ q-steps of buffer `cbuf`; the fragments (av,bv) of its first q-step are
    //      live on entry, those of the NEXT chunk's first q-step on exit (when there is one).
    //      `next` = global image of the following chunk (nullptr: none).
    //      KIND 0: wide source, tap TAP; 1: rgb source; 2: 1x1 branch (centre tap, A scaled by ps)
    //      NEXT_TAP: tap whose first A fragment is prefetched after the barrier (-1: sA will be
    //      restaged first / nothing follows)
    // Scheduling: sched_barrier(0) fences each q-step (otherwise hipcc sinks every prefetch back to
    // its first use); inside, sched_group_barrier deals ONE non-MFMA instruction into each of the
    // first MFMA gaps -- measured (tools/ubench/ub_kloop.hip): bunching the ds_reads / address VALU
    // between two q-steps costs +6..12 cycles per 64-cycle MFMA for a wave alone on its SIMD.
    int cbuf = 0;
    f32x4 av, bv[NT];
    auto run_chunk = [&](auto kind_c, auto nq_c, auto tap_c, auto ntap_c, float ps, const f32x4* next) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr int NQ = decltype(nq_c)::value;
        constexpr int TAP = decltype(tap_c)::value;
        constexpr int NEXT_TAP = decltype(ntap_c)::value;
        const f32x4* bb = sB + cbuf * CH4 + bofs;
        const f32x4* bn_base = sB + (cbuf ^ 1) * CH4 + bofs;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            f32x4 an = av, bn[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bn[j] = bv[j];
            if (q == NQ - 1) __syncthreads();   // after this chunk's last LDS read, before the next buffer's first
            if (q == 0 && next) load_b(next);
            if (q < NQ - 1) {
                an = (KIND == 1) ? a_rgb(q + 1) : a_wide(TAP, q + 1);
#pragma unroll
                for (int j = 0; j < NT; ++j) bn[j] = bb[((q + 1) * NTB + j) * 64];
            } else if (next) {
                if (NEXT_TAP >= 0) an = a_wide(NEXT_TAP, 0);
#pragma unroll
                for (int j = 0; j < NT; ++j) bn[j] = bn_base[j * 64];
            }
            f32x4 ax = av;
            if (KIND == 2) ax *= ps;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = mfma32(ax[kk], bv[j][kk], acc[j]);
            if (q == NQ - 2 && next) store_b(cbuf ^ 1);
            // interleave: MFMA, ds_read, MFMA, ds_read, ... then the rest
#pragma unroll
            for (int g = 0; g < 1 + NT; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
            }
            if (q == 0) {
#pragma unroll
                for (int g = 0; g < BPT; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
                }
            }
            if (q == NQ - 2) {
#pragma unroll
                for (int g = 0; g < BPT; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 DS write
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            av = an;
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = bn[j];
        }
        cbuf ^= 1;
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using Q8 = std::integral_constant<int, 8>;
    using Q5 = std::integral_constant<int, 5>;
    // (tap constants are T<n>)

    // ---- prologue: halo tile of source 0, weight chunk 0 and the epilogue operands are requested
    //      back to back (one memory latency instead of three), then written to LDS in that order
    const bool rgb0 = (a.src_c[0] != 64);
    if (rgb0) stage_rgb(a.src[0]);
    else stage_wide_load(a.src[0]);
    load_b(reinterpret_cast<const f32x4*>(a.wsrc[0] + yoff));
    prefetch_epilogue_operands();
    if (!rgb0) stage_wide_store();
    store_b(0);

    // ---- which 1x1 partition branches this tile needs (flags are per 8x16 tile; a 4x16 tile uses its parent's)
    int par_need = a.wpar ? 7 : 0;
    if (a.wpar && a.par_flags) par_need = a.par_flags[(ty0 >> 3) * ((W + 15) >> 4) + (tx0 >> 4)] & 7;
    const int par_cnt = (par_need & 1) + ((par_need >> 1) & 1) + ((par_need >> 2) & 1);
    const int par_j0 = (par_need & 1) ? 0 : ((par_need & 2) ? 1 : 2);
    const int par_j1 = ((par_need & 3) == 3) ? 1 : 2;

    // ---- sources (only source 0 may be the 4-channel frame); chunk images are 9 (wide) or 1 (rgb)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s < a.nsrc) {
            const bool is_rgb = (s == 0) && rgb0;
            const f32x4* wb = reinterpret_cast<const f32x4*>(a.wsrc[s] + yoff);
            const f32x4* after = nullptr;      // first chunk after this source
            if (s + 1 < a.nsrc) after = reinterpret_cast<const f32x4*>(a.wsrc[s + 1 < 4 ? s + 1 : 3] + yoff);
            else if (par_cnt) after = reinterpret_cast<const f32x4*>(a.wpar + yoff) + (long)par_j0 * CH4;
            // sA is free here: every wave passed the last chunk's barrier after its final A read
            if (s > 0) {
                __builtin_amdgcn_s_setprio(3);
                stage_wide_load(a.src[s]);
                stage_wide_store();
            }
            __syncthreads();
            __builtin_amdgcn_s_setprio(0);
            if (a.dbg && s == 0) dbg_t1 = __builtin_amdgcn_s_memtime();
            av = is_rgb ? a_rgb(0) : a_wide(0, 0);
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = sB[cbuf * CH4 + bofs + j * 64];
            if (is_rgb) {
                run_chunk(K1{}, Q5{}, T<0>{}, T<-1>{}, 1.f, after);
            } else {
                const bool par_next = (s + 1 >= a.nsrc) && par_cnt;
                run_chunk(K0{}, Q8{}, T<0>{}, T<1>{}, 1.f, wb + 1L * CH4);
                run_chunk(K0{}, Q8{}, T<1>{}, T<2>{}, 1.f, wb + 2L * CH4);
                run_chunk(K0{}, Q8{}, T<2>{}, T<3>{}, 1.f, wb + 3L * CH4);
                run_chunk(K0{}, Q8{}, T<3>{}, T<4>{}, 1.f, wb + 4L * CH4);
                run_chunk(K0{}, Q8{}, T<4>{}, T<5>{}, 1.f, wb + 5L * CH4);
                run_chunk(K0{}, Q8{}, T<5>{}, T<6>{}, 1.f, wb + 6L * CH4);
                run_chunk(K0{}, Q8{}, T<6>{}, T<7>{}, 1.f, wb + 7L * CH4);
                run_chunk(K0{}, Q8{}, T<7>{}, T<8>{}, 1.f, wb + 8L * CH4);
                if (par_next) run_chunk(K0{}, Q8{}, T<8>{}, T<4>{}, 1.f, after);
                else run_chunk(K0{}, Q8{}, T<8>{}, T<-1>{}, 1.f, after);
            }
        }
    }

    // ---- (conv + bias) * gamma, then the fused 1x1 partition branches as a K extension
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = (acc[j][r] + bco[j]) * gco[j];
    if (par_cnt) {      // only the branches whose plane is nonzero somewhere in the tile (ConvArgs::par_flags)
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpar + yoff);
        int cur = par_j0;
        for (int i = 0; i + 1 < par_cnt; ++i) {
            const int nxt = i == 0 ? par_j1 : 2;
            run_chunk(K2{}, Q8{}, T<4>{}, T<4>{}, cur == 0 ? pv[0] : (cur == 1 ? pv[1] : pv[2]), wp + (long)nxt * CH4);
            cur = nxt;
        }
        run_chunk(K2{}, Q8{}, T<4>{}, T<-1>{}, cur == 0 ? pv[0] : (cur == 1 ? pv[1] : pv[2]), nullptr);
    }
    if (a.dbg) dbg_t2 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_setprio(3);

    // ---- epilogue.  Accumulator register r of lane (n0,h) is pixel m = (r&3) + 8*(r>>2) + 4*h of the
    //      wave's M tile, channel n0 of N tile j.
    if (rowwise) {
        // transpose through the wave's private 32 px x (NT*32) ch slice of the (now free) A region:
        // every wave passed the last chunk's barrier after its final read of sA
        float* sT = reinterpret_cast<float*>(sA) + wave * (32 * NT * 32);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[j][r];
                v = fmaxf(v, 0.f) + neg_slope * fminf(v, 0.f);     // branch-free none / relu / leaky-relu
                const int mm = (r & 3) + 8 * (r >> 2) + 4 * h;                  // MFMA row -> pixel of the slice
                sT[((r >> 3) ? 16 + ((mm + 14) & 15) : mm) * (NT * 32) + j * 32 + n0] = v;
            }
        asm volatile("" ::: "memory");      // keep the row reads below behind the column writes above
        const f32x4* sT4 = reinterpret_cast<const f32x4*>(sT);
#pragma unroll
        for (int i = 0; i < EIT; ++i) {
            const int p = ep + i * PPI;
            const int gy = ty0 + 2 * wm + (p >> 4), gx = tx0 + (p & 15);
            const f32x4 v = sT4[p * CW + ec] + res4[i];
            if (gy < H && gx < W) {
                long o;
                if (a.out_mode == 0) o = ((long)gy * W + gx) * 64;
                else if (a.out_mode == 4) o = ((long)gy * W + gx) * a.out_cstride + 64 * yimg;
                else o = ((long)(2 * gy + (yimg >> 1)) * (2 * W) + 2 * gx + (yimg & 1)) * 64;
                *reinterpret_cast<f32x4*>(a.out + o + wn * NT * 32 + ec * 4) = v;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = (wn * NT + j) * 32 + n0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int mm = (r >> 3) ? 16 + ((mrow + 14) & 15) : mrow;       // MFMA row -> pixel of the slice
                const int gy = ty0 + 2 * wm + (mm >> 4), gx = tx0 + (mm & 15);
                if (gy >= H || gx >= W || co >= 3) continue;
                float v = acc[j][r];
                v = fmaxf(v, 0.f) + neg_slope * fminf(v, 0.f);     // branch-free none / relu / leaky-relu
                float base;
                if (a.out_mode == 2) {
                    base = a.lr[co * a.lr_plane + (long)gy * W + gx];
                } else {
                    // F.interpolate(scale_factor=4, bilinear, align_corners=False)
                    // (iconvsr_ipb_par.py:41,140): src = (dst + 0.5) / 4 - 0.5, clamped at 0
                    const int lh = H >> 2, lw = W >> 2;
                    float sy = (gy + 0.5f) * 0.25f - 0.5f, sx = (gx + 0.5f) * 0.25f - 0.5f;
                    sy = sy < 0.f ? 0.f : sy;
                    sx = sx < 0.f ? 0.f : sx;
                    const int y0 = (int)sy, x0 = (int)sx;
                    const int y1 = y0 + (y0 < lh - 1 ? 1 : 0), x1 = x0 + (x0 < lw - 1 ? 1 : 0);
                    const float ly = sy - y0, lx = sx - x0;
                    const float* p = a.lr + co * a.lr_plane;
                    const float v00 = p[(long)y0 * lw + x0], v01 = p[(long)y0 * lw + x1];
                    const float v10 = p[(long)y1 * lw + x0], v11 = p[(long)y1 * lw + x1];
                    base = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
                }
                a.out[(long)co * H * W + (long)gy * W + gx] = v + base;
            }
        }
    }
    if (a.dbg) dbg_e1 = __builtin_amdgcn_s_memtime();
    if (a.dbg && t == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long* d = a.dbg + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16;
        d[0] = dbg_t0;
        d[1] = dbg_t1;
        d[2] = dbg_t2;
        d[3] = __builtin_amdgcn_s_memtime();
        d[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));     // HW_REG_HW_ID
        d[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));    // HW_REG_XCC_ID
        d[6] = __builtin_amdgcn_s_memrealtime();
        d[7] = tile;
        d[11] = dbg_e1;
        d[12] = dbg_r0;
    }
}

template <int WM, int WN, int NT, int NTB>
int launch_cfg(const ConvArgs& a, int grid_y, hipStream_t stream) {
    using C = ConvCfg<WM, WN, NT, NTB>;
    auto kern = conv3x3_mfma_kernel<WM, WN, NT, NTB>;
    static PnpPerDevice once;
    const hipError_t attr_err = once.run([&](int, int&) {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   C::LDS_BYTES);
    });
    if (attr_err != hipSuccess) return (int)attr_err;
    const int tiles = ((a.W + C::TW - 1) / C::TW) * ((a.H + C::TH - 1) / C::TH);
    hipLaunchKernelGGL(kern, dim3(tiles, grid_y), dim3(256), C::LDS_BYTES, stream, a);
    return (int)hipGetLastError();
}

}  // namespace

int conv_pick_cfg(int H, int W) {
    // 8x16 tiles need >= 2 blocks per CU (512 tiles) to fill the chip; below that use 4x16 tiles.
    const long tiles_big = (long)((W + 15) / 16) * ((H + 7) / 8);
    return tiles_big >= 1024 ? CONV_CFG_BIG : CONV_CFG_SMALL;
}

int launch_conv3x3(const ConvArgs& a, int cfg, int grid_y, hipStream_t stream) {
    if (a.nsrc < 1 || a.nsrc > 4) return PNP_ERR_BAD_ARG;
    if (a.wpar && (a.nsrc != 1 || a.src_c[0] != 64 || !a.par)) return PNP_ERR_BAD_ARG;
    for (int s = 0; s < a.nsrc; ++s) {
        if (a.src_c[s] != 64 && a.src_c[s] != 4) return PNP_ERR_BAD_ARG;
        if (s > 0 && a.src_c[s] != 64) return PNP_ERR_BAD_ARG;   // the RGB frame may only be source 0
    }
    if (a.prec == 1 && conv_f16_eligible(a, cfg, grid_y)) return launch_conv3x3_f16(a, grid_y, stream);
    if (a.prec == 2 && conv_f16x3_eligible(a, cfg, grid_y)) return launch_conv3x3_f16x3(a, cfg, stream);
    if (a.src_f16 || a.out_f16) return PNP_ERR_UNSUPPORTED;     // fp16 maps exist only between two fp16 launches
    if (conv_wino_eligible(a, cfg, grid_y) || conv_wino_ms_eligible(a, cfg, grid_y)) return launch_conv3x3_wino(a, stream);
    if (conv_last_valu_eligible(a, cfg, grid_y)) return launch_conv_last_valu(a, stream);
    if (conv_persist_eligible(a, cfg, grid_y)) return launch_conv3x3_persist(a, stream);
    switch (cfg) {
        case CONV_CFG_BIG: return launch_cfg<4, 1, 2, 2>(a, grid_y, stream);
        case CONV_CFG_SMALL: return launch_cfg<2, 2, 1, 2>(a, grid_y, stream);
        case CONV_CFG_RGB: return launch_cfg<4, 1, 1, 1>(a, grid_y, stream);
    }
    return PNP_ERR_BAD_ARG;
}
