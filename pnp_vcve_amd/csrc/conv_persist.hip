// Persistent variant of the fused 3x3 conv (conv_mfma.hip) for its dominant case: ONE 64-channel source,
// 64 output channels, frames with >= 1024 tiles -- i.e. the 16 BAE-block convs per frame and conv_hr, ~88 % of
// the device time of a 720p clip.
//
// Why: the r01 timeline (tools/trace_conv.py) shows a block spending ~20 k cycles per 8x16 tile waiting for
// its halo tile (prologue) against ~41 k cycles of MFMA work, so the two blocks of a CU alternate instead of
// both feeding the matrix pipe.  Here a block walks a strip of tiles and requests the NEXT tile's halo into
// registers during the LAST weight chunk of the current tile; after the epilogue it only has to drop those
// registers into LDS.  Weight chunks keep streaming round-robin (chunk 0 of the next tile follows the last
// chunk of this one), epilogue operands of the next tile are requested right after this tile's stores.
//
// Same arithmetic, same K order, same LDS layouts as conv_mfma.hip: results are bit-identical to it.
#include "conv_mfma.h"
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace {

template <int N>
using T = std::integral_constant<int, N>;

constexpr int TH = 8, TW = 16, PW = 18, PIX = (TH + 2) * PW, PSTR = 17, NT = 2, NTB = 2;
constexpr int CH4 = PNP_CHUNK_Q * NTB * 64;
constexpr int LDS_BYTES = (PIX * PSTR + 2 * CH4) * 16;
constexpr int AIT = (PIX * 16 + 255) / 256;
constexpr int BPT = CH4 / 256;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <bool PAR>
__global__ __launch_bounds__(256, 2) void conv3x3_persist_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* sA = reinterpret_cast<f32x4*>(smem_raw);
    f32x4* sB = sA + PIX * PSTR;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave;
    // MFMA row m of the wave's 2 x 16 pixel slice: row 0 is pixels 0..15 in order, row 1 is ROTATED by two pixels
    // (m = 16 + i is pixel (i - 2) mod 16).  With the 272-B pixel stride the 18-pixel halo rows are 32 B off a multiple
    // of 256 B, so in a ds_read_b128 lane group (4 + 4 lanes of row 0, 8 of row 1) the unrotated rows shared 8 banks
    // (2-way conflicts on 18 % of the LDS cycles); rotated, the two rows tile all 64 banks.
    const int m = lane & 31, h = lane >> 5, my = m >> 4, mx = my ? ((m + 14) & 15) : (m & 15);
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TW - 1) / TW;
    const int ntiles = tiles_x * ((H + TH - 1) / TH);

    // ---- this block's strip: XCD x (blocks b with b % 8 == x) owns a contiguous band of tiles, the
    //      band is dealt round-robin to the XCD's resident blocks (halo rows / weights stay in its L2)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int bq = ntiles >> 3, br = ntiles & 7;
    const int xbeg = xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq;
    const int xend = xbeg + bq + (xcd < br ? 1 : 0);
    // (a per-XCD atomic ticket queue instead of these static strips was measured: blocks then take 12..16 tiles
    //  instead of 14..15 but the kernel time does not move -- the CU's tile rate, not the tail, sets it)
    int tile = xbeg + slot;
    if (tile >= xend) return;

    int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;

    // activation as ONE slope for negative values (a runtime switch per element compiles to ~10 scalar
    // branch instructions per accumulator register -- 28 k cycles of epilogue in the first timeline)
    const float neg_slope = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.1f);
    constexpr int CW = NT * 8, PPI = 64 / CW, EIT = 32 / PPI;
    const int n0 = lane & 31;

    // Per-tile index arithmetic reads the thread/lane id through `tq`/`lq`, which are laundered through an
    // empty asm once per tile: otherwise LICM hoists ~100 VGPRs of tile-invariant indices out of the strip
    // loop and the kernel spills.
    int tq = t, lq = lane;
    float bco[NT], gco[NT], pv[3] = {0.f, 0.f, 0.f};
    f32x4 res4[EIT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        bco[j] = a.bias ? a.bias[j * 32 + n0] : 0.f;
        gco[j] = a.gamma ? a.gamma[j * 32 + n0] : 1.f;
    }
    auto prefetch_tile_operands = [&](int y0, int x0) {
#pragma unroll
        for (int i = 0; i < EIT; ++i) {
            const int p = (lq / CW) + i * PPI;
            const int gy = y0 + 2 * wm + (p >> 4), gx = x0 + (p & 15);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (a.residual && gy < H && gx < W)
                v = *reinterpret_cast<const f32x4*>(a.residual + ((long)gy * W + gx) * 64 + (lq % CW) * 4);
            res4[i] = v;
        }
        if (PAR) {
            const int gy = y0 + 2 * wm + my, gx = x0 + mx;
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
                pv[jj] = (gy < H && gx < W) ? a.par[jj * a.par_plane + (long)gy * W + gx] : 0.f;
        }
    };

    f32x4 breg[BPT];
    auto load_b = [&](const f32x4* g) {
#pragma unroll
        for (int i = 0; i < BPT; ++i) breg[i] = g[tq + 256 * i];     // tq: see below (not hoistable)
    };
    auto store_b = [&](int buf) {
        f32x4* d = sB + buf * CH4;
#pragma unroll
        for (int i = 0; i < BPT; ++i) d[tq + 256 * i] = breg[i];
    };
    const int bofs = lane;

    f32x4 areg[AIT];
    const float* sp = a.src[0];
    // Per-tile index arithmetic reads the thread/lane id through `tq`/`lq`, which are laundered through an
    // empty asm once per tile: otherwise LICM hoists ~100 VGPRs of tile-invariant indices out of the strip
    // loop and the kernel spills.

    // branch-free (clamped address + select) so that the loads form ONE basic block and can be dealt into
    // the MFMA gaps of the chunk that issues them
    auto stage_load = [&](int y0, int x0) {
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int i = tq + 256 * k;
            const int pix = (i >> 4) < PIX ? (i >> 4) : PIX - 1, c16 = i & 15;
            const int ry = pix / PW, rx = pix - ry * PW;
            const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
            const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int cy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), cx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
            const unsigned off = ((unsigned)(cy * W + cx) * 64u + (unsigned)c16 * 4u) * 4u;   // < 4 GiB per map
            f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(sp) + off);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            areg[k] = inb ? v : z;
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int i = tq + 256 * k;
            if (i < PIX * 16) sA[(i >> 4) * PSTR + (i & 15)] = areg[k];
        }
    };

    const f32x4* a_lane = sA + ((2 * wm + my) * PW + mx) * PSTR + h;
    auto a_wide = [&](int tap, int q) -> f32x4 {
        const int dy = tap / 3, dx = tap - dy * 3;
        return a_lane[(dy * PW + dx) * PSTR + 2 * q];
    };

    const f32x4* wb = reinterpret_cast<const f32x4*>(a.wsrc[0]);
    const f32x4* wp = PAR ? reinterpret_cast<const f32x4*>(a.wpar) : nullptr;

    f32x16 acc[NT];
    int cbuf = 0;
    f32x4 av, bv[NT];
    int nty0 = 0, ntx0 = 0;          // next tile of the strip (valid when has_next)
    bool has_next = false;
    // output rows of the PREVIOUS tile, already transposed (+ residual), waiting to be stored from inside the
    // current tile's first chunk -- no VMEM instruction is issued in the serial hand-over between two tiles
    f32x4 orow[EIT];
    bool have_rows = false;
    int oy0 = 0, ox0 = 0;
    auto store_rows = [&]() {
#pragma unroll
        for (int i = 0; i < EIT; ++i) {
            const int p = (lq / CW) + i * PPI;
            const int gy = oy0 + 2 * wm + (p >> 4), gx = ox0 + (p & 15);
            if (gy < H && gx < W) *reinterpret_cast<f32x4*>(a.out + ((long)gy * W + gx) * 64 + (lq % CW) * 4) = orow[i];
        }
    };

    // KIND 0: 3x3 tap TAP; 2: 1x1 branch (centre tap, A scaled by ps).  NEXT_TAP < 0: nothing of sA is read
    // after this chunk's barrier.  ROLE bit 0: the tile's last chunk -> request the next tile's halo;
    // bit 1: first chunk -> store the previous tile's rows; bit 2: second chunk -> request residual / par.
    auto run_chunk = [&](auto kind_c, auto tap_c, auto ntap_c, auto pf_c, float ps, const f32x4* next) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr int TAP = decltype(tap_c)::value;
        constexpr int NEXT_TAP = decltype(ntap_c)::value;
        constexpr int ROLE = decltype(pf_c)::value;
        constexpr bool PF = (ROLE & 1) != 0;
        const f32x4* bb = sB + cbuf * CH4 + bofs;
        const f32x4* bn_base = sB + (cbuf ^ 1) * CH4 + bofs;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f32x4 an = av, bn[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bn[j] = bv[j];
            if (q == 7) __syncthreads();
            if (q == 0 && next) load_b(next);
            if (PF && q == 1) stage_load(nty0, ntx0);     // (the current tile again when the strip ends)
            if ((ROLE & 2) && q == 1 && have_rows) store_rows();
            if ((ROLE & 4) && q == 1) prefetch_tile_operands(ty0, tx0);
            if (q < 7) {
                an = a_wide(TAP, q + 1);
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
            if (q == 6 && next) store_b(cbuf ^ 1);
#pragma unroll
            for (int g = 0; g < 1 + NT; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (q == 0) {
#pragma unroll
                for (int g = 0; g < BPT; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
            if (PF && q == 1) {
#pragma unroll
                for (int g = 0; g < 5; ++g) {      // spread the 12 halo loads over the remaining MFMA gaps
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
                }
            }
            if (q == 6) {
#pragma unroll
                for (int g = 0; g < BPT; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            av = an;
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = bn[j];
        }
        cbuf ^= 1;
    };
    using K0 = T<0>;
    using K2 = T<2>;

    unsigned long long dbg_k = 0, dbg_e = 0, dbg_h = 0, dbg_t0 = 0, dbg_a = 0, dbg_p[6] = {0, 0, 0, 0, 0, 0};
    int dbg_n = 0;
    unsigned long long dbg_r0 = 0;      // the constant 100 MHz counter next to the shader clock: their ratio is the clock held
    if (a.dbg) {
        dbg_t0 = __builtin_amdgcn_s_memtime();
        dbg_r0 = __builtin_amdgcn_s_memrealtime();
    }
    // ---- prologue for the first tile of the strip
    __builtin_amdgcn_s_setprio(3);
    stage_load(ty0, tx0);
    load_b(wb);
    stage_store();
    store_b(0);
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);

    for (;;) {
        asm volatile("" : "+v"(tq), "+v"(lq));
        if (a.dbg) dbg_a = __builtin_amdgcn_s_memtime();
        const int ntile = tile + nslots;
        has_next = ntile < xend;
        nty0 = has_next ? (ntile / tiles_x) * TH : ty0;
        ntx0 = has_next ? (ntile % tiles_x) * TW : tx0;
        // chunk 0 of the NEXT tile follows this tile's last chunk in the weight stream
        const f32x4* wrap = has_next ? wb : nullptr;

#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        av = a_wide(0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = sB[cbuf * CH4 + bofs + j * 64];

        run_chunk(K0{}, T<0>{}, T<1>{}, T<2>{}, 1.f, wb + 1L * CH4);
        run_chunk(K0{}, T<1>{}, T<2>{}, T<4>{}, 1.f, wb + 2L * CH4);
        run_chunk(K0{}, T<2>{}, T<3>{}, T<0>{}, 1.f, wb + 3L * CH4);
        run_chunk(K0{}, T<3>{}, T<4>{}, T<0>{}, 1.f, wb + 4L * CH4);
        run_chunk(K0{}, T<4>{}, T<5>{}, T<0>{}, 1.f, wb + 5L * CH4);
        run_chunk(K0{}, T<5>{}, T<6>{}, T<0>{}, 1.f, wb + 6L * CH4);
        run_chunk(K0{}, T<6>{}, T<7>{}, T<0>{}, 1.f, wb + 7L * CH4);
        run_chunk(K0{}, T<7>{}, T<8>{}, T<0>{}, 1.f, wb + 8L * CH4);
        if constexpr (PAR) {
            // The 1x1 partition branches are a K extension whose A operand is x scaled by the branch's plane: a branch
            // whose plane is zero over the whole tile contributes exact zeros and is skipped (codec partition maps are
            // one-hot per >= 8x8 block, so an 8x16 tile needs 0..2 of the 3 branches; a P/B frame of random blocks 1.67).
            const int need = a.par_flags ? (a.par_flags[tile] & 7) : 7;           // block-uniform (scalar load)
            const int cnt = (need & 1) + ((need >> 1) & 1) + ((need >> 2) & 1);
            const int j0 = (need & 1) ? 0 : ((need & 2) ? 1 : 2);                 // first needed branch (branch 2 with an
            const int j1 = ((need & 3) == 3) ? 1 : 2;                             // all-zero plane if none); second; third = 2
            run_chunk(K0{}, T<8>{}, T<4>{}, T<0>{}, 1.f, wp + (long)j0 * CH4);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = (acc[j][r] + bco[j]) * gco[j];
            // the tile's last chunk is always a branch chunk (it carries the next halo request): a tile that needs no
            // branch still runs one, on a plane of zeros.  (Measured and rejected: a second, branch-free tile body for
            // frames without any record -- I frames -- selected by a per-frame flag: the merged kernel spills 31 instead
            // of 8 registers and the plain kernel grows from 205 to 231, which costs what the 32 dummy chunks saved.)
            int cur = j0;
            for (int i = 0; i + 1 < cnt; ++i) {
                const int nxt = i == 0 ? j1 : 2;
                run_chunk(K2{}, T<4>{}, T<4>{}, T<0>{}, cur == 0 ? pv[0] : (cur == 1 ? pv[1] : pv[2]), wp + (long)nxt * CH4);
                cur = nxt;
            }
            run_chunk(K2{}, T<4>{}, T<-1>{}, T<1>{}, cur == 0 ? pv[0] : (cur == 1 ? pv[1] : pv[2]), wrap);
        } else {
            run_chunk(K0{}, T<8>{}, T<-1>{}, T<1>{}, 1.f, wrap);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = (acc[j][r] + bco[j]) * gco[j];
        }

        unsigned long long dbg_b = 0, dbg_c = 0;
        if (a.dbg) {
            dbg_b = __builtin_amdgcn_s_memtime();
            dbg_k += dbg_b - dbg_a;
            ++dbg_n;
        }
        // ---- epilogue of this tile: transpose through the wave's slice of sA (free: every wave passed the
        //      last chunk's barrier after its final A read), whole 256-B pixel rows to HBM
        __builtin_amdgcn_s_setprio(3);
        float* sT = reinterpret_cast<float*>(sA) + wave * (32 * NT * 32);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[j][r];
                v = fmaxf(v, 0.f) + neg_slope * fminf(v, 0.f);     // branch-free none / relu / leaky-relu
                const int mm = (r & 3) + 8 * (r >> 2) + 4 * h;                  // MFMA row -> pixel of the slice
                const int pp = (r >> 3) ? 16 + ((mm + 14) & 15) : mm;
                sT[pp * (NT * 32) + j * 32 + n0] = v;
            }
        asm volatile("" ::: "memory");
        unsigned long long dbg_x = 0;
        if (a.dbg) {
            dbg_x = __builtin_amdgcn_s_memtime();
            dbg_p[0] += dbg_x - dbg_b;
        }
        // rows back into registers (+ residual); they are stored from inside the next tile's first chunk
        const f32x4* sT4 = reinterpret_cast<const f32x4*>(sT);
#pragma unroll
        for (int i = 0; i < EIT; ++i) orow[i] = sT4[((lq / CW) + i * PPI) * CW + (lq % CW)] + res4[i];
        have_rows = true;
        oy0 = ty0;
        ox0 = tx0;
        if (a.dbg) {
            dbg_c = __builtin_amdgcn_s_memtime();
            dbg_e += dbg_c - dbg_b;
        }
        if (!has_next) {
            store_rows();
            break;
        }
        // ---- hand over to the next tile: its halo is already in registers; no VMEM instruction here
        __syncthreads();                 // all transposes read back: sA may be overwritten
        unsigned long long dbg_y = 0;
        if (a.dbg) {
            dbg_y = __builtin_amdgcn_s_memtime();
            dbg_p[1] += dbg_y - dbg_c;
        }
        stage_store();
        if (a.dbg) {
            const unsigned long long z = __builtin_amdgcn_s_memtime();
            dbg_p[2] += z - dbg_y;
            dbg_y = z;
        }
        tile = ntile;
        ty0 = nty0;
        tx0 = ntx0;
        if (a.dbg) {
            const unsigned long long z = __builtin_amdgcn_s_memtime();
            dbg_p[3] += z - dbg_y;
            dbg_y = z;
        }
        __syncthreads();                 // halo visible
        __builtin_amdgcn_s_setprio(0);
        if (a.dbg) {
            const unsigned long long z = __builtin_amdgcn_s_memtime();
            dbg_p[4] += z - dbg_y;
            dbg_h += z - dbg_c;
        }
    }
    if (a.dbg && t == 0) {
        unsigned long long* d = a.dbg + (size_t)blockIdx.x * 16;
        d[0] = dbg_t0;
        d[1] = dbg_k;
        d[2] = dbg_e;
        d[3] = __builtin_amdgcn_s_memtime();
        d[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
        d[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
        d[6] = dbg_h;
        d[7] = dbg_n;
        for (int i = 0; i < 5; ++i) d[8 + i] = dbg_p[i];
        d[13] = dbg_r0;
        d[14] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace

namespace {
// blockIdx.y = frame: par advances by 3 planes, flags by gridDim.x
__global__ __launch_bounds__(128) void par_tile_flags_kernel(const float* __restrict__ par, long plane, int* __restrict__ flags,
                                                             int H, int W, int tiles_x) {
    par += (long)blockIdx.y * 3 * plane;
    flags += (long)blockIdx.y * gridDim.x;
    const int tile = blockIdx.x, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int gy = ty * TH + (threadIdx.x >> 4), gx = tx * TW + (threadIdx.x & 15);
    const bool in = gy < H && gx < W;
    int bits = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float v = in ? par[j * plane + (long)gy * W + gx] : 0.f;
        if (__syncthreads_or(v != 0.f)) bits |= 1 << j;
        // bit 3 + j: every value of plane j in the tile is 0 or exactly 1/255 -- what the reference's loader writes
        // (loading_ipb.py: one-hot uint8 planes / 255.).  The split-fp16 kernel then contracts the branch with weights scaled by
        // 1/255 at pack time and a MASKED A operand instead of re-splitting par_j(pixel) * x per fragment (conv_f16x3.hip).
        if (__syncthreads_and(v == 0.f || v == PNP_PAR_UNIT)) bits |= 8 << j;
    }
    // bit 6: each 8x8 half of the tile (a wave's quadrant in the Winograd kernels) is, ON ITS PIXELS INSIDE THE IMAGE, all zero or has
    // exactly ONE live plane that is constant there -- what a one-hot map on >= 8x8 codec blocks gives; conv_wino.hip then folds the plane
    // into its weights (launch_par_frame_any ANDs this over the frame: the fold-only kernel's gate).  A half cut by the frame's edge
    // (180x320: the last row of quadrants is 4 pixels high) counts with the pixels it has, one wholly outside counts as empty: the
    // Winograd kernels read the values of outside pixels from the nearest inside one (pv_request's clamp), so they decide alike
    {
        __shared__ float first[2][3];
        const int half = (threadIdx.x & 15) >> 3;
        float vv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            vv[j] = in ? par[j * plane + (long)gy * W + gx] : 0.f;
            if ((threadIdx.x & 7) == 0 && (threadIdx.x >> 4) == 0) first[half][j] = vv[j];       // (row 0 of a tile is inside the image)
        }
        __syncthreads();
        bool ok = true;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            int live = 0, varies = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (__syncthreads_or(half == h2 && vv[j] != 0.f)) ++live;
                if (__syncthreads_or(half == h2 && in && vv[j] != first[h2][j])) ++varies;
            }
            ok = ok && (live == 0 || (live == 1 && varies == 0));
        }
        if (ok) bits |= 64;
    }
    if (threadIdx.x == 0) flags[tile] = bits;
}
}  // namespace

namespace {
// any[f]: frame f's tile flags reduced (one block per frame)
__global__ __launch_bounds__(256) void par_frame_any_kernel(const int* __restrict__ flags, int* __restrict__ any, int tiles) {
    int bits = 0, notfold = 0;
    for (int i = threadIdx.x; i < tiles; i += 256) {
        const int f = flags[(long)blockIdx.x * tiles + i];
        bits |= f & 7;
        notfold |= !(f & 64);
    }
    const int b0 = __syncthreads_or(bits & 1), b1 = __syncthreads_or(bits & 2), b2 = __syncthreads_or(bits & 4);
    const int nf = __syncthreads_or(notfold);
    // bits 0-2: a plane is nonzero somewhere in the frame; bit 3: every 8x8 quadrant of the frame is all zero or carries one constant plane
    if (threadIdx.x == 0) any[blockIdx.x] = (b0 ? 1 : 0) | (b1 ? 2 : 0) | (b2 ? 4 : 0) | (nf ? 0 : 8);
}
}  // namespace

int launch_par_frame_any(const int* flags, int* any, int frames, int H, int W, hipStream_t stream) {
    const int tiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    hipLaunchKernelGGL(par_frame_any_kernel, dim3(frames), dim3(256), 0, stream, flags, any, tiles);
    return (int)hipGetLastError();
}

int launch_par_tile_flags(const float* par, long par_plane, int* flags, int frames, int H, int W, hipStream_t stream) {
    const int tiles_x = (W + TW - 1) / TW, tiles = tiles_x * ((H + TH - 1) / TH);
    hipLaunchKernelGGL(par_tile_flags_kernel, dim3(tiles, frames), dim3(128), 0, stream, par, par_plane, flags, H, W, tiles_x);
    return (int)hipGetLastError();
}

bool conv_persist_eligible(const ConvArgs& a, int cfg, int grid_y) {
    if (a.no_persist) return false;         // the caller asked for the tile-per-block kernel
    const long tiles = (long)((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    return cfg == CONV_CFG_BIG && grid_y == 1 && a.nsrc == 1 && a.src_c[0] == 64 && a.out_mode == 0 && tiles >= 1024;
}

int launch_conv3x3_persist(const ConvArgs& a, hipStream_t stream) {
    static PnpPerDevice once;
    int grid = 512;
    const hipError_t attr_err = once.run([](int dev, int& g) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_persist_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_persist_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        g = 2 * cus;                    // two resident blocks per CU (LDS-limited)
        g -= g % 8;
        return e;
    }, &grid);
    if (attr_err != hipSuccess) return (int)attr_err;
    if (a.wpar) hipLaunchKernelGGL(conv3x3_persist_kernel<true>, dim3(grid), dim3(256), LDS_BYTES, stream, a);
    else hipLaunchKernelGGL(conv3x3_persist_kernel<false>, dim3(grid), dim3(256), LDS_BYTES, stream, a);
    return (int)hipGetLastError();
}
