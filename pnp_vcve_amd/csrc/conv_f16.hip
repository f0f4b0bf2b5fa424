// fp16-operand variant of the fused 3x3 conv (BASELINE configs[4], "fp16 MFMA convs"): the same op as
// conv_mfma.hip -- same sources, same epilogue, fp32 accumulation -- with the A (activation) and B (weight)
// operands rounded to fp16 on their way into LDS and contracted by v_mfma_f32_32x32x16_f16 (16x the fp32
// matrix rate).  Opt-in (pnp_generator_set_precision); the fp32 kernels stay the default and the parity
// reference.  Feature maps stay fp32 in HBM, except the intermediate of a BAE block (SRC16 / OUT16 below).
//
// At this MFMA rate a 128-pixel tile needs ~1.2 us of matrix pipe but moves 64..110 KB through the CU's memory
// pipe, which takes roughly one 1-KiB wave instruction per 100 cycles and stalls the issuing wave meanwhile.
// So the kernel is organised around memory, not around the K loop:
//   * persistent: one 512-thread block per CU; ALL weights of the launch live in LDS for its lifetime
//     (9 taps x 64 x 64 fp16 = 72 KiB, + 24 KiB for the three 1x1 partition branches or 6 KiB for the RGB
//     source) -- no weight streaming, so the K loop has no barrier and its only traffic is LDS reads
//     (3 ds_read_b128 per 2 MFMAs, fetched 4 k-steps ahead);
//   * the block is TWO groups of 4 waves in anti-phase.  Each group walks its own strip of 8x16 tiles with its
//     own A tile; in every phase one group contracts a tile (matrix phase) while the other finishes the tile it
//     contracted one phase earlier (memory phase: transpose, residual, activation, 256-B pixel rows to HBM, next
//     halo fp32 -> fp16 -> LDS, request the halo after it).  Two block barriers per phase; the accumulators wait
//     in registers across them.  The two groups of a CU take adjacent tiles (shared halo columns hit L1/L2);
//   * halo tiles are requested ~1.3 phases before they are needed and the residual / partition values one
//     phase before, all through buffer descriptors: out-of-image lanes, ragged tiles and an absent residual are
//     out-of-range offsets (load 0 / store dropped), so a phase is branch-free and hipcc's vmcnt bookkeeping
//     stays exact with ~110 KB in flight per group.
// Measured and rejected: issuing the loads / stores one per k-step from inside the K loop (a wave waiting on the
// memory pipe cannot issue its MFMAs either: K loop 2500 -> 5500 cycles); 4 transposes of 8 pixels through a
// 2-KiB slice (8 LDS round trips per tile instead of 2 under the other group's K-loop traffic); the two groups
// free-running instead of phase-locked (a 4-wave barrier out of an LDS counter, transposes in two passes through
// the group's own A tile): same 122 us per block-conv launch -- what limits a CU here is not the phase coupling but
// how many global requests it keeps in flight (~10 GB/s per CU against a 24 GB/s HBM share on the front half).
//
// LDS map (bytes):  A tiles 2 x 28160 (10 rows x 2816: 18 px x 144 B, row stride = 0 mod 256 -- every
// ds_read_b128 of the K loop is conflict-free) | B 73728 | X 24576 (par branches, or RGB weights + the two RGB
// halos) | T 8192  = 162816 <= 163840.  Transposes (32 px x 64 ch fp32 = 8 KiB per wave) go through the
// group's own A tile, which is dead in its memory phase (waves 0..2), and T (wave 3).
//
// fp16 image of a weight chunk (made by f16_image_kernel from the fp32 "B image" of common.h): 1-KiB
// units [k-step s][n-tile][lane][8], lane (n, h) holding input channels 16 s + 8 h + 0..7 of output
// channel 32 nt + n -- the B fragment of v_mfma_f32_32x32x16_f16 -- so a fragment is ONE lane-linear
// ds_read_b128 and the global image is copied to LDS verbatim.
#include "conv_mfma.h"
#include "f16_util.h"
#include <mutex>
#include <type_traits>

// The fragment reads a k-step issues (for a k-step DEPTH ahead) are DEALT into its MFMA gaps instead of trailing the MFMAs as a
// burst: a wave issues in order, a 1-KiB ds_read_b128 holds its issue slot for ~16 cycles, and in the matrix phase of these kernels
// there is ONE wave per SIMD -- with the burst it ran 60 cycles per MFMA at 1.5 reads per MFMA (what r03 took for a structural
// limit), with the reads inside the gaps 33.5 (tools/ubench/ub_mfma_issue.hip, profiles/r04_ub_mfma_issue.txt).
#define F16_DEAL_READS(NT_)                                                         \
    if ((NT_) == 2) {                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
    } else {                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                          \
    }

namespace {

constexpr int B_BYTES = 9 * 4 * 2 * UNIT;         // 73728
constexpr int X_BYTES = 3 * 4 * 2 * UNIT;         // 24576
constexpr int T_BYTES = 4 * 8 * 64 * 4;           // 8192: 4 waves x 8 pixels x 64 channels fp32
constexpr int L_BYTES = 1536;                     // RGB halo of one group (180 x 8 B), inside X behind its 6 units
constexpr int OFF_A = 0, OFF_B = 2 * A_BYTES, OFF_X = OFF_B + B_BYTES, OFF_T = OFF_X + X_BYTES;
constexpr int OFF_L = OFF_X + 8 * UNIT;
constexpr int LDS_BYTES = OFF_T + T_BYTES;        // 162816
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(OFF_L + 2 * L_BYTES <= OFF_T, "RGB halos fit behind the RGB weights");
constexpr int AIT32 = (NPIX * 16 + 255) / 256;    // 16-byte halo loads per thread, fp32 source (4 channels each)
constexpr int AIT16 = (NPIX * 8 + 255) / 256;     //                               fp16 source (8 channels each)

struct F16Args {
    const void* src;        // NHWC64, fp32 or (SRC16) fp16
    const _Float16* w;      // 72 units per blockIdx.y
    const float* lr4;       // NHWC4 fp32 (RGB0) or nullptr
    const _Float16* wlr;    // 8 units (k = 4*tap + channel; the first 6 are read)
    const _Float16* wpar;   // 24 units or nullptr
    const float* par;
    long par_plane;
    const float* bias;
    const float* gamma;
    const float* residual;
    const float* lr;        // RGB head (out_mode 2 / 3): the low-quality frame, 3 NCHW planes
    long lr_plane;
    void* out;              // fp32, or (OUT == 1, out_mode 0, no residual) fp16 NHWC64
    void* out16;            // OUT == 2 (out_mode 0): an fp16 NHWC64 mirror of the fp32 output, written in the same pass
    const int* par_flags;   // PAR: per 8x16 tile, bit j set <=> partition plane j is nonzero somewhere in the tile (nullptr: all)
    long w_ystride;         // halfs
    int bias_ystride;
    int res_pre;            // residual is added BEFORE the activation (partial sum of a K-split launch chain)
    int H, W, act, out_mode, out_cstride;
    unsigned long long* dbg;
};

// fp16 mirror of a wave's 2 x 16 pixel output slice (OUT == 2).  In the fp32 epilogue lane (ec, ep) holds channels 4 ec .. 4 ec + 3
// of pixel i (row i >> 2, column ep + 4 (i & 3)) for i = 0..7: 8 bytes of fp16 per pixel.  Lanes ec and ec ^ 1 swap one value per
// pair of pixels (DPP quad_perm [1,0,3,2]) so that the even lane stores 16 B of pixel i and the odd lane 16 B of pixel i + 1:
// four 16-byte stores per lane instead of eight 8-byte ones.
__device__ __forceinline__ void store_mirror16(const h4 (&hv)[8], __amdgpu_buffer_rsrc_t r16, int row0, int tx0, int W, int ec, int ep) {
    typedef int i32x2 __attribute__((ext_vector_type(2)));
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
        const unsigned o = ((unsigned)(row0 + (ii >> 2)) * (unsigned)W + (unsigned)gx) * 128u + (unsigned)(ec >> 1) * 16u;
        buf_store4(r16, gx < W ? o : OOB, __builtin_bit_cast(f32x4, pk));
    }
}

// SRC16 / OUT == 1: the source / the output is an fp16 NHWC64 map.  Used for the intermediate of a BAE block
// (front half writes it, back half reads it): it is consumed only as an MFMA A operand, i.e. it would be
// rounded to fp16 by its reader anyway, so storing it rounded is bit-identical and halves its HBM traffic.
// OUT == 2: the fp32 output AND an fp16 mirror of it (F16Args::out16).  The running feature map x of a branch is
// needed in fp32 as the residual of the next block and in fp16 as the A operand of that block's front half (and of
// conv_hr / the neighbouring frames' input convs): the producer writes both, so no consumer reads 256 B per halo
// pixel only to round them to 128.
// RGB: the conv_last head -- ONE 32-channel N tile (3 valid), output = 3 NCHW planes + the low-quality frame
// (out_mode 2) or + its bilinear x4 upsampling (out_mode 3).
template <bool PAR, bool LR4, bool SRC16, int OUT, bool RGB>
__global__ __launch_bounds__(512) void conv3x3_f16_kernel(const F16Args a) {
    constexpr bool OUT16 = OUT == 1;
    static_assert(!(PAR && LR4), "the X region holds either the par branches or the RGB weights");
    static_assert(!RGB || (!PAR && !LR4 && OUT == 0), "the RGB head is a plain single-source conv");
    constexpr int NT = RGB ? 1 : 2;                // 32-channel N tiles per wave
    constexpr int AIT = SRC16 ? AIT16 : AIT32;
    constexpr int CPP = SRC16 ? 8 : 16;            // 16-byte slots per halo pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // 8 waves = two groups of 4.  A group is what a whole block was in the first version of this kernel: it walks
    // its own strip of tiles with its own A tile.  `t`, `wave` are relative to the group.
    const int grp = threadIdx.x >> 8, t = threadIdx.x & 255, lane = t & 63, wave = t >> 6;
    const int m = lane & 31, h = lane >> 5, my = m >> 4, mx = m & 15;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TW - 1) / TW;
    const int ntiles = tiles_x * ((H + TH - 1) / TH);
    const int yimg = blockIdx.y;

    // strips: XCD x (blocks with blockIdx.x % 8 == x) owns a contiguous band of tiles, dealt round-robin to the
    // 2 * nslots groups resident on it
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, step = 2 * (gridDim.x >> 3);
    const int bq = ntiles >> 3, br = ntiles & 7;
    const int xbeg = xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq;
    const int xend = xbeg + bq + (xcd < br ? 1 : 0);
    // The two groups of a block take ADJACENT tiles (their halos overlap: L1 / L2 hits; measured +12 % at 720p over
    // tiles half a round apart) -- unless the band has fewer tiles than groups: then group 1 of every block comes
    // after group 0 of all blocks, so that a small frame spreads over the CUs first.
    const bool dense = xend - xbeg >= step;
    const int first0 = xbeg + (dense ? 2 * slot : slot), first1 = first0 + (dense ? 1 : (step >> 1));
    const int cnt0 = first0 < xend ? (xend - 1 - first0) / step + 1 : 0;
    const int cnt1 = first1 < xend ? (xend - 1 - first1) / step + 1 : 0;
    if (cnt0 == 0) return;                                   // the whole block (cnt1 <= cnt0)
    const int cnt = grp ? cnt1 : cnt0;
    // group g contracts its j-th tile in phase g + 2 j and finishes it (epilogue, next halo) in phase g + 2 j + 1
    const int nphase = (2 * cnt1 + 1 > 2 * cnt0) ? 2 * cnt1 + 1 : 2 * cnt0;
    int tile = grp ? first1 : first0;

    unsigned long long dbg_t0 = 0, dbg_k = 0, dbg_e = 0, dbg_h = 0, dbg_w = 0, dbg_b = 0;
    int dbg_n = 0;
    if (a.dbg) dbg_t0 = __builtin_amdgcn_s_memtime();

    char* const sA = smem + OFF_A + grp * A_BYTES;
    char* const sL = smem + OFF_L + grp * L_BYTES;

    // ---- halo staging (fp32 or fp16 registers -> fp16 LDS)
    f32x4 areg[AIT], lreg = {0.f, 0.f, 0.f, 0.f};
    const unsigned map_bytes = (unsigned)H * (unsigned)W * 256u;       // < 4 GiB per feature map
    const __amdgpu_buffer_rsrc_t r_src = make_rsrc(a.src, SRC16 ? map_bytes / 2 : map_bytes);
    const __amdgpu_buffer_rsrc_t r_lr = make_rsrc(LR4 ? (const void*)a.lr4 : a.src, LR4 ? map_bytes / 16 : 0);
    const __amdgpu_buffer_rsrc_t r_res = RGB ? make_rsrc(a.lr, (unsigned)(3 * a.lr_plane * 4))
                                             : make_rsrc(a.residual ? (const void*)a.residual : a.src, a.residual ? map_bytes : 0);
    const __amdgpu_buffer_rsrc_t r_par = make_rsrc(PAR ? (const void*)a.par : a.src, PAR ? (unsigned)(3 * a.par_plane * 4) : 0);
    // Tile-invariant parts of the halo addressing, once per thread: 16-byte slot i = t + 256 k is channel group
    // i % CPP of halo pixel i / CPP.  Rows above / below the image need no test (the offset leaves the
    // descriptor's range by itself); columns left / right of it would wrap into the neighbouring row.
    unsigned hrel[AIT];
    int hpk[AIT];                                   // LDS byte offset | halo column << 16 (registers are scarce at 2 waves/SIMD)
#pragma unroll
    for (int k = 0; k < AIT; ++k) {
        const int i = t + 256 * k;
        const int pix = i / CPP, cs = i % CPP;
        const int ry = pix / PW, rx = pix - ry * PW;
        hrel[k] = (unsigned)(ry * W + rx) * (SRC16 ? 128u : 256u) + (unsigned)cs * 16u;
        const int col = pix < NPIX ? rx : 0x4000;   // slots past the tile: never in range
        hpk[k] = (ry * RSB + rx * PSB + cs * (SRC16 ? 16 : 8)) | (col << 16);
    }
    unsigned lrel = 0;
    int lrx = 0x4000;
    if (LR4) {
        const int ry = t / PW, rx = t - ry * PW;
        lrel = (unsigned)(ry * W + rx) * 16u;
        lrx = t < NPIX ? rx : 0x4000;
    }
    // `live` false (no such tile): every offset is out of range and nothing is fetched
    auto stage_load = [&](int y0, int x0, bool live) {
        const unsigned hbase = (unsigned)((y0 - 1) * W + (x0 - 1)) * (SRC16 ? 128u : 256u);
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            int pk = hpk[k];
            asm volatile("" : "+v"(pk));            // keeps LICM from unpacking it back into two loop-invariant registers
            const bool ok = live & ((unsigned)(x0 - 1 + (pk >> 16)) < (unsigned)W);
            areg[k] = buf_load4(r_src, ok ? hbase + hrel[k] : OOB);
        }
        if (LR4) {
            const bool ok = live & ((unsigned)(x0 - 1 + lrx) < (unsigned)W);
            lreg = buf_load4(r_lr, ok ? (unsigned)((y0 - 1) * W + (x0 - 1)) * 16u + lrel : OOB);
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            int pk = hpk[k];
            asm volatile("" : "+v"(pk));
            if ((pk >> 16) < PW) {
                if (SRC16) *reinterpret_cast<f32x4*>(sA + (pk & 0xffff)) = areg[k];      // 8 halfs, verbatim
                else *reinterpret_cast<h4*>(sA + (pk & 0xffff)) = to_h4(areg[k]);
            }
        }
        if (LR4 && t < NPIX) *reinterpret_cast<h4*>(sL + t * 8) = to_h4(lreg);
    };

    // ---- epilogue geometry: a lane owns 16 B (4 channels) of one pixel row per iteration
    constexpr int EIT = 8;
    const int ec = lane & 15, ep = lane >> 4;
    const int n0 = lane & 31;
    const float neg_slope = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.1f);
    const float k_pre = a.res_pre ? 1.f : 0.f, k_post = 1.f - k_pre;

    // PAR: which 1x1 branches the tile whose operands were prefetched last needs.  Fetched as a plain lane value with the
    // other tile operands (no wait here) and made wave-uniform only at the start of the matrix phase that consumes it.
    int pfl_v = 0;
    const __amdgpu_buffer_rsrc_t r_flags = make_rsrc((PAR && a.par_flags) ? (const void*)a.par_flags : a.src,
                                                     (PAR && a.par_flags) ? (unsigned)ntiles * 4u : 0);
    float bco[NT], gco[NT], pv[3] = {0.f, 0.f, 0.f};
    f32x4 res4[EIT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        bco[j] = a.bias ? a.bias[yimg * a.bias_ystride + j * 32 + n0] : 0.f;
        gco[j] = a.gamma ? a.gamma[j * 32 + n0] : 1.f;
    }
    // rows of the wave's 2 x 16 pixel slice: iteration i is pixel row i >> 2, column (lane >> 4) + 4 (i & 3)
    const unsigned row_bytes = (unsigned)W * 256u;
    // RGB head: value slot (pass, lane) is channel 2 pass + (lane >> 5), pixel lane & 31 of the wave's slice; its base
    // is the low-quality pixel (mode 2) or 4 bilinear taps of the quarter-size frame (mode 3,
    // F.interpolate(scale_factor=4, 'bilinear', align_corners=False), iconvsr_ipb_par.py:41,140)
    float lrv[2][4], lrw[2];
    auto prefetch_rgb_operands = [&](int y0, int x0, bool live) {
        const int gy = y0 + 2 * wave + (m >> 4), gx = x0 + (m & 15);
        const bool inb = live & (gy < H) & (gx < W);
        if (a.out_mode == 2) {
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int c = 2 * pass + h;
                lrv[pass][0] = buf_load1(r_res, (inb & (c < 3)) ? (unsigned)(c * a.lr_plane + (long)gy * W + gx) * 4u : OOB);
            }
        } else {
            const int lh = H >> 2, lw = W >> 2;
            float sy = (gy + 0.5f) * 0.25f - 0.5f, sx = (gx + 0.5f) * 0.25f - 0.5f;
            sy = sy < 0.f ? 0.f : sy;
            sx = sx < 0.f ? 0.f : sx;
            const int y0i = (int)sy, x0i = (int)sx;
            const int y1i = y0i + (y0i < lh - 1 ? 1 : 0), x1i = x0i + (x0i < lw - 1 ? 1 : 0);
            lrw[0] = sy - y0i;
            lrw[1] = sx - x0i;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int c = 2 * pass + h;
                const bool ok = inb & (c < 3);
                const long pb = c * a.lr_plane;
                lrv[pass][0] = buf_load1(r_res, ok ? (unsigned)(pb + (long)y0i * lw + x0i) * 4u : OOB);
                lrv[pass][1] = buf_load1(r_res, ok ? (unsigned)(pb + (long)y0i * lw + x1i) * 4u : OOB);
                lrv[pass][2] = buf_load1(r_res, ok ? (unsigned)(pb + (long)y1i * lw + x0i) * 4u : OOB);
                lrv[pass][3] = buf_load1(r_res, ok ? (unsigned)(pb + (long)y1i * lw + x1i) * 4u : OOB);
            }
        }
    };
    auto prefetch_tile_operands = [&](int y0, int x0, bool live) {
        if (RGB) {
            prefetch_rgb_operands(y0, x0, live);
            return;
        }
        const unsigned rbase = ((unsigned)((y0 + 2 * wave) * W + x0 + ep) * 64u + (unsigned)ec * 4u) * 4u;
#pragma unroll
        for (int i = 0; i < (OUT16 ? 0 : EIT); ++i) {
            const bool ok = live & (x0 + ep + 4 * (i & 3) < W);
            res4[i] = buf_load4(r_res, ok ? rbase + (unsigned)(i >> 2) * row_bytes + (unsigned)(i & 3) * 1024u : OOB);
        }
        if (PAR) {
            const int gy = y0 + 2 * wave + my, gx = x0 + mx;
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
                pv[jj] = buf_load1(r_par, (live & (gy < H) & (gx < W)) ? (unsigned)(jj * a.par_plane + (long)gy * W + gx) * 4u : OOB);
            // a plane that is zero over the whole tile contributes exact zeros: its 4 k-steps are skipped (value-identical)
            pfl_v = __builtin_bit_cast(int, buf_load1(r_flags, live ? (unsigned)((y0 / TH) * tiles_x + x0 / TW) * 4u : OOB));
        }
    };
    // output addressing of the three row-wise modes as one affine form (uniform scalars, no per-store switch)
    //   byte offset of output pixel (gy, gx) = gy * o_sy + gx * o_sx + o_c0  (+ 16 B per float4 of the row)
    unsigned o_sy, o_sx, o_c0;
    __amdgpu_buffer_rsrc_t r_out;
    {
        const int o_mul = a.out_mode == 1 ? 2 : 1;
        const unsigned o_pix = OUT16 ? 128u : (a.out_mode == 4 ? (unsigned)a.out_cstride : 64u) * 4u;   // bytes per output pixel
        const unsigned o_row = (unsigned)(o_mul * W) * o_pix;
        o_sy = (unsigned)__builtin_amdgcn_readfirstlane((int)(o_mul * o_row));
        o_sx = (unsigned)__builtin_amdgcn_readfirstlane((int)(o_mul * o_pix));
        o_c0 = (unsigned)__builtin_amdgcn_readfirstlane(
            (int)(a.out_mode == 1 ? (yimg >> 1) * o_row + (yimg & 1) * o_pix : (a.out_mode == 4 ? 256u * (unsigned)yimg : 0u)));
        r_out = RGB ? make_rsrc(a.out, (unsigned)H * (unsigned)W * 12u) : make_rsrc(a.out, (unsigned)(o_mul * H) * o_row);
    }
    const __amdgpu_buffer_rsrc_t r_out16 = make_rsrc(OUT == 2 ? a.out16 : a.out, OUT == 2 ? map_bytes / 2 : 0);

    int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    int done = 0;                       // tiles this group has finished
    bool acc_live = false;              // a contracted tile is waiting for its epilogue

    // ---- prologue: first halo + operands requested, then the weights (global fp16 image -> LDS verbatim, all 512
    //      threads), the halo converted into the group's A tile, the second halo requested
    stage_load(ty0, tx0, cnt > 0);
    prefetch_tile_operands(ty0, tx0, cnt > 0);
    {
        const int tt = threadIdx.x;
        const f32x4* g = reinterpret_cast<const f32x4*>(a.w + (long)yimg * a.w_ystride);
        f32x4* d = reinterpret_cast<f32x4*>(smem + OFF_B);
#pragma unroll
        for (int i = 0; i < (B_BYTES / 2 * NT / 16 + 511) / 512; ++i)
            if (tt + 512 * i < B_BYTES / 2 * NT / 16) d[tt + 512 * i] = g[tt + 512 * i];
        if (PAR) {
            const f32x4* gp = reinterpret_cast<const f32x4*>(a.wpar);
            f32x4* dp = reinterpret_cast<f32x4*>(smem + OFF_X);
#pragma unroll
            for (int i = 0; i < X_BYTES / 16 / 512; ++i) dp[tt + 512 * i] = gp[tt + 512 * i];
        }
        if (LR4) {
            const f32x4* gp = reinterpret_cast<const f32x4*>(a.wlr);
            f32x4* dp = reinterpret_cast<f32x4*>(smem + OFF_X);
            if (tt < 6 * UNIT / 16) dp[tt] = gp[tt];
        }
    }
    stage_store();
    {
        const int nt = tile + step;
        stage_load((nt / tiles_x) * TH, (nt % tiles_x) * TW, cnt > 1);
    }
    lds_barrier();

    const char* a_lane = sA + (2 * wave + my) * RSB + mx * PSB + 16 * h;
    const char* b_lane = smem + OFF_B + lane * 16;
    const char* x_lane = smem + OFF_X + lane * 16;
    // RGB source: k = 16 s + 8 h + j  ->  tap 4 s + 2 h + (j >> 2), channel j & 3; taps beyond 8 carry zero
    // weights and re-read tap 8 (finite values)
    const int l_base = ((2 * wave + my) * PW + mx) * 8;
    auto l_off = [&](int sk, int u) -> int {      // compile-time sk, u: two constants selected by the lane's k-half
        auto tap_off = [](int tap) {
            tap = tap > 8 ? 8 : tap;
            return ((tap / 3) * PW + tap % 3) * 8;
        };
        return l_base + (h ? tap_off(4 * sk + 2 + u) : tap_off(4 * sk + u));
    };
    // Transpose slices (32 px x 64 ch fp32 = 8 KiB per wave): in its memory phase a group's A tile is dead until the
    // next halo is written, so waves 0..2 use it and wave 3 the 8 KiB T region (shared by the groups: only one is in
    // its memory phase at a time).  A block barrier in the middle of every phase separates the transposes from the
    // halo write; the group in its matrix phase passes it between two k-steps.
    static_assert(3 * 8192 <= A_BYTES && T_BYTES >= 8192, "transpose slices");
    float* sT = reinterpret_cast<float*>(wave < 3 ? sA + wave * 8192 : smem + OFF_T);
    f32x16 acc[NT];

    for (int phase = 0; phase < nphase; ++phase) {
        if ((phase & 1) == grp) {
            if (done < cnt) {
                // =================== matrix phase: contract the tile in the group's A tile.  No barrier, no global
                // traffic.  Fragments are fetched DEPTH k-steps ahead of the MFMAs that use them (the compiler's own
                // schedule keeps ~1 step in flight and stalls on LDS latency); sched_barrier pins one fetch group
                // per MFMA pair.
                unsigned long long dbg_a = 0;
                if (a.dbg) dbg_a = __builtin_amdgcn_s_memtime();
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
                constexpr int NS = 36 + (LR4 ? 3 : 0);   // static k-steps; PAR adds 4 per partition branch the tile needs
                constexpr int DEPTH = LR4 ? 3 : 4;       // 3 * DEPTH <= 15 (lgkmcnt); the RGB variant is out of registers at 4
                static_assert(!PAR || (DEPTH == 4 && NS % DEPTH == 0), "branch step q lives in fetch slot q");
                // where the phase's middle barrier is passed: the partner group reaches it after its epilogue, which is the longer
                // part of its memory phase (r03 timelines, tools/bench_f16_block.py) -- the two halves of both phases are matched
                constexpr int MID = PAR ? 25 : (OUT == 1 ? 22 : 19);
                // wave-uniform list of the branches to run (exact zeros otherwise: value-identical)
                const int pflags = (PAR && a.par_flags) ? (__builtin_amdgcn_readfirstlane(pfl_v) & 7) : 7;
                h8 fa[DEPTH], fb0[DEPTH], fb1[DEPTH];
                auto fetch = [&](int k) {       // k is a compile-time constant after unrolling
                    const int sl = k % DEPTH;
                    if (k < 36) {
                        const int tap = k >> 2, sk = k & 3, dy = tap / 3, dx = tap - dy * 3;
                        fa[sl] = *reinterpret_cast<const h8*>(a_lane + dy * RSB + dx * PSB + 32 * sk);
                        fb0[sl] = *reinterpret_cast<const h8*>(b_lane + (k * NT + 0) * UNIT);
                        if (NT == 2) fb1[sl] = *reinterpret_cast<const h8*>(b_lane + (k * NT + 1) * UNIT);
                    } else {
                        const int sk = k - 36;
                        const h4 lo = *reinterpret_cast<const h4*>(sL + l_off(sk, 0));
                        const h4 hi = *reinterpret_cast<const h4*>(sL + l_off(sk, 1));
                        fa[sl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        fb0[sl] = *reinterpret_cast<const h8*>(x_lane + (sk * 2 + 0) * UNIT);
                        fb1[sl] = *reinterpret_cast<const h8*>(x_lane + (sk * 2 + 1) * UNIT);
                    }
                };
                // k-step q (compile time) of partition branch br (wave-uniform run-time value) -> fetch slot q
                auto fetch_par = [&](int br, int q) {
                    const char* xb = x_lane + br * (8 * UNIT);
                    fa[q] = *reinterpret_cast<const h8*>(a_lane + RSB + PSB + 32 * q);
                    fb0[q] = *reinterpret_cast<const h8*>(xb + (q * 2 + 0) * UNIT);
                    fb1[q] = *reinterpret_cast<const h8*>(xb + (q * 2 + 1) * UNIT);
                };
                auto bias_gamma = [&]() {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[j][r] = (acc[j][r] + bco[j]) * gco[j];
                };
                const int br0 = pflags ? __builtin_ctz(pflags) : -1;       // first branch to run
#pragma unroll
                for (int k = 0; k < DEPTH; ++k) fetch(k);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int sl = k % DEPTH;
                    const h8 av = fa[sl];
                    const h8 b0 = fb0[sl], b1 = fb1[sl];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b0, acc[0], 0, 0, 0);
                    if (NT == 2) acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b1, acc[NT - 1], 0, 0, 0);
                    if (k + DEPTH < NS) fetch(k + DEPTH);
                    else if (PAR && br0 >= 0) fetch_par(br0, k + DEPTH - NS);
                    F16_DEAL_READS(NT)
                    if (k == MID) __builtin_amdgcn_s_barrier();         // the phase's middle barrier (no LDS hand-off here)
                    __builtin_amdgcn_sched_barrier(0);
                }
                bias_gamma();                  // (conv + bias) * gamma BEFORE the 1x1 partition branches
                if (PAR) {
#pragma unroll
                    for (int br = 0; br < 3; ++br) {
                        if (!((pflags >> br) & 1)) continue;
                        const int rest = pflags >> (br + 1);
                        const int nxt = rest ? br + 1 + __builtin_ctz(rest) : -1;      // the branch after this one, if any
                        const _Float16 pj = (_Float16)pv[br];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const h8 av = fa[q] * pj;
                            const h8 b0 = fb0[q], b1 = fb1[q];
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b0, acc[0], 0, 0, 0);
                            acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b1, acc[NT - 1], 0, 0, 0);
                            if (nxt >= 0) fetch_par(nxt, q);
                            F16_DEAL_READS(2)
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                acc_live = true;
                if (a.dbg) dbg_k += __builtin_amdgcn_s_memtime() - dbg_a;
            } else {
                __builtin_amdgcn_s_barrier();
            }
        } else if (acc_live) {
            // =================== memory phase (the other group is in its matrix phase): finish the tile contracted
            // one phase ago, bring the next halo into the A tile, request the one after it.
            unsigned long long dbg_a = 0, dbg_c = 0;
            if (a.dbg) dbg_a = __builtin_amdgcn_s_memtime();
            // Accumulator register r of lane (n0, h) is pixel (r&3) + 8 (r>>2) + 4 h of the wave's 32-pixel M tile,
            // channel 32 j + n0.  Pass q transposes pixels 8 q .. 8 q + 7 through the wave's 2-KiB slice.
            const f32x4* sT4 = reinterpret_cast<const f32x4*>(sT);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sT[((r & 3) + 8 * (r >> 2) + 4 * h) * (32 * NT) + j * 32 + n0] = acc[j][r];
            asm volatile("" ::: "memory");
            if (RGB) {
                // [pixel][32 channels] slice: two passes of 64 (channel, pixel) values -> whole 64-B row segments of the
                // output planes instead of 16 six-lane stores
                float val[2];
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) val[pass] = sT[m * 32 + ((2 * pass + h) & 3)];
                asm volatile("" ::: "memory");
                const int gy = ty0 + 2 * wave + (m >> 4), gx = tx0 + (m & 15);
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int c = 2 * pass + h;
                    float v = val[pass];
                    v = fmaxf(v, 0.f) + neg_slope * fminf(v, 0.f);
                    float base;
                    if (a.out_mode == 2) {
                        base = lrv[pass][0];
                    } else {
                        const float ly = lrw[0], lx = lrw[1];
                        base = (1.f - ly) * ((1.f - lx) * lrv[pass][0] + lx * lrv[pass][1]) +
                               ly * ((1.f - lx) * lrv[pass][2] + lx * lrv[pass][3]);
                    }
                    const bool ok = (c < 3) & (gy < H) & (gx < W);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + base), r_out,
                                                          (int)(ok ? (unsigned)((long)c * H * W + (long)gy * W + gx) * 4u : OOB), 0, 0);
                }
            } else if (OUT16) {
                // a lane owns 8 channels (16 B of fp16) of one pixel: 4 instead of 8 stores per wave
                const int ec8 = lane & 7, ep8 = lane >> 3;
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
                    const int gx = tx0 + ep8 + 8 * (i & 1);
                    const unsigned o = (unsigned)(ty0 + 2 * wave + (i >> 1)) * o_sy + (unsigned)gx * o_sx + o_c0 + (unsigned)ec8 * 16u;
                    buf_store4(r_out, gx < W ? o : OOB, __builtin_bit_cast(f32x4, pk));
                }
            } else {
                f32x4 rows[EIT];
                h4 hv[EIT];
#pragma unroll
                for (int i = 0; i < EIT; ++i) rows[i] = sT4[(ep + 4 * i) * 16 + ec];      // all reads in flight together
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < EIT; ++i) {        // pixel row i >> 2, column ep + 4 (i & 3) of the wave's slice
                    f32x4 v = rows[i] + k_pre * res4[i];
                    v = __builtin_elementwise_max(v, (f32x4)(0.f)) + neg_slope * __builtin_elementwise_min(v, (f32x4)(0.f));
                    v += k_post * res4[i];
                    const int gx = tx0 + ep + 4 * (i & 3);
                    const unsigned o = (unsigned)(ty0 + 2 * wave + (i >> 2)) * o_sy + (unsigned)gx * o_sx + o_c0 + (unsigned)ec * 16u;
                    buf_store4(r_out, gx < W ? o : OOB, v);
                    if (OUT == 2) hv[i] = to_h4(v);
                }
                if (OUT == 2) store_mirror16(hv, r_out16, ty0 + 2 * wave, tx0, W, ec, ep);
            }
            acc_live = false;
            ++done;
            if (a.dbg) {
                dbg_c = __builtin_amdgcn_s_memtime();
                dbg_e += dbg_c - dbg_a;
                ++dbg_n;
            }
            const bool more = done < cnt;
            if (more) {
                tile += step;
                ty0 = (tile / tiles_x) * TH;
                tx0 = (tile % tiles_x) * TW;
                prefetch_tile_operands(ty0, tx0, true);
            }
            lds_barrier();      // middle barrier: every wave of the group has read its transposed rows back
            if (more) {
                unsigned long long dbg_d = 0;
                if (a.dbg) dbg_d = __builtin_amdgcn_s_memtime();
                stage_store();                   // waits for the halo requested one memory phase ago
                if (a.dbg) dbg_w += __builtin_amdgcn_s_memtime() - dbg_d;
                const int nt = tile + step;
                stage_load((nt / tiles_x) * TH, (nt % tiles_x) * TW, done + 1 < cnt);
            }
            if (a.dbg) dbg_h += __builtin_amdgcn_s_memtime() - dbg_c;
        } else {
            __builtin_amdgcn_s_barrier();
        }
        unsigned long long dbg_x = 0;
        if (a.dbg) dbg_x = __builtin_amdgcn_s_memtime();
        lds_barrier();
        if (a.dbg) dbg_b += __builtin_amdgcn_s_memtime() - dbg_x;
    }
    if (a.dbg && t == 0) {
        unsigned long long* d = a.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + grp) * 16;
        d[0] = dbg_t0;
        d[1] = dbg_k;
        d[2] = dbg_e;
        d[3] = __builtin_amdgcn_s_memtime();
        d[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
        d[5] = nphase;
        d[6] = dbg_h;
        d[7] = dbg_n;
        d[8] = dbg_w;
        d[9] = dbg_b;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Small frames (fewer tiles than ~4 per CU: 180x320, 128x128).  The persistent kernel above pays its 72..96 KiB weight
// prologue and its anti-phase pipeline for one or two tiles per block (12-14 us per launch at 180x320, 3 phases for two
// tiles, one launch monopolising every CU's LDS).  Here a block is ONE 8x16 tile: 4 waves, the A tile + a 3-slot ring of
// 8-KiB weight chunks (one 3x3 tap or one 1x1 branch = 4 k-steps x 2 N tiles) = 52 KiB of LDS -> three blocks per CU, the
// whole frame resident at once, every block's latencies (halo from L2, chunk hand-over barriers, epilogue) hidden by its two
// neighbours.  Chunk c+2 is requested while chunk c is contracted; one barrier per chunk.  Same k order, same fp32
// accumulation and same epilogue arithmetic as the persistent kernel: bit-identical results (tested).
// Covers the single-source 64 -> 64 launches (both halves of a BAE block, conv_hr): 33 of the ~38 launches of a frame.
// ---------------------------------------------------------------------------------------------------------------
constexpr int S_RING = 3;
constexpr int S_PAIR_TILES = 512;     // from this many tiles on, a block takes two adjacent tiles (measured: r02)
constexpr int S_CHUNK = 8 * UNIT;                          // 8192 B: 4 k-steps x 2 N tiles
constexpr int s_lds_bytes(int g) { return g * A_BYTES + S_RING * S_CHUNK; }      // 52,736 (G = 1) / 80,896 (G = 2)
static_assert(3 * s_lds_bytes(1) <= 160 * 1024 && 2 * s_lds_bytes(2) <= 160 * 1024, "three / two blocks per CU");
static_assert(s_lds_bytes(1) >= 4 * 8192 && s_lds_bytes(2) >= 8 * 8192, "the epilogue transposes 8 KiB per wave through the dead LDS");

// G = groups of 4 waves per block, each group one tile (adjacent tiles), all sharing the block's weight ring: G = 2 halves the
// weight bytes a tile pulls from L2 (two blocks per CU); G = 1 keeps three blocks per CU for the smallest frames.
template <bool PAR, bool SRC16, int OUT, int G>
__global__ __launch_bounds__(256 * G, G == 1 ? 3 : 4) void conv3x3_f16_small_kernel(const F16Args a) {
    constexpr bool OUT16 = OUT == 1;
    constexpr int AIT = SRC16 ? AIT16 : AIT32;
    constexpr int CPP = SRC16 ? 8 : 16;
    constexpr int NC = 9 + (PAR ? 3 : 0);                  // weight chunks
    constexpr int NT_ = 256 * G;                           // threads
    constexpr int WPT = 512 / NT_;                         // float4 of a weight chunk per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tt = threadIdx.x, t = tt & 255, grp = tt >> 8, lane = t & 63, wave = t >> 6;
    const int m = lane & 31, h = lane >> 5, my = m >> 4, mx = m & 15;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TW - 1) / TW;
    const int ntiles = tiles_x * ((H + TH - 1) / TH);
    int tile;
    {   // XCD-aware remap: each XCD walks a contiguous band of tiles (halo rows meet in its L2)
        const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7;
        const int q = nwg >> 3, r = nwg & 7;
        tile = ((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3)) * G + grp;
    }
    const bool live = tile < ntiles;                       // an odd tile count leaves the last block's second group idle
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    char* const sA = smem + grp * A_BYTES;
    char* const sR = smem + G * A_BYTES;
    unsigned long long d_t0 = 0, d_t1 = 0, d_t2 = 0, d_t3 = 0;
    if (a.dbg) d_t0 = __builtin_amdgcn_s_memtime();

    const unsigned map_bytes = (unsigned)H * (unsigned)W * 256u;
    const __amdgpu_buffer_rsrc_t r_src = make_rsrc(a.src, SRC16 ? map_bytes / 2 : map_bytes);
    const __amdgpu_buffer_rsrc_t r_res = make_rsrc(a.residual ? (const void*)a.residual : a.src, a.residual ? map_bytes : 0);
    const __amdgpu_buffer_rsrc_t r_par = make_rsrc(PAR ? (const void*)a.par : a.src, PAR ? (unsigned)(3 * a.par_plane * 4) : 0);
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(a.out, (unsigned)H * (unsigned)W * (OUT16 ? 128u : 256u));
    const __amdgpu_buffer_rsrc_t r_out16 = make_rsrc(OUT == 2 ? a.out16 : a.out, OUT == 2 ? map_bytes / 2 : 0);
    // PAR: which 1x1 branches this tile needs (fetched now, made wave-uniform when the first branch chunk is scheduled)
    const __amdgpu_buffer_rsrc_t r_flags = make_rsrc((PAR && a.par_flags) ? (const void*)a.par_flags : a.src,
                                                     (PAR && a.par_flags) ? (unsigned)ntiles * 4u : 0);
    // G = 2: the two groups of a block share the weight ring and its barriers, so they run the same chunk list -- the union of
    // what their two tiles need (a branch the other tile alone needs adds exact zeros here: its plane is zero on this tile)
    int pfl_v = 0;
    if (PAR) {
        const int t0 = tile - grp;
#pragma unroll
        for (int gq = 0; gq < G; ++gq)
            pfl_v |= __builtin_bit_cast(int, buf_load1(r_flags, (t0 + gq < ntiles) ? (unsigned)(t0 + gq) * 4u : OOB));
    }

    // ---- requests: halo, the first weight chunks, residual rows / partition values
    f32x4 areg[AIT];
    const unsigned hbase = (unsigned)((ty0 - 1) * W + (tx0 - 1)) * (SRC16 ? 128u : 256u);
#pragma unroll
    for (int k = 0; k < AIT; ++k) {
        const int i = t + 256 * k;
        const int pix = i / CPP, cs = i % CPP;
        const int ry = pix / PW, rx = pix - ry * PW;
        const bool ok = live & (pix < NPIX) & ((unsigned)(tx0 - 1 + rx) < (unsigned)W);  // rows outside the image leave the descriptor by themselves
        areg[k] = buf_load4(r_src, ok ? hbase + (unsigned)(ry * W + rx) * (SRC16 ? 128u : 256u) + (unsigned)cs * 16u : OOB);
    }
    const f32x4* wg = reinterpret_cast<const f32x4*>(a.w);
    const f32x4* wgp = reinterpret_cast<const f32x4*>(a.wpar);
    auto chunk_ptr = [&](int c) -> const f32x4* {          // 512 float4 per chunk
        return (PAR && c >= 9) ? wgp + (c - 9) * 512 : wg + c * 512;
    };
    // weight chunks: chunk c is requested at the start of chunk c - 3's contraction, written to its ring slot at the end of
    // chunk c - 2's and used after the barrier that ends chunk c - 1 -- two chunk durations for the load to land
    f32x4 wreg[2][WPT];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const f32x4* g = chunk_ptr(c);
        f32x4 v[WPT];
#pragma unroll
        for (int i = 0; i < WPT; ++i) v[i] = g[tt + NT_ * i];
#pragma unroll
        for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(sR + c * S_CHUNK + (tt + NT_ * i) * 16) = v[i];
    }
    {
        const f32x4* g = chunk_ptr(2);
#pragma unroll
        for (int i = 0; i < WPT; ++i) wreg[0][i] = g[tt + NT_ * i];
    }
    constexpr int EIT = 8;
    const int ec = lane & 15, ep = lane >> 4, n0 = lane & 31;
    const float neg_slope = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.1f);
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
        for (int i = 0; i < (OUT16 ? 0 : EIT); ++i) {
            const bool ok = live & (tx0 + ep + 4 * (i & 3) < W);
            res4[i] = buf_load4(r_res, ok ? rbase + (unsigned)(i >> 2) * row_bytes + (unsigned)(i & 3) * 1024u : OOB);
        }
        if (PAR) {
            const int gy = ty0 + 2 * wave + my, gx = tx0 + mx;
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
                pv[jj] = buf_load1(r_par, (live & (gy < H) & (gx < W)) ? (unsigned)(jj * a.par_plane + (long)gy * W + gx) * 4u : OOB);
        }
    }
    // ---- halo -> fp16 A tile
#pragma unroll
    for (int k = 0; k < AIT; ++k) {
        const int i = t + 256 * k;
        const int pix = i / CPP, cs = i % CPP;
        const int ry = pix / PW, rx = pix - ry * PW;
        if (pix < NPIX) {
            char* d = sA + ry * RSB + rx * PSB + cs * (SRC16 ? 16 : 8);
            if (SRC16) *reinterpret_cast<f32x4*>(d) = areg[k];
            else *reinterpret_cast<h4*>(d) = to_h4(areg[k]);
        }
    }
    lds_barrier();
    if (a.dbg) d_t1 = __builtin_amdgcn_s_memtime();

    // ---- K loop: chunk c from ring slot c % 3
    const char* a_lane = sA + (2 * wave + my) * RSB + mx * PSB + 16 * h;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // PAR: chunks 9.. are the partition branches the tile needs, in plane order (a plane that is zero over the whole tile adds
    // exact zeros: skipped, value-identical).  ncr = chunks to run; bsel(j) = plane of the j-th branch chunk.
    int ncr = NC, bs0 = 0, bs1 = 1, bs2 = 2;
    auto bsel = [&](int j) { return j == 0 ? bs0 : (j == 1 ? bs1 : bs2); };
    auto bias_gamma = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = (acc[j][r] + bco[j]) * gco[j];
    };
    // One continuous stream of k-steps (4 per chunk; chunk c lives in ring slot c % 3).  Fragments are fetched DEPTH k-steps ahead
    // ACROSS chunk boundaries -- chunk c + 1 has been in the ring since the barrier that ended chunk c - 1 -- so a chunk no longer
    // starts with an exposed LDS round trip (r03; a tile is 9-12 chunks and a small frame is latency-, not throughput-bound).
    // Per chunk: the request for chunk c + 3 at its top, the ring write of chunk c + 2 (into the slot of chunk c - 1, which every
    // wave left before that barrier) behind its second k-step, and ONE barrier at its end that waits only for that write: lgkmcnt
    // counts in order, so the 2 x 3 fragment reads issued after it may stay in flight.
    constexpr int DEPTH = G == 2 ? 2 : 3;          // 3 h8 per k-step in flight; the register budget is 128 (G = 2) / 170 (G = 1)
    h8 fa[DEPTH], fb0[DEPTH], fb1[DEPTH];
    auto fetch = [&](int step) {        // compile-time step; a branch chunk reads the centre tap whichever branch it is
        const int c = step >> 2, sk = step & 3, sl = step % DEPTH;
        const int dy = c < 9 ? c / 3 : 1, dx = c < 9 ? c % 3 : 1;
        const char* b_lane = sR + (c % S_RING) * S_CHUNK + lane * 16;
        fa[sl] = *reinterpret_cast<const h8*>(a_lane + dy * RSB + dx * PSB + 32 * sk);
        fb0[sl] = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 0) * UNIT);
        fb1[sl] = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 1) * UNIT);
    };
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) fetch(k);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (PAR && c == 6 && a.par_flags) {          // first use of the flags: chunk 9 is requested below
            const int f0 = __builtin_amdgcn_readfirstlane(pfl_v) & 7;
            const int f1 = f0 & (f0 - 1), f2 = f1 & (f1 - 1);
            ncr = 9 + __builtin_popcount(f0);
            bs0 = f0 ? __builtin_ctz(f0) : 0;
            bs1 = f1 ? __builtin_ctz(f1) : 0;
            bs2 = f2 ? __builtin_ctz(f2) : 0;
        }
        if (PAR && c >= 9 && c >= ncr) break;
        if (c + 3 < NC && (!PAR || c + 3 < ncr)) {
            const f32x4* g = (PAR && c + 3 >= 9) ? wgp + bsel(c + 3 - 9) * 512 : chunk_ptr(c + 3);
#pragma unroll
            for (int i = 0; i < WPT; ++i) wreg[(c + 1) & 1][i] = g[tt + NT_ * i];
        }
        _Float16 pj = (_Float16)1.f;
        if (PAR && c >= 9) {
            if (c == 9) bias_gamma();                  // (conv + bias) * gamma BEFORE the 1x1 partition branches
            const int bi = bsel(c - 9);
            pj = (_Float16)(bi == 0 ? pv[0] : (bi == 1 ? pv[1] : pv[2]));
        }
#pragma unroll
        for (int sk = 0; sk < 4; ++sk) {
            const int step = c * 4 + sk, sl = step % DEPTH;
            h8 av = fa[sl];
            const h8 b0 = fb0[sl], b1 = fb1[sl];
            if (PAR && c >= 9) av *= pj;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b1, acc[1], 0, 0, 0);
            if (step + DEPTH < NC * 4) fetch(step + DEPTH);       // (past the tile's last chunk: unused stale bytes)
            if (sk == 1 && c + 2 < NC && (!PAR || c + 2 < ncr)) {
                char* d = sR + ((c + 2) % S_RING) * S_CHUNK;       // slot of chunk c - 1
#pragma unroll
                for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(d + (tt + NT_ * i) * 16) = wreg[c & 1][i];
            }
            F16_DEAL_READS(2)
            __builtin_amdgcn_sched_barrier(0);
        }
        // the ring write above is older than the 6 fragment reads of k-steps 2 and 3: wait for it, not for them
        asm volatile("s_waitcnt lgkmcnt(6)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // nobody still reads the LDS the epilogue overwrites
    if (!PAR || ncr == 9) bias_gamma();

    if (a.dbg) d_t2 = __builtin_amdgcn_s_memtime();
    // ---- epilogue (the persistent kernel's, one tile): transpose through the dead LDS, activation, residual, whole pixel
    //      rows to HBM
    float* sT = reinterpret_cast<float*>(smem + (grp * 4 + wave) * 8192);
    const f32x4* sT4 = reinterpret_cast<const f32x4*>(sT);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + j * 32 + n0] = acc[j][r];
    asm volatile("" ::: "memory");
    if (OUT16) {
        const int ec8 = lane & 7, ep8 = lane >> 3;
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
            const int gx = tx0 + ep8 + 8 * (i & 1);
            const unsigned o = ((unsigned)(ty0 + 2 * wave + (i >> 1)) * (unsigned)W + (unsigned)gx) * 128u + (unsigned)ec8 * 16u;
            buf_store4(r_out, (live & (gx < W)) ? o : OOB, __builtin_bit_cast(f32x4, pk));
        }
    } else {
        f32x4 rows[EIT];
        h4 hv[EIT];
#pragma unroll
        for (int i = 0; i < EIT; ++i) rows[i] = sT4[(ep + 4 * i) * 16 + ec];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < EIT; ++i) {
            f32x4 v = rows[i];
            v = __builtin_elementwise_max(v, (f32x4)(0.f)) + neg_slope * __builtin_elementwise_min(v, (f32x4)(0.f));
            v += res4[i];
            const int gx = tx0 + ep + 4 * (i & 3);
            const unsigned o = ((unsigned)(ty0 + 2 * wave + (i >> 2)) * (unsigned)W + (unsigned)gx) * 256u + (unsigned)ec * 16u;
            buf_store4(r_out, (live & (gx < W)) ? o : OOB, v);
            if (OUT == 2) hv[i] = to_h4(v);
        }
        if (OUT == 2) store_mirror16(hv, r_out16, ty0 + 2 * wave, tx0, live ? W : 0, ec, ep);
    }
    if (a.dbg && t == 0) {      // diagnostic timeline (pnp_conv3x3_f16_ex): start, prologue end, K loop end, end (shader clock)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        d_t3 = __builtin_amdgcn_s_memtime();
        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * G + grp) * 16;
        d[0] = d_t0;
        d[1] = d_t1;
        d[2] = d_t2;
        d[3] = d_t3;
        d[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
        d[7] = 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The input conv of a branch (basicvsr_net.py:484,515 over the concat of iconvsr_ipb_par.py:90,125) in ONE launch: up to three
// 64-channel sources, all read through their fp16 mirrors, + the RGB frame.  The resident-weight kernel above runs such a conv
// as a chain of single-source launches whose fp32 partial sums go through HBM (512 B per pixel and link); here a block is one
// 8x16 tile (the small-frame kernel's structure: 3 blocks per CU, weight chunks streamed from L2 through the 3-slot ring) and
// walks the sources one after another: source s+1's halo is requested while source s is contracted and replaces it in the A
// tile between two chunks.  BIT-IDENTICAL to the chain: every source is accumulated from zero in its own MFMA chain (k order:
// 9 taps x 4 k-steps, the RGB frame's 3 k-steps behind source 0) and folded as the chain's epilogues do --
// sum = (acc_0 + bias), then sum = acc_s + sum -- activation last.
// ---------------------------------------------------------------------------------------------------------------
struct F16MultiArgs {
    const void* src[3];       // fp16 NHWC64 maps
    const _Float16* w[3];     // 72 units each
    const float* lr4;         // NHWC4 fp32 (RGB0) or nullptr
    const _Float16* wlr;      // 8 units
    const float* bias;
    float* out;               // fp32 NHWC64
    void* out16;              // OUT == 2: fp16 mirror of out
    int H, W, act;
    unsigned long long* dbg;
};
constexpr int M_LOFF = PW * PSB;            // RGB halo row (18 px x 8 B = 144 B) in the padding behind an A-tile row (2816 - 2592 = 224 B)
static_assert(M_LOFF + PW * 8 <= RSB, "the RGB halo row fits behind the A-tile row");

template <int NW, bool LR4, int OUT>
__global__ __launch_bounds__(256, 3) void conv3x3_f16_multi_kernel(const F16MultiArgs a) {
    static_assert(OUT == 0 || OUT == 2, "fp32 output, optionally with its fp16 mirror");
    constexpr int AIT = AIT16, CPP = 8;
    constexpr int NC = 9 * NW + (LR4 ? 1 : 0);             // weight chunks: source 0's taps, [RGB], source 1's taps, ...
    constexpr int WPT = 2;                                 // float4 of a weight chunk per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m = lane & 31, h = lane >> 5, my = m >> 4, mx = m & 15;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TW - 1) / TW;
    int tile;
    {   // XCD-aware remap: each XCD walks a contiguous band of tiles (halo rows meet in its L2)
        const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7;
        const int q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    char* const sA = smem;
    char* const sR = smem + A_BYTES;
    unsigned long long d_t0 = 0, d_t1 = 0, d_t2 = 0, d_t3 = 0;
    if (a.dbg) d_t0 = __builtin_amdgcn_s_memtime();

    const unsigned map16 = (unsigned)H * (unsigned)W * 128u;
    __amdgpu_buffer_rsrc_t r_src[NW];
#pragma unroll
    for (int s = 0; s < NW; ++s) r_src[s] = make_rsrc(a.src[s], map16);
    const __amdgpu_buffer_rsrc_t r_lr = make_rsrc(LR4 ? (const void*)a.lr4 : a.src[0], LR4 ? map16 / 8 : 0);
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(a.out, map16 * 2);
    const __amdgpu_buffer_rsrc_t r_out16 = make_rsrc(OUT == 2 ? a.out16 : (void*)a.out, OUT == 2 ? map16 : 0);

    // ---- halo staging: fp16 source -> A tile verbatim (16-byte slot i = t + 256 k is channel group i % 8 of halo pixel i / 8)
    f32x4 areg[AIT];
    const unsigned hbase = (unsigned)((ty0 - 1) * W + (tx0 - 1)) * 128u;
    auto halo_load = [&](int s) {
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int i = t + 256 * k;
            const int pix = i / CPP, cs = i % CPP;
            const int ry = pix / PW, rx = pix - ry * PW;
            const bool ok = (pix < NPIX) & ((unsigned)(tx0 - 1 + rx) < (unsigned)W);  // rows outside the image leave the descriptor by themselves
            areg[k] = buf_load4(r_src[s], ok ? hbase + (unsigned)(ry * W + rx) * 128u + (unsigned)cs * 16u : OOB);
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int k = 0; k < AIT; ++k) {
            const int i = t + 256 * k;
            const int pix = i / CPP, cs = i % CPP;
            const int ry = pix / PW, rx = pix - ry * PW;
            if (pix < NPIX) *reinterpret_cast<f32x4*>(sA + ry * RSB + rx * PSB + cs * 16) = areg[k];
        }
    };
    halo_load(0);
    f32x4 lreg = {0.f, 0.f, 0.f, 0.f};
    if (LR4) {
        const int ry = t / PW, rx = t - ry * PW;
        const bool ok = (t < NPIX) & ((unsigned)(tx0 - 1 + rx) < (unsigned)W);
        lreg = buf_load4(r_lr, ok ? (unsigned)((ty0 - 1 + ry) * W + (tx0 - 1 + rx)) * 16u : OOB);
    }
    auto chunk_ptr = [&](int c) -> const f32x4* {          // 512 float4 per chunk; c is a compile-time constant after unrolling
        if (c < 9) return reinterpret_cast<const f32x4*>(a.w[0]) + c * 512;
        if (LR4 && c == 9) return reinterpret_cast<const f32x4*>(a.wlr);
        const int cc = c - (LR4 ? 10 : 9);
        return reinterpret_cast<const f32x4*>(a.w[(1 + cc / 9) < NW ? 1 + cc / 9 : NW - 1]) + (cc % 9) * 512;
    };
    f32x4 wreg[2][WPT];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const f32x4* g = chunk_ptr(c);
        f32x4 v[WPT];
#pragma unroll
        for (int i = 0; i < WPT; ++i) v[i] = g[t + 256 * i];
#pragma unroll
        for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(sR + c * S_CHUNK + (t + 256 * i) * 16) = v[i];
    }
    {
        const f32x4* g = chunk_ptr(2);
#pragma unroll
        for (int i = 0; i < WPT; ++i) wreg[0][i] = g[t + 256 * i];
    }
    constexpr int EIT = 8;
    const int ec = lane & 15, ep = lane >> 4, n0 = lane & 31;
    const float neg_slope = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.1f);
    float bco[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bco[j] = a.bias ? a.bias[j * 32 + n0] : 0.f;
    halo_store();
    if (LR4 && t < NPIX) {
        const int ry = t / PW, rx = t - ry * PW;
        *reinterpret_cast<h4*>(sA + ry * RSB + M_LOFF + rx * 8) = to_h4(lreg);
    }
    lds_barrier();
    if (a.dbg) d_t1 = __builtin_amdgcn_s_memtime();

    const char* a_lane = sA + (2 * wave + my) * RSB + mx * PSB + 16 * h;
    // RGB source: k = 16 s + 8 h + j  ->  tap 4 s + 2 h + (j >> 2), channel j & 3; taps beyond 8 carry zero weights and re-read
    // tap 8 (finite values)
    const char* l_lane = sA + (2 * wave + my) * RSB + M_LOFF + mx * 8;
    auto l_off = [&](int sk, int u) -> int {
        auto tap_off = [](int tap) {
            tap = tap > 8 ? 8 : tap;
            return (tap / 3) * RSB + (tap % 3) * 8;
        };
        return h ? tap_off(4 * sk + 2 + u) : tap_off(4 * sk + u);
    };
    f32x16 acc[2], sum[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[j][r] = 0.f;
            sum[j][r] = 0.f;
        }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const bool rgb = LR4 && c == 9;
        const int cw = (LR4 && c > 9) ? c - 1 : c;             // index among the wide chunks
        const int s = rgb ? 0 : cw / 9, tap = cw % 9;
        if (!rgb && tap == 0 && s + 1 < NW) halo_load(s + 1);   // the next source's halo travels while this one is contracted
        if (c + 3 < NC) {
            const f32x4* g = chunk_ptr(c + 3);
#pragma unroll
            for (int i = 0; i < WPT; ++i) wreg[(c + 1) & 1][i] = g[t + 256 * i];
        }
        const char* b_lane = sR + (c % S_RING) * S_CHUNK + lane * 16;
        if (!rgb) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int sk = 0; sk < 4; ++sk) {
                const h8 av = *reinterpret_cast<const h8*>(a_lane + dy * RSB + dx * PSB + 32 * sk);
                const h8 b0 = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 0) * UNIT);
                const h8 b1 = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 1) * UNIT);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b1, acc[1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int sk = 0; sk < 3; ++sk) {
                const h4 lo = *reinterpret_cast<const h4*>(l_lane + l_off(sk, 0));
                const h4 hi = *reinterpret_cast<const h4*>(l_lane + l_off(sk, 1));
                const h8 av = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                const h8 b0 = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 0) * UNIT);
                const h8 b1 = *reinterpret_cast<const h8*>(b_lane + (sk * 2 + 1) * UNIT);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b1, acc[1], 0, 0, 0);
            }
        }
        if (c + 2 < NC) {
            char* d = sR + ((c + 2) % S_RING) * S_CHUNK;       // slot of chunk c - 1: every wave left it at the previous barrier
#pragma unroll
            for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(d + (t + 256 * i) * 16) = wreg[c & 1][i];
        }
        // end of a source's own MFMA chain (source 0's includes the RGB frame): fold it as the launch chain's epilogues do
        const bool src_done = rgb || (tap == 8 && !(LR4 && s == 0));
        if (src_done) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sum[j][r] = (s == 0) ? (acc[j][r] + bco[j]) * 1.f : acc[j][r] + sum[j][r];
                    acc[j][r] = 0.f;
                }
        }
        lds_barrier();
        if (!rgb && tap == 8 && s + 1 < NW) {                  // every wave has left source s's A tile: bring in source s + 1
            halo_store();
            lds_barrier();
        }
    }

    if (a.dbg) d_t2 = __builtin_amdgcn_s_memtime();
    // ---- epilogue: transpose through the dead LDS, activation, whole pixel rows to HBM (+ the fp16 mirror)
    float* sT = reinterpret_cast<float*>(smem + wave * 8192);
    const f32x4* sT4 = reinterpret_cast<const f32x4*>(sT);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + j * 32 + n0] = sum[j][r];
    asm volatile("" ::: "memory");
    {
        f32x4 rows[EIT];
        h4 hv[EIT];
#pragma unroll
        for (int i = 0; i < EIT; ++i) rows[i] = sT4[(ep + 4 * i) * 16 + ec];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < EIT; ++i) {
            f32x4 v = rows[i];
            v = __builtin_elementwise_max(v, (f32x4)(0.f)) + neg_slope * __builtin_elementwise_min(v, (f32x4)(0.f));
            const int gx = tx0 + ep + 4 * (i & 3);
            const unsigned o = ((unsigned)(ty0 + 2 * wave + (i >> 2)) * (unsigned)W + (unsigned)gx) * 256u + (unsigned)ec * 16u;
            buf_store4(r_out, gx < W ? o : OOB, v);
            if (OUT == 2) hv[i] = to_h4(v);
        }
        if (OUT == 2) store_mirror16(hv, r_out16, ty0 + 2 * wave, tx0, W, ec, ep);
    }
    if (a.dbg && t == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        d_t3 = __builtin_amdgcn_s_memtime();
        unsigned long long* d = a.dbg + (size_t)blockIdx.x * 16;
        d[0] = d_t0;
        d[1] = d_t1;
        d[2] = d_t2;
        d[3] = d_t3;
        d[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
        d[7] = 1;
    }
}

// fp32 B image (common.h) -> fp16 image, chunk by chunk: element (s, nt, lane = (h, n), j) of the fp16 chunk is
// input channel k = 16 s + 8 h + j, i.e. fp32 element ((k >> 3) * NTB + nt, ((k >> 2) & 1) * 32 + n, k & 3).
__global__ __launch_bounds__(256) void f16_image_kernel(const float* __restrict__ src, _Float16* __restrict__ dst,
                                                       int ntb, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int per_chunk = PNP_CHUNK_Q * ntb * 256;
    const long chunk = idx / per_chunk;
    const int rem = (int)(idx - chunk * per_chunk);
    const int j = rem & 7, lane = (rem >> 3) & 63, nt = (rem >> 9) % ntb, s = rem / (512 * ntb);
    const int n = lane & 31, hh = lane >> 5;
    const int k = 16 * s + 8 * hh + j;
    const float v = src[chunk * per_chunk + (((k >> 3) * ntb + nt) * 64 + ((k >> 2) & 1) * 32 + n) * 4 + (k & 3)];
    dst[idx] = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
}

int f16_grid(int grid_y) {
    static PnpPerDevice once;
    int cus = 256;
    (void)once.run([](int dev, int& v) {
        v = 256;
        return hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    }, &cus);
    int g = cus / grid_y;
    g -= g % 8;
    return g < 8 ? 8 : g;
}

template <bool PAR, bool LR4, bool SRC16, int OUT, bool RGB = false>
int launch_one(const F16Args& fa, int grid_y, hipStream_t stream) {
    auto kern = conv3x3_f16_kernel<PAR, LR4, SRC16, OUT, RGB>;
    static PnpPerDevice once;
    const hipError_t attr_err = once.run([&](int, int&) {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   LDS_BYTES);
    });
    if (attr_err != hipSuccess) return (int)attr_err;
    hipLaunchKernelGGL(kern, dim3(f16_grid(grid_y), grid_y), dim3(512), LDS_BYTES, stream, fa);
    return (int)hipGetLastError();
}

}  // namespace

int launch_f16_image(const float* src, void* dst, int nchunks, int ntb, hipStream_t stream) {
    if (nchunks < 1 || (ntb != 1 && ntb != 2)) return PNP_ERR_BAD_ARG;
    const long total = (long)nchunks * pnp_chunk_floats(ntb);
    hipLaunchKernelGGL(f16_image_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src,
                       reinterpret_cast<_Float16*>(dst), ntb, total);
    return (int)hipGetLastError();
}

// single-source 64 -> 64 launches on frames with fewer than 1024 tiles (what the persistent design cannot amortise)
static bool f16_small_eligible(const ConvArgs& a, int grid_y) {
    if (a.no_small16) return false;
    const long tiles = (long)((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    return tiles < 1024 && grid_y == 1 && a.nsrc == 1 && a.src_c[0] == 64 && a.out_mode == 0;
}

template <int NW, bool LR4, int OUT>
static int launch_multi_t(const F16MultiArgs& fa, hipStream_t stream) {
    auto kern = conv3x3_f16_multi_kernel<NW, LR4, OUT>;
    static PnpPerDevice once;
    const hipError_t attr_err = once.run([&](int, int&) {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, s_lds_bytes(1));
    });
    if (attr_err != hipSuccess) return (int)attr_err;
    const int tiles = ((fa.W + TW - 1) / TW) * ((fa.H + TH - 1) / TH);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), s_lds_bytes(1), stream, fa);
    return (int)hipGetLastError();
}

template <int NW>
static int launch_multi(const F16MultiArgs& fa, hipStream_t stream) {
    if (fa.lr4) return fa.out16 ? launch_multi_t<NW, true, 2>(fa, stream) : launch_multi_t<NW, true, 0>(fa, stream);
    return fa.out16 ? launch_multi_t<NW, false, 2>(fa, stream) : launch_multi_t<NW, false, 0>(fa, stream);
}

template <bool PAR, bool SRC16, int OUT, int G>
static int launch_small_g(const F16Args& fa, hipStream_t stream) {
    auto kern = conv3x3_f16_small_kernel<PAR, SRC16, OUT, G>;
    static PnpPerDevice once;
    const hipError_t attr_err = once.run([&](int, int&) {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, s_lds_bytes(G));
    });
    if (attr_err != hipSuccess) return (int)attr_err;
    const int tiles = ((fa.W + TW - 1) / TW) * ((fa.H + TH - 1) / TH);
    hipLaunchKernelGGL(kern, dim3((tiles + G - 1) / G), dim3(256 * G), s_lds_bytes(G), stream, fa);
    return (int)hipGetLastError();
}

template <bool PAR, bool SRC16, int OUT>
static int launch_small(const F16Args& fa, hipStream_t stream) {
    // two tiles per block (shared weight ring, half the weight bytes per tile) once the frame has a tile pair per CU slot
    const int tiles = ((fa.W + TW - 1) / TW) * ((fa.H + TH - 1) / TH);
    return tiles >= S_PAIR_TILES ? launch_small_g<PAR, SRC16, OUT, 2>(fa, stream) : launch_small_g<PAR, SRC16, OUT, 1>(fa, stream);
}

// run-time (par, fp16 source, output mode) -> compile-time template arguments
template <class F>
static int pick3(bool par, bool s16, int om, F f) {
    auto l2 = [&](auto P, auto S) {
        return om == 1 ? f(P, S, std::integral_constant<int, 1>{})
                       : om == 2 ? f(P, S, std::integral_constant<int, 2>{}) : f(P, S, std::integral_constant<int, 0>{});
    };
    auto l1 = [&](auto P) { return s16 ? l2(P, std::true_type{}) : l2(P, std::false_type{}); };
    return par ? l1(std::true_type{}) : l1(std::false_type{});
}

// A conv over several 64-channel sources runs as a chain of single-source launches that accumulate through
// `out` (fp32, added before the activation of the last link): every link keeps its weights resident in LDS.
int launch_conv3x3_f16(const ConvArgs& a, int grid_y, hipStream_t stream) {
    int lr_idx = -1, wide[4], nwide = 0;
    for (int s = 0; s < a.nsrc; ++s) {
        if (a.src_c[s] == 4) lr_idx = s;
        else wide[nwide++] = s;
    }
    // the input conv of a branch with every wide source available as an fp16 mirror: ONE launch (bit-identical to the chain)
    bool all16 = nwide >= 1 && nwide <= 3;
    for (int k = 0; k < nwide; ++k) all16 = all16 && ((a.src_f16 >> wide[k]) & 1);
    if (all16 && (nwide >= 2 || lr_idx >= 0) && !a.no_multi16 && a.out_mode == 0 && grid_y == 1 && !a.wpar_h && !a.gamma &&
        !a.residual && !a.out_f16) {
        F16MultiArgs f;
        for (int k = 0; k < 3; ++k) {
            f.src[k] = k < nwide ? a.src[wide[k]] : nullptr;
            f.w[k] = k < nwide ? reinterpret_cast<const _Float16*>(a.wsrc_h[wide[k]]) : nullptr;
        }
        f.lr4 = lr_idx >= 0 ? a.src[lr_idx] : nullptr;
        f.wlr = lr_idx >= 0 ? reinterpret_cast<const _Float16*>(a.wsrc_h[lr_idx]) : nullptr;
        f.bias = a.bias;
        f.out = a.out;
        f.out16 = a.out16;
        f.H = a.H;
        f.W = a.W;
        f.act = a.act;
        f.dbg = a.dbg;
        return nwide == 1 ? launch_multi<1>(f, stream) : nwide == 2 ? launch_multi<2>(f, stream) : launch_multi<3>(f, stream);
    }
    if (nwide > 1 && a.src_f16) return PNP_ERR_UNSUPPORTED;      // the launch chain reads fp32 sources
    for (int k = 0; k < nwide; ++k) {
        const bool first = k == 0, last = k == nwide - 1;
        F16Args f;
        f.src = a.src[wide[k]];
        f.w = reinterpret_cast<const _Float16*>(a.wsrc_h[wide[k]]);
        f.lr4 = (first && lr_idx >= 0) ? a.src[lr_idx] : nullptr;
        f.wlr = (first && lr_idx >= 0) ? reinterpret_cast<const _Float16*>(a.wsrc_h[lr_idx]) : nullptr;
        f.wpar = reinterpret_cast<const _Float16*>(a.wpar_h);
        f.par = a.par;
        f.par_plane = a.par_plane;
        f.bias = first ? a.bias : nullptr;
        f.gamma = a.gamma;
        f.residual = first ? (nwide == 1 ? a.residual : nullptr) : a.out;
        f.lr = a.lr;
        f.lr_plane = a.lr_plane;
        f.res_pre = first ? 0 : 1;
        f.out = a.out;
        f.w_ystride = a.w_ystride;
        f.bias_ystride = a.bias_ystride;
        f.H = a.H;
        f.W = a.W;
        f.act = last ? a.act : 0;
        f.out_mode = a.out_mode;
        f.out_cstride = a.out_cstride;
        f.dbg = a.dbg;
        f.out16 = last ? a.out16 : nullptr;
        f.par_flags = a.par_flags;
        // fp16 maps: single-source launches only (conv_f16_eligible).  om: 0 fp32 out, 1 fp16 out, 2 fp32 out + fp16 mirror
        const bool s16 = (a.src_f16 >> wide[k]) & 1;
        const int om = a.out_f16 ? 1 : (f.out16 ? 2 : 0);
        int rc;
        const bool hp = f.wpar != nullptr;
        if (f16_small_eligible(a, grid_y) && !(om == 1 && f.residual)) {
            rc = pick3(hp, s16, om, [&](auto P, auto S, auto O) {
                return launch_small<decltype(P)::value, decltype(S)::value, decltype(O)::value>(f, stream);
            });
            if (rc) return rc;
            continue;
        }
        if (a.out_mode == 2 || a.out_mode == 3)
            rc = s16 ? launch_one<false, false, true, 0, true>(f, grid_y, stream)
                     : launch_one<false, false, false, 0, true>(f, grid_y, stream);
        else if (f.lr4) rc = om == 2 ? launch_one<false, true, false, 2>(f, grid_y, stream) : launch_one<false, true, false, 0>(f, grid_y, stream);
        else
            rc = pick3(hp, s16, om, [&](auto P, auto S, auto O) {
                return launch_one<decltype(P)::value, false, decltype(S)::value, decltype(O)::value>(f, grid_y, stream);
            });
        if (rc) return rc;
    }
    return PNP_OK;
}
