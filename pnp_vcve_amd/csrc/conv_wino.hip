// Winograd F(2x2,3x3) form of the 64 -> 64 channel 3x3 convs (BAE block halves, conv_hr, input convs) on the fp32 matrix pipe of gfx950.
//
// Why: the direct implicit-GEMM kernels (conv_mfma.hip / conv_persist.hip) keep the chip inside fp32 MFMAs for > 99 % of a 720p clip
// at 0.80 of the (power-limited) matrix peak -- the only lever left is fewer MFMAs per output pixel.  F(2x2,3x3) computes a 2x2
// output tile from a 4x4 input patch with 16 channel contractions instead of 36: 2.25x fewer matrix FLOPs for
//   conv3x3(x; g) = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A         (Lavin & Gray 2015; B, G, A below)
// The arithmetic is still fp32 (exact products, fp32 accumulation); the result differs from the direct form by summation
// order and by the cancellation of the +-1 input transform: 5e-7 abs on unit-scale data (tools/ubench/ub_winograd.hip), the gates are
// in tests/test_gpu_wino.py.  Reference arithmetic restated: mmedit/models/common/sr_backbone_utils.py:304-333 (block),
// basicvsr_net.py:506-519 (branch), iconvsr_ipb_par.py:144 (conv_hr).
//
// Structure (measured first as a micro-benchmark, profiles/r05_ub_winograd_*.txt):
//   * block = 4 waves, ONE block per CU: all 256 accumulator registers of a lane hold 16 transform positions x 4 N tiles of
//     v_mfma_f32_16x16x4_f32 (M = 16 Winograd tiles, N = 16 channels).  Block tile = 16x16 output pixels = 8x8 Winograd tiles;
//     wave w owns the 8x8-pixel quadrant (4x4 tiles) w of it and ALL 64 output channels, so the input transform is computed once per
//     (tile, channel) and the output transform needs no exchange between waves.  Lane (m = lane & 15, kq = lane >> 4) = tile m, k-quarter kq.
//   * K outermost: 4 steps of 16 input channels (lane: channels 16 s + 4 kq + j, j = the four MFMA k-steps).  Per step a lane reads
//     its 4x4 patch (16 ds_read_b128), forms V = B^T d B with 32 float4 adds -- rolled over the step's chunks, row i of V being
//     rewritten for step s+1 right after chunk i of step s has consumed it -- and issues 16 positions x 4 N tiles x 4 = 256 MFMAs.
//     fp32 MFMAs execute on the vector ALUs: a gap between two MFMAs that holds any VALU instruction costs ~9 cycles of matrix time
//     + 4 per instruction, LDS / buffer / scalar instructions nothing (tools/ubench/ub_valu_gap.hip), so addresses are buffer
//     descriptors + scalar offsets + immediates, accumulators start from an inline-constant zero C operand instead of being cleared,
//     and products / transforms are lumped into few gaps where the register budget allows.
//   * transformed weights (256 KB per conv; gamma of a dynamic conv folded in per frame, see launch_wino_images) stream L2 -> registers
//     -> a 4-slot LDS ring in 16 chunks of 16 KB (chunk = step s, position row i), requested three chunks ahead; B fragments are read
//     one position ahead of the MFMAs, across chunk seams too (chunk c+1 is visible since the barrier at the top of chunk c).
//   * the halo tile (18 x 18 pixels x 64 channels = 81 KB) lives in LDS as four 16-channel slabs; K-outer order frees slab s after
//     step s-1's reads, so the NEXT tile's slab s replaces it during step s: no second halo buffer, no serial halo fill between tiles.
//     Pixels of a halo row are stored even columns first: the 8 tiles of a row then read 8 consecutive 64-B pixels (2-way instead of
//     8-way bank conflicts on the patch reads).
//   * partition branches (front half): the three per-pixel-weighted 1x1 convs accumulate straight into the transform domain -- a value
//     added to position (0,0) / -(0,3) / -(3,0) / (3,3) reaches exactly output pixel (0,0) / (0,1) / (1,0) / (1,1) through A^T . A --
//     as a fifth chunk per step whose A operand is par_j(pixel) * x(centre pixel).  A wave's quadrant is one 8x8 codec block, so on a one-hot
//     map each wave runs ONE branch: which, it decides itself from the values it loaded (a zero plane adds exact zeros).
//   * epilogue: Y = A^T M A in registers, + bias (* gamma), activation; one N tile at a time through 4 KiB of LDS per wave (the ring
//     slot the tile's last chunk has just left) so that residual loads and stores move 16 B per lane (64-B channel runs per pixel).
//   * quadrant units: the tiles beyond an XCD band's whole rounds are cut into four 8x8 units, one per block, the four waves splitting
//     the output channels (the kernel's tail); conv3x3_wino_quad_kernel / _quad_ms_kernel run whole small frames that way.  Same
//     arithmetic in the same order per accumulator: a pixel's value does not depend on the form that computed it.
//   * the input convs over the virtual concat [frame, wide sources ...] are the MS instantiation (frame = one k-step, then one
//     16-chunk segment per source into the same accumulators).
#include "conv_mfma.h"
#include <cstring>
#include <type_traits>

namespace {

template <int N>
using I = std::integral_constant<int, N>;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int HP = 18, NPX = HP * HP;
constexpr int SLAB_B = NPX * 64;              // bytes per 16-channel slab of the halo tile
constexpr int RING_B = 4 * 16384;             // four 16-KiB weight chunks
constexpr int PV_B = RING_B + 4 * SLAB_B;      // [plane j][thread] float4: the partition values of the thread's 4 pixels, signed (PAR); 3 x 4 KiB
constexpr int BG_B = PV_B + 256 * 48;         // 64 floats: bias * gamma
constexpr int DUMP_B = BG_B + 256;            // 256 B per wave nobody reads: where the residual warm-up loads land (LDS-DMA: no destination register)
constexpr int WINO_LDS = DUMP_B + 1024;       // 162048 of 163840
constexpr unsigned OOBW = 0xFFFFFFF0u;
#ifndef WINO_FOLD
#define WINO_FOLD 1       // A/B switch: a wave-uniform partition plane folded into the B fragments instead of run as MFMAs
#endif
#ifndef WINO_JIT_ROWS
#define WINO_JIT_ROWS 1   // A/B switch: the two patch rows a V row needs are read from the slab in the chunk that transforms them (32 transient
#endif                    // registers) instead of living in four row arrays across the K loop and the epilogue (48-64 registers)
#ifndef WINO_MS_FLAT
#define WINO_MS_FLAT 0    // A/B switch (MS): 1 = no run-time `last source?` branch inside a chunk (descriptor / strides of the next segment as per-segment
#endif                    // scalars, the RGB halo re-fetched by every segment); 0 = round 5's branches
#ifndef WINO_MS_LAUNDER
#define WINO_MS_LAUNDER 1 // A/B switch (MS): the thread id laundered per segment for the halo offsets (2: for every per-lane constant of the chunks)
#endif
#ifndef WINO_SEAM_CUT
#define WINO_SEAM_CUT 2   // A/B switch: the rolling input transform stops at the tile seam -- V rows 0-2 of the next tile's first step are made behind
#endif                    // the epilogue instead of in the tile's last three chunks (48 registers do not live across it).  1 = every single-source
                          // kernel: 86.1 -> 72.8 frames/s (r06; r05 found the same: the allocator answers with 144-264 B of scratch); 2 = the branch
                          // kernels only (default: 120 -> 56 B of scratch there, front half 414 -> 407 us); 0 = nowhere
#ifndef WINO_LUMP
#define WINO_LUMP 1       // A/B switch (non-branch kernels): the 8 float4 sums of a chunk's input transform in this many MFMA gaps.  8 = one sum per gap
#endif                    // (round 5: lumped, the residual kernel spilled); r06, registers to spare: 8 / 4 / 2 / 1 gaps = 88.2 / 88.8 / 88.9 / 89.2 frames/s
#ifndef WINO_QUAD_AHEAD
#define WINO_QUAD_AHEAD 2   // A/B switch (conv3x3_wino_quad_kernel): B fragments requested this many steps ahead.  1 / 2 / 3 = 2076 / 2130 / 2040
#endif                      // frames/s on 7x3x128x128 clips (3: the fourth fragment set lives in AGPRs, moved back and forth)
#ifndef WINO_PK_FOLD
#define WINO_PK_FOLD 0     // A/B switch (tile bodies): the folded plane's FMAs on the B fragments as v_pk_fma_f32 (8 instead of 16 per gap):
#endif                    // bit-identical, 88.55 -> 88.15 frames/s (r06: fewer VALU cycles, denser MFMA issue, lower clock -- DESIGN.md 3.1)
#ifndef WINO_RING_DMA
#define WINO_RING_DMA 1   // A/B switch (plain / residual / fold-only kernels): the weight chunks arrive in the ring as LDS-DMA loads too -- no staging
#endif                    // registers (16), no ring write, ONE counted wait per chunk placed a chunk and a half behind the request
#ifndef WINO_HALO_DMA
#define WINO_HALO_DMA 1   // A/B switch: the next tile's halo slabs arrive as LDS-DMA loads (no staging registers, no ds_write, no wait for the data in
#endif                    // the instruction stream) instead of load -> register -> ds_write a chunk later
#ifndef WINO_QUAD
#define WINO_QUAD 1      // A/B switch of the quadrant units (conv3x3_wino_kernel's tail)
#endif

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
// (A 16-byte store with its scalar offset in an SGPR -- this one -- followed at once by a vector-ALU write of one of its data registers
//  stores the NEW value in lanes 12-15 of each row: a gfx950 hazard LLVM does not pad for this store form.  build_native.py pads the
//  listing instead of the source -- any statement added here costs the tile loop its register allocation; pnp_vcve_amd/isa_hazards.py.)
__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void bstore1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}
// b + c * w as two v_pk_fma_f32 (the same fused multiply-add per element as __builtin_elementwise_fma: bit-identical).  The unit is
// compiled without packed fp32 ops (build_native.py), so this is inline asm: the compile step passes it through to the listing and the
// assembler step of the build knows the instruction.  Half the vector-ALU instructions of the folded plane (WINO_PK_FOLD).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 pk_fma4(f32x2 c, f32x4 w, f32x4 b) {
    f32x2 b01 = {b[0], b[1]}, b23 = {b[2], b[3]};
    const f32x2 w01 = {w[0], w[1]}, w23 = {w[2], w[3]};
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(b01) : "v"(c), "v"(w01));
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(b23) : "v"(c), "v"(w23));
    return f32x4{b01[0], b01[1], b23[0], b23[1]};
}
// Float4 sums written element by element.  As `a - b` on the vector type the four lanes stay one 128-bit value and the register
// allocator needs an aligned quad for every intermediate of the rolling input transform; as four scalar ops they are independent
// 32-bit values: 13 instead of 33 / 37 spill slots in the plain kernels, back half 373 -> 351 us, conv_hr 336 -> 318 us (same session).
// The branch kernels pin V as quads anyway (asm volatile "+v") and lose 1.4 % with this form: they keep the vector ops.
__device__ __forceinline__ f32x4 add4(f32x4 a, f32x4 b) { return f32x4{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}; }
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) { return f32x4{a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]}; }
// LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the weight / halo requests kept in flight
__device__ __forceinline__ void lds_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt [3:0] and [15:14], expcnt [6:4] = 7, lgkmcnt [11:8] = 15: "do not wait")
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is six bits");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

struct WinoArgs {
    const float* src;       // NHWC64
    const float* U;         // 16 chunks x 4096 floats (launch_wino_images)
    const float* Upar;      // PAR: 4 steps x 3 branches x 1024 floats (launch_wino_par_image)
    const float* par;       // 3 planes
    long par_plane;
    const int* par_flags;   // per 8x16 tile (launch_par_tile_flags) or nullptr
    const float *bias, *gamma, *residual;
    float* out;
    int H, W, act;
    unsigned long long* dbg;
    // MS (the input conv over the virtual concat [lr, wide sources...], iconvsr_ipb_par.py:90,125): src / U above are unused
    const float* srcs[3];   // the 64-channel sources, NHWC64
    unsigned u_off[3];      // byte offset of each source's 16-chunk image from ubase
    const float* ubase;
    int nsrc;               // 1..3
    const float* rgb;       // the frame as (H,W,4) RGB0
    const float* Urgb;      // launch_wino_rgb_image: 4 chunks of 4 KiB
    const int* gate;        // nullptr, or one word (ConvArgs::par_any): the launch runs iff ((*gate & gate_mask) != 0) == (gate_want != 0)
    int gate_mask, gate_want;
    int quad;               // the tiles beyond an XCD band's whole rounds are worked on as four 8x8 quadrants by four blocks (see the kernel's tail)
};

// FO ("fold only"): the front half of a frame whose EVERY 8x8 quadrant is all zero or carries one constant partition plane (the gate
// of launch_conv3x3_wino: par_frame_any's bit 3): the plain kernel's chunk structure -- no branch chunks, no branch MFMAs, accumulators
// from an inline zero -- plus, per step, the quadrant's plane folded into the B fragments of positions (1,1) (1,2) (2,1) (2,2) exactly
// as the branch kernel does it for such a wave (same FMAs, same MFMA order: bit-identical values).  The plane and its value come from
// one pixel of the quadrant; its 1x1 fragments straight from L2.
template <bool PAR, bool RES, bool MS, bool FO = false>
__device__ __forceinline__ void wino_tile_body(const WinoArgs& a) {
    static_assert(!MS || (!PAR && !RES), "the multi-source form is the input conv: no branches, no residual");
    static_assert(!FO || (!PAR && !MS), "fold-only is the plain structure");
    // (declared HERE, not handed in by the kernel: as a pointer parameter the LDS base stops being the constant 0 for address folding,
    //  and the K loop keeps four more base registers -- in scratch)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + 15) >> 4, ntiles = tiles_x * ((H + 15) >> 4);
    // MS with the ring as LDS-DMA loads: no run-time `last source?` branch inside a chunk (every chunk's wait counts its own requests)
    constexpr bool MSF = WINO_MS_FLAT || (MS && WINO_RING_DMA && WINO_HALO_DMA);
    // the rolling input transform cut at the tile seam: 1 = every single-source kernel, 2 = the branch kernels only
    constexpr bool SEAM = !MS && WINO_JIT_ROWS && (WINO_SEAM_CUT == 1 || (WINO_SEAM_CUT == 2 && PAR));
    if (a.gate) {      // (block-uniform: a scalar load)
        const int gv = __builtin_nontemporal_load(a.gate);
        if (((gv & a.gate_mask) != 0) != (a.gate_want != 0)) return;
    }

    // ---- strip of tiles: XCD x owns a contiguous band, dealt round-robin to its resident blocks (neighbouring halos share its L2)
    int tile, tstep, tend, qtile = -1, qquad = 0;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int bq = ntiles >> 3, br = ntiles & 7;
        const int xbeg = xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq;
        tend = xbeg + bq + (xcd < br ? 1 : 0);
        tile = xbeg + slot;
        tstep = gridDim.x >> 3;
        // 720p: 450 tiles per band on 32 blocks = 14 rounds + 2 tiles, i.e. a fifteenth round with 2 of 32 CUs busy (-6 %).  Those
        // L tiles are cut into 4 L quadrant units instead, one per block, worked on behind the block's last whole tile with the four
        // waves splitting the output channels (a quarter of a tile's MFMAs per wave): the tail costs a third of a round
        const int nband = tend - xbeg, rounds = nband / tstep, left = nband - rounds * tstep;
        if (!MS && a.quad && rounds >= 1 && left > 0 && 4 * left <= tstep) {
            tend = xbeg + rounds * tstep;
            if (slot < 4 * left) {
                qtile = tend + (slot >> 2);
                qquad = slot & 3;
            }
        }
    } else {
        tile = blockIdx.x;
        tstep = gridDim.x;
        tend = ntiles;
    }
    if (tile >= tend) return;

    const unsigned map_bytes = (unsigned)H * (unsigned)W * 256u;
    // halo pixel (ry, rx) of a tile at (y0, x0) is image pixel (y0 - 1 + ry, x0 - 1 + rx): the descriptor's base sits one row and one
    // pixel before the map so that lane offsets stay non-negative; lanes outside the image carry the offset OOBW (-> zeros)
    // (MS: r_src is the source the NEXT halo slabs come from -- the next source of this tile or the first one of the next tile --
    //  re-made per segment; r_u spans every source's image, the segment adds its byte offset)
    __amdgpu_buffer_rsrc_t r_src = rsrc_of(reinterpret_cast<const char*>(MS ? a.srcs[0] : a.src) - ((long)W + 1) * 256, OOBW);
    const __amdgpu_buffer_rsrc_t r_u = rsrc_of(MS ? a.ubase : a.U, MS ? OOBW : 16u * 16384u);
    const __amdgpu_buffer_rsrc_t r_urgb = rsrc_of(MS ? a.Urgb : a.U, 4u * 4096u);
    const __amdgpu_buffer_rsrc_t r_rgb = rsrc_of(reinterpret_cast<const char*>(MS ? a.rgb : a.src) - ((long)W + 1) * 16, OOBW);
    const __amdgpu_buffer_rsrc_t r_up = rsrc_of((PAR || FO) ? a.Upar : a.U, 4u * 12288u);
    const __amdgpu_buffer_rsrc_t r_out = rsrc_of(a.out, map_bytes);
    const __amdgpu_buffer_rsrc_t r_res = rsrc_of(RES ? a.residual : a.src, RES ? map_bytes : 0u);
    const __amdgpu_buffer_rsrc_t r_par = rsrc_of(a.par, PAR ? (unsigned)(3 * a.par_plane * 4) : 0u);

    const unsigned t16 = (unsigned)t * 16u;
    // patch base of this lane's tile (even-columns-first pixel order inside a halo row): tile (ty, tx) -> pixel row 2 ty, column pair tx
    // wave w owns the 8x8-pixel quadrant (w >> 1, w & 1) of the block tile = 4x4 Winograd tiles: one codec partition block, so a one-hot
    // partition map needs ONE of the three branches per wave (decided per wave, below)
    const int ty = 4 * (wave >> 1) + (m >> 2), tx = 4 * (wave & 1) + (m & 3);
    const unsigned dbase0 = RING_B + ((2 * ty) * HP + tx) * 64 + kq * 16, dbase1 = dbase0 + 2 * SLAB_B;
    auto lds4 = [&](unsigned byte) -> f32x4 { return *reinterpret_cast<const f32x4*>(smem + byte); };
    // d[r][c] of slab S: column c of the patch is halo column 2 tx + c = pair tx + (c >> 1) of parity c & 1
    auto patch = [&](auto s_c, int r, int c) -> f32x4 {
        constexpr int S = decltype(s_c)::value;
        const unsigned off = (S & 1) * SLAB_B + (r * HP + (c & 1) * 9 + (c >> 1)) * 64;
        return lds4((S < 2 ? dbase0 : dbase1) + off);
    };

    const float act_lo = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.1f);
    // bias * gamma and the partition values live in LDS, not in registers: the K loop runs at the 256-VGPR limit, and a value
    // spilled to scratch comes back behind an s_waitcnt vmcnt(0) that also waits for every weight / halo request in flight
    // (channel t = 16 n + m at float index 4 m + n: lane m reads its four N tiles' values as one float4)
    if (t < 64) *reinterpret_cast<float*>(smem + BG_B + ((t & 15) * 4 + (t >> 4)) * 4) = (a.bias ? a.bias[t] : 0.f) * (a.gamma ? a.gamma[t] : 1.f);
    // (accumulator register r of N tile nt is tile 4 kq + r -- C/D layout of the 16x16 MFMA -- channel 16 nt + m;
    //  tile (ty', tx') = (2 wave + (kq >> 1), 4 (kq & 1) + r), pixel (a, b) of it: see the epilogue)

    f32x4 acc[16][4];
    f32x4 V[16], d0[4], d1[4], d2[4], d3[4], bf[2][4], breg[4], hreg[3], tt[4];
    unsigned hoff[6];

    auto row_tf = [&](int i) {
        f32x4 tt[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) tt[c] = i == 0 ? d0[c] - d2[c] : (i == 1 ? d1[c] + d2[c] : (i == 2 ? d2[c] - d1[c] : d1[c] - d3[c]));
        V[4 * i + 0] = tt[0] - tt[2];
        V[4 * i + 1] = tt[1] + tt[2];
        V[4 * i + 2] = tt[2] - tt[1];
        V[4 * i + 3] = tt[1] - tt[3];
    };
    // per-lane offsets of the six halo float4 this thread moves per slab (element e = t + 256 i of the slab's LDS image), for the tile at
    // (y0, x0); recomputed per tile from the laundered thread id so that they do not stay live as 12 more registers
    auto halo_offsets = [&](int tq, int y0, int x0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int e = tq + 256 * i;
            e = e < NPX * 4 ? e : NPX * 4 - 1;
            const int pl = e >> 2, quad = e & 3, ry = (pl * 3641) >> 16, col = pl - ry * HP;      // pl / 18 for pl < 324
            const int rx = col < 9 ? 2 * col : 2 * col - 17;
            const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
            const bool inb = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            hoff[i] = inb ? (unsigned)(ry * W + rx) * 256u + (unsigned)quad * 16u : OOBW;
        }
    };
    // partition values of the wave's pixels for the tile at (y0, x0), plane J: global -> LDS (signed), and whether the WAVE needs the
    // branch at all.  One plane per call: the next tile's values are fetched plane by plane in the last three position chunks of a
    // tile (all twelve at once, with their offsets, were what tipped the K loop into scratch spills)
    // partition values of the wave's pixels for the tile at (y0, x0): global -> registers (pv_request), then -> LDS, signed, with the
    // decision whether the WAVE needs each branch at all (pv_finish).  The next tile's values are requested at the top of the epilogue and
    // finished behind it: inside the K loop their twelve registers tipped the branch kernels into scratch spills, and a spill reload is
    // an s_waitcnt vmcnt(0) -- it waits for every weight / halo request in flight
    int need_next = 0, fold_next = -1;
    float foldc_next = 0.f;
    f32x4 pvr[3];
    auto pv_request = [&](int tq_, int y0, int x0) {
        if constexpr (PAR) {
            const int mq = tq_ & 15, wq_ = tq_ >> 6;
            const int py = y0 + 2 * (4 * (wq_ >> 1) + (mq >> 2)), px = x0 + 2 * (4 * (wq_ & 1) + (mq & 3));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (a pixel outside the image takes the value of the nearest one inside: its own product is with x = 0 and its output is
                //  never stored, but a quadrant cut by the frame's edge then looks as constant to the wave as its inside part is -- and
                //  folds, like in the fold-only kernel and in par_tile_flags' bit 6)
                const int gy = min(py + (q >> 1), H - 1), gx = min(px + (q & 1), W - 1);
                const unsigned po = (unsigned)(gy * W + gx) * 4u;
#pragma unroll
                for (int j = 0; j < 3; ++j) pvr[j][q] = bload1(r_par, po, (unsigned)(j * a.par_plane * 4));
            }
        }
    };
    auto pv_finish = [&](int tq_) {
        if constexpr (PAR) {
            need_next = 0;
            int uni = 0;               // bit j: plane j has ONE value on all 64 pixels of the wave
            float uval[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f32x4 v = pvr[j];
                // which branches this WAVE needs: a plane that is zero on all of its 8x8 pixels contributes exact zeros
                const bool nz = v[0] != 0.f || v[1] != 0.f || v[2] != 0.f || v[3] != 0.f;
                if (__builtin_amdgcn_ballot_w64(nz) != 0) need_next |= 1 << j;
                uval[j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v[0])));
                const bool same = v[0] == uval[j] && v[1] == uval[j] && v[2] == uval[j] && v[3] == uval[j];
                if (__builtin_amdgcn_ballot_w64(!same) == 0) uni |= 1 << j;
                v[1] = -v[1];                                     // positions (0,3) and (3,0) enter the output transform negated
                v[2] = -v[2];
                *reinterpret_cast<f32x4*>(smem + PV_B + j * 4096 + tq_ * 16) = v;
            }
            // A one-hot map with one value per codec block (the loader's): exactly one plane is live on the wave's quadrant and constant
            // there.  Its branch  p * conv1x1_j(x(centre))  is the 3x3 conv with the centre tap p w_j, whose Winograd image is
            // p w_j / 4 * [+ -; - +] on positions (1,1) (1,2) (2,1) (2,2): the wave adds that to the B fragments of these four positions
            // (64 FMAs per step) instead of running 64 more MFMAs per step.
            fold_next = -1;
            foldc_next = 0.f;
            if (WINO_FOLD) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (need_next == (1 << j) && (uni >> j & 1)) {
                        fold_next = j;
                        foldc_next = 0.25f * uval[j];
                    }
            }
        }
    };
    // FO: plane and factor of the wave's quadrant of the tile at (y0, x0), from its first pixel (the frame passed the gate: the quadrant
    // is all zero or one constant plane); requested at the top of an epilogue, finished behind it
    float fov[3] = {0.f, 0.f, 0.f};
    int fo_next = -1;
    float foc_next = 0.f;
    auto fo_request = [&](int y0, int x0, int quadrant) {       // quadrant < 0: the wave's own
        if constexpr (FO) {
            const int wv = quadrant >= 0 ? quadrant : __builtin_amdgcn_readfirstlane(t >> 6);
            const int gy = y0 + 8 * (wv >> 1), gx = x0 + 8 * (wv & 1);
#pragma unroll
            for (int j = 0; j < 3; ++j) fov[j] = (gy < H && gx < W) ? a.par[(long)j * a.par_plane + (long)gy * W + gx] : 0.f;
        }
    };
    auto fo_finish = [&]() {
        if constexpr (FO) {
            fo_next = -1;
            foc_next = 0.f;
#pragma unroll
            for (int j = 2; j >= 0; --j)
                if (fov[j] != 0.f) {
                    fo_next = j;
                    foc_next = 0.25f * fov[j];
                }
        }
    };
    // MS: the frame's halo (18 x 18 pixels x RGB0 = 5 KiB) lives where the branch kernels keep their partition values; pixel order as in
    // the slabs (even columns of a row first).  Thread t moves pixels t and t + 256 (clamped to 323: duplicates rewrite the same value)
    f32x4 rgbreg[2];
    float Vr[16];
    auto rgb_request = [&](int tq_, int y0, int x0) {
        if constexpr (MS) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int e = tq_ + 256 * i;
                e = e < NPX ? e : NPX - 1;
                const int ry = (e * 3641) >> 16, col = e - ry * HP, rx = col < 9 ? 2 * col : 2 * col - 17;
                const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
                const bool inb = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                rgbreg[i] = bload4(r_rgb, inb ? (unsigned)(ry * W + rx) * 16u : OOBW, (unsigned)(y0 * W + x0) * 16u);
            }
        }
    };
    // ... inside the K loop as LDS-DMA loads (16 B per lane straight into the halo's place: lane-linear, like rgb_store's layout): with a
    // destination register the two float4 had to live from the request in one chunk to the LDS write in the next, and at the register
    // limit hipcc spilled them right behind the loads -- a wait for the memory latency in the middle of the K loop (r05).  The lanes past
    // pixel 323 re-fetch that pixel and land in the unused rest of the partition-value rows.
    auto rgb_dma = [&](int tq_, int y0, int x0) {
        if constexpr (MS) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int e = tq_ + 256 * i;
                e = e < NPX ? e : NPX - 1;
                const int ry = (e * 3641) >> 16, col = e - ry * HP, rx = col < 9 ? 2 * col : 2 * col - 17;
                const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
                const bool inb = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                auto* dst = (__attribute__((address_space(3))) void*)(smem + PV_B + (tq_ >> 6) * 1024 + i * 4096);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rgb, dst, 16, (int)(inb ? (unsigned)(ry * W + rx) * 16u : OOBW), (int)((unsigned)(y0 * W + x0) * 16u), 0, 0);
            }
        }
    };
    auto rgb_store = [&](int tq_) {
        if constexpr (MS) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int e = tq_ + 256 * i;
                e = e < NPX ? e : NPX - 1;
                *reinterpret_cast<f32x4*>(smem + PV_B + e * 16) = rgbreg[i];
            }
        }
    };
    int ty0 = (tile / tiles_x) * 16, tx0 = (tile % tiles_x) * 16;
    // ---- prologue: whole halo of the first tile, chunks 0..2, first patch
    {
        halo_offsets(t, ty0, tx0);
        const unsigned so = (unsigned)(ty0 * W + tx0) * 256u;
        // every request first (24 halo + 12 weight float4 per thread: the registers are free here), then the LDS writes: one
        // memory latency for the whole prologue instead of seven in a row
        f32x4 hh[4][6], rg[3][4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 6; ++i) hh[s][i] = bload4(r_src, hoff[i], so + s * 64);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool brc = PAR && (c % 5 == 0);
#pragma unroll
            for (int i = 0; i < (MS ? 1 : brc ? 3 : 4); ++i)
                rg[c][i] = MS ? bload4(r_urgb, t16, c * 4096)
                         : brc ? bload4(r_up, t16, (c / 5) * 12288 + i * 4096)
                               : bload4(r_u, t16, (PAR ? (c / 5) * 4 + (c % 5) - 1 : c) * 16384 + i * 4096);
        }
        rgb_request(t, ty0, tx0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                int e = t + 256 * i;
                e = e < NPX * 4 ? e : NPX * 4 - 1;
                *reinterpret_cast<f32x4*>(smem + RING_B + s * SLAB_B + e * 16) = hh[s][i];
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool brc = PAR && (c % 5 == 0);
#pragma unroll
            for (int i = 0; i < (MS ? 1 : brc ? 3 : 4); ++i) *reinterpret_cast<f32x4*>(smem + (c & 3) * 16384 + i * 4096 + t16) = rg[c][i];
        }
        rgb_store(t);
        pv_request(t, ty0, tx0);
        pv_finish(t);
        fo_request(ty0, tx0, -1);
        fo_finish();
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            d0[c] = patch(I<0>{}, 0, c);
            d1[c] = patch(I<0>{}, 1, c);
            d2[c] = patch(I<0>{}, 2, c);
            d3[c] = patch(I<0>{}, 3, c);
        }
        row_tf(0);
        row_tf(1);
        row_tf(2);
    }
    // (the B fragment of lane l, N tile n, position column pj of a chunk in slot z sits at z * 16384 + (pj * 4 + n) * 1024 + l * 16)
    const unsigned bl = (unsigned)lane * 16u;
    if constexpr (!PAR && !MS) {
#pragma unroll
        for (int n = 0; n < 4; ++n) bf[0][n] = lds4(bl + n * 1024);
    }

    unsigned long long dbg_t0 = 0, dbg_r0 = 0, dbg_k = 0, dbg_e = 0;
    if (a.dbg) {
        dbg_t0 = __builtin_amdgcn_s_memtime();
        dbg_r0 = __builtin_amdgcn_s_memrealtime();
    }
    int dbg_n = 0;
    unsigned long long dbg_q0 = 0, dbg_q1 = 0, dbg_q2 = 0;
    int tq = t;
    for (;;) {
        asm volatile("" : "+v"(tq));
        const int ntile = tile + tstep;
        const bool has_next = ntile < tend;
        // (behind the block's last whole tile the "next tile" of the halo pipeline is the tile its quadrant unit belongs to)
        const int ptile = has_next ? ntile : (qtile >= 0 ? qtile : tile);
        const int nty0 = (ptile / tiles_x) * 16, ntx0 = (ptile % tiles_x) * 16;
        const unsigned nso = (unsigned)(nty0 * W + ntx0) * 256u;
        if constexpr (!MS) halo_offsets(tq, nty0, ntx0);
        unsigned tq16 = (unsigned)tq * 16u;
        const int wave_s = __builtin_amdgcn_readfirstlane(tq >> 6);      // the wave's index as a scalar (LDS-DMA destinations go through M0)
        int tqk = tq;            // the thread id the chunks derive their per-lane constants from (MS: laundered once more per segment)
        // MS: per segment (= one 64-channel source): where its weight image starts, where the next segment's does, whether it is the
        // tile's last one (then the next chunks are the next tile's RGB chunks), and the tile origin of the slabs it refills
        unsigned u_so = 0, ref_so = nso, nx_base = 0, nx_cs = 16384, nx_ps = 4096;
        bool last_seg = true;
        __amdgpu_buffer_rsrc_t r_nx = r_u;
        int need = 7, fold = -1;
        float foldc = 0.f;
        f32x4 wj3, wjt[3];      // the folded plane's fragments: N tile 3 in registers, 0-2 parked in the wave's own partition-value rows of LDS
        f32x4 wjl[4];           // FO: the fragments on their way from L2
        unsigned fo_so = 0;
        if constexpr (FO) {
            fold = __builtin_amdgcn_readfirstlane(fo_next);
            foldc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, foc_next)));
            fo_so = (unsigned)(fold > 0 ? fold : 0) * 4096u;
            wj3 = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (PAR) {
            // par_flags != nullptr only ENABLES branch skipping here (the caller's PNP_OPT_PAR_SKIP switch): the decision is per wave,
            // taken from the values themselves when they were loaded (load_pv), not per 8x16 tile
            if (a.par_flags) need = __builtin_amdgcn_readfirstlane(need_next);
            fold = __builtin_amdgcn_readfirstlane(fold_next);
            foldc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, foldc_next)));
            if (fold >= 0) need = 0;      // (the other two planes are zero on the whole quadrant: they would add exact zeros -- and the wave's
                                          //  partition-value rows are about to hold fragments instead)
            wj3 = f32x4{0.f, 0.f, 0.f, 0.f};
            // the four accumulators the branches add into start from zero (the others from an inline-constant zero C operand)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[0][n] = acc[3][n] = acc[12][n] = acc[15][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        // one chunk of the K loop.  C = chunk index inside the tile (slot C & 3); KIND 0: position row PG of step S; KIND 1: the branch
        // chunk of step S (PAR only)
        auto chunk = [&](auto c_c, auto s_c, auto pg_c, auto kind_c) {
            constexpr int C = decltype(c_c)::value, S = decltype(s_c)::value, PG = decltype(pg_c)::value, KIND = decltype(kind_c)::value;
            constexpr int CPT = PAR ? 20 : 16;
            constexpr int NC = (C + 3) % CPT;                        // the chunk requested now (three ahead)
            constexpr bool NBR = PAR && (NC % 5 == 0);
            constexpr int NSTEP = PAR ? NC / 5 : NC / 4, NPG = PAR ? NC % 5 - 1 : NC % 4;
            lds_bar();
            if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < (NBR ? 3 : 4); ++i)
                    breg[i] = NBR ? bload4(r_up, tq16, NSTEP * 12288 + i * 4096) : bload4(r_u, tq16, (NSTEP * 4 + NPG) * 16384 + i * 4096);
            }
            // (the next tile's slab S travels in two halves: requested in position chunks 0 / 1 of step S, written one chunk later)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KIND == 1) {
                // ---- partition branches of step S: A = par_j(pixel) * x(pixel), x = the patch centre (d rows 1, 2 x columns 1, 2 of step S)
                // first the fragments of the following position chunk (visible since the previous barrier)
#pragma unroll
                for (int n = 0; n < 4; ++n) bf[0][n] = lds4(((C + 1) & 3) * 16384 + bl + n * 1024);
                // (WINO_JIT_ROWS: the patch centre of step S straight from slab S -- its refill starts in position chunk 1 of this step)
                f32x4 xc[4];
                if (WINO_JIT_ROWS && need != 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) xc[q] = patch(s_c, 1 + (q >> 1), 1 + (q & 1));
                }
                auto branch = [&](auto j_c) {
                    constexpr int J = decltype(j_c)::value;
                    // (the branch's B fragments go where position 3 of the chunk in front kept its own: bf[1] is free, bf[0] holds
                    //  the first fragments of the chunk behind)
#pragma unroll
                    for (int n = 0; n < 4; ++n) bf[1][n] = lds4((C & 3) * 16384 + (J * 4 + n) * 1024 + bl);
                    const f32x4 pv = lds4(PV_B + J * 4096 + tq16);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4 ax = (WINO_JIT_ROWS ? xc[q] : (q == 0 ? d1[1] : (q == 1 ? d1[2] : (q == 2 ? d2[1] : d2[2])))) * pv[q];
                        // (all four products in ONE gap: left alone each v_mul sits in front of its first MFMA, and a gap with any VALU
                        //  instruction costs ~20 cycles of matrix time before the 4 per instruction)
                        asm volatile("" : "+v"(ax));
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int n = 0; n < 4; ++n) {
                                f32x4& ac = q == 0 ? acc[0][n] : (q == 1 ? acc[3][n] : (q == 2 ? acc[12][n] : acc[15][n]));
                                ac = mfma16(ax[k], bf[1][n][k], ac);
                            }
                    }
                };
                if (need & 1) branch(I<0>{});
                if (need & 2) branch(I<1>{});
                if (need & 4) branch(I<2>{});
                if (fold >= 0) {
                    // the folded plane's 1x1 fragments of this step, wanted again in position rows 1 and 2 (the ring slot is gone by then):
                    // N tiles 0-2 move to the wave's own partition-value rows (three 1-KiB pieces, unused while its plane is folded; the
                    // address is the thread's ring offset + a constant), N tile 3 stays here
#pragma unroll
                    for (int n = 0; n < 3; ++n) wjt[n] = lds4((C & 3) * 16384 + (unsigned)fold * 4096u + n * 1024 + bl);
                    wj3 = lds4((C & 3) * 16384 + (unsigned)fold * 4096u + 3 * 1024 + bl);
#pragma unroll
                    for (int n = 0; n < 3; ++n) *reinterpret_cast<f32x4*>(smem + PV_B + n * 4096 + tq16) = wjt[n];
                }
            }
            if constexpr (KIND == 0) {
                // ---- position row PG of step S: 64 MFMAs, and between them -- ONE thing per MFMA gap, in this order, pinned with a
                //      scheduling barrier per gap (a 16x16x4 fp32 MFMA issues in a few cycles and executes for 32; left to itself hipcc
                //      clusters the loads, the LDS traffic and the transform in front of and behind the MFMAs) --
                //        gaps 0-3 of every position: the B fragments of the next position (across the chunk seam too)
                //        gaps 4-7: the weight requests;  8-10: halo pieces requested a chunk ago -> LDS;  11-13: halo requests
                //        gaps 36-43 (branch kernels: 16, 17): the rolling input transform;  20-27 (28-35): patch rows of step S + 1
                //        gaps 52-55: the weight chunk requested at the top -> ring
                using SN = I<(S + 1) & 3>;
                constexpr int NRING = NBR ? 3 : 4;
                constexpr bool NEXT_IS_BR = PAR && PG == 3;      // the next chunk is a branch chunk: it reads its own fragments
                constexpr int TR = PG == 0 ? 3 : PG - 1;         // the V row rewritten in this chunk (row 3 of step S, or row PG - 1 of S + 1)
                constexpr bool RDMA = WINO_RING_DMA && WINO_HALO_DMA && !PAR;
                constexpr bool XF = !(SEAM && S == 3 && PG >= 1);      // this chunk transforms a V row at all
                // WINO_JIT_ROWS: V row TR = column transform of (patch row RA -/+ patch row RB): row 3 = d1 - d3 of step S (slab S is refilled from
                // position chunk 1 of step S on: still this tile's here), rows 0 / 1 / 2 = d0 - d2 / d1 + d2 / d2 - d1 of step S + 1.  Both rows are
                // read in this chunk and dead behind the transform: no patch row lives across a chunk, the epilogue or the tile seam
                using SR = I<PG == 0 ? S : ((S + 1) & 3)>;
                constexpr int RA = PG == 0 ? 1 : (PG == 1 ? 0 : (PG == 2 ? 1 : 2)), RB = PG == 0 ? 3 : (PG == 1 ? 2 : (PG == 2 ? 2 : 1));
                f32x4 da[4], db[4];
                f32x4 bgv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 64; ++g) {
                    const int pj = g >> 4, k = (g >> 2) & 3, n = g & 3, p = PG * 4 + pj;
                    // step 0, first k-step: C operand = 0 (no accumulator clearing; the four positions the branches use excepted),
                    // or the bias at position (1,1), which A^T . A carries to all four output pixels with weight 1
                    // (MS: the RGB chunks in front of the first source start the accumulators)
                    const bool fresh = !MS && S == 0 && k == 0 && !(PAR && (p == 0 || p == 3 || p == 12 || p == 15));
                    const f32x4 c0 = p == 5 ? f32x4{bgv[n], bgv[n], bgv[n], bgv[n]} : f32x4{0.f, 0.f, 0.f, 0.f};
                    acc[p][n] = mfma16(V[p][k], bf[pj & 1][n][k], fresh ? c0 : acc[p][n]);
                    // ---- the gap behind it
                    if (PAR && PG == 0 && g < 4) {
                        // the weight chunk the branch chunk in front of this one requested: a branch chunk may hold no MFMA at all (no
                        // partition record on the wave's pixels), so it does not wait for its requests -- they land in the ring here
                        constexpr int NCB = (C - 1 + 3) % CPT;
                        *reinterpret_cast<f32x4*>(smem + (NCB & 3) * 16384 + g * 4096 + tq16) = breg[g];
                    }
                    if ((g & 15) < 4 && !(pj == 3 && NEXT_IS_BR)) {
                        // (MS, the tile's last chunk: the next one is an RGB chunk of the next tile, which reads its own fragments -- the read
                        //  here is then of no use, and harmless; a run-time `if (last_seg)` per gap would cut the straight-line schedule)
                        if (MSF || !(MS && C == 15 && pj == 3 && last_seg)) {
                            const unsigned nb = pj < 3 ? (C & 3) * 16384 + (pj + 1) * 4096 : ((C + 1) & 3) * 16384;
                            bf[(pj + 1) & 1][g & 3] = lds4(nb + bl + (g & 3) * 1024);
                        }
                    }
                    if constexpr (MS) {
                        // the chunk three ahead: this source's, the next source's first chunks, or (last source) the next tile's RGB chunks,
                        // which are one 4-KiB piece each.  Branch-free: descriptor, base, chunk and piece stride of "the next segment" are
                        // per-segment scalars (last source: the RGB image with piece stride 0 -- its one piece is fetched four times and lands
                        // in all four quarters of the ring slot, of which an RGB chunk reads the first)
                        if (RDMA && g >= 4 && g < 8) {
                            auto* dst = (__attribute__((address_space(3))) void*)(smem + (NC & 3) * 16384 + (g - 4) * 4096 + wave_s * 1024);
                            if (C < 13) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_u, dst, 16, (int)tq16, (int)(u_so + (C + 3) * 16384 + (g - 4) * 4096), 0, 0);
                            else __builtin_amdgcn_raw_ptr_buffer_load_lds(r_nx, dst, 16, (int)tq16, (int)(nx_base + (C - 13) * nx_cs + (g - 4) * nx_ps), 0, 0);
                        } else if (g >= 4 && g < 8) {
                            if (C < 13) breg[g - 4] = bload4(r_u, tq16, u_so + (C + 3) * 16384 + (g - 4) * 4096);
                            else if (MSF) breg[g - 4] = bload4(r_nx, tq16, nx_base + (C - 13) * nx_cs + (g - 4) * nx_ps);
                            else if (!last_seg) breg[g - 4] = bload4(r_u, tq16, nx_base + (C - 13) * 16384 + (g - 4) * 4096);
                            else if (g == 4) breg[0] = bload4(r_urgb, tq16, (C - 13) * 4096);
                        }
                    } else if (RDMA && g >= 4 && g < 8) {
                        // chunk C + 3 straight into its ring slot (the slot of chunk C - 1: every wave left it before the barrier at the top of
                        // this chunk), piece g - 4 of wave w = 64 lanes x 16 B at byte (g - 4) * 4096 + 1024 w
                        auto* dst = (__attribute__((address_space(3))) void*)(smem + (NC & 3) * 16384 + (g - 4) * 4096 + wave_s * 1024);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_u, dst, 16, (int)tq16, (NSTEP * 4 + NPG) * 16384 + (g - 4) * 4096, 0, 0);
                    } else if (!RDMA && g >= 4 && g < 4 + NRING)
                        breg[g - 4] = NBR ? bload4(r_up, tq16, NSTEP * 12288 + (g - 4) * 4096) : bload4(r_u, tq16, (NSTEP * 4 + NPG) * 16384 + (g - 4) * 4096);
                    if (S == 0 && PG == 1 && g == 7) bgv = *reinterpret_cast<const f32x4*>(smem + BG_B + (tqk & 15) * 16);
                    if constexpr (WINO_HALO_DMA) {
                        // The next tile's slab S (1296 float4) as LDS-DMA loads: pieces 0-2 in position chunk 1, pieces 3, 4 in chunk 2 -- every
                        // wave has read its last patch rows of slab S in chunk 0, the barrier at the top of chunk 1 is behind -- each straight to
                        // its place (piece i of wave w: 64 lanes x 16 B at element 256 i + 64 w).  No staging registers (12), no ds_write, and no
                        // wait for HBM data in the instruction stream: the loads retire in order in front of the weight requests the ring
                        // writes wait for, and the slab's first reader is five chunks away.  The sixth piece is 16 elements (1280 .. 1295): a
                        // DMA load would write 240 lanes past the slab, so it stays load -> register -> ds_write (chunk 2 -> chunk 3).
                        if ((PG == 1 || PG == 2) && g >= 11 && g < (PG == 1 ? 14 : 13)) {
                            const int i = g - 11 + 3 * (PG - 1);
                            auto* dst = (__attribute__((address_space(3))) void*)(smem + RING_B + S * SLAB_B + i * 4096 + wave_s * 1024);
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_src, dst, 16, (int)hoff[i], (int)(ref_so + S * 64), 0, 0);
                        }
                        if (PG == 2 && g == 13) hreg[0] = bload4(r_src, hoff[5], ref_so + S * 64);
                        if (PG == 3 && g == 8) {
                            int e = tqk + 256 * 5;
                            e = e < NPX * 4 ? e : NPX * 4 - 1;
                            *reinterpret_cast<f32x4*>(smem + RING_B + S * SLAB_B + e * 16) = hreg[0];
                        }
                    } else {
                        if ((PG == 1 || PG == 2) && g >= 8 && g < 11) {        // (before this chunk's own halo requests reuse the registers)
                            int e = tqk + 256 * (g - 8 + 3 * (PG - 1));
                            e = e < NPX * 4 ? e : NPX * 4 - 1;
                            *reinterpret_cast<f32x4*>(smem + RING_B + S * SLAB_B + e * 16) = hreg[g - 8];
                        }
                        if ((PG == 0 || PG == 1) && g >= 11 && g < 14) hreg[g - 11] = bload4(r_src, hoff[g - 11 + 3 * PG], ref_so + S * 64);
                    }
                    // MS, last source: the next tile's RGB halo, requested in step 1 and stored a chunk later (this tile's RGB patch was read
                    // before its first chunk)
                    // (every segment does it: the same pixels again, but no run-time branch in the chunk and no value that lives across one)
                    if (MS && S == 1 && PG == 0 && g == 14 && (MSF || last_seg)) rgb_dma(tqk, nty0, ntx0);
                    if (RES && S == 3 && PG == 0 && g == 16) {
                        // The residual map was last touched a whole launch ago: its lines come from HBM.  Touch this wave's 128 lines (8 rows x
                        // 8 pixels x 256 B) now, four chunks ahead of the epilogue, so that its 16-B loads find them in L2.  As LDS-DMA loads
                        // into a dump area: a load with a destination REGISTER has to keep it until the data is back, and at the register
                        // limit hipcc spilled the two values right behind the loads -- i.e. every wave waited out the HBM latency in the
                        // middle of the K loop for the sake of a prefetch (r05; ~4 k cycles per tile, tools/trace_wino.py: 42.3 k against
                        // conv_hr's 37.9 k).  (A ragged tile touches lines of other pixels or, beyond the map, nothing: the descriptor's range.)
                        const unsigned wo = (unsigned)((8 * (tqk >> 7) + ((tqk >> 4) & 3)) * W + 8 * ((tqk >> 6) & 1)) * 256u + (unsigned)(tqk & 15) * 128u;
                        auto* dump = (__attribute__((address_space(3))) void*)(smem + DUMP_B + (tqk >> 6) * 256);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_res, dump, 4, (int)wo, (int)((unsigned)(ty0 * W + tx0) * 256u), 0, 0);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_res, dump, 4, (int)(wo + (unsigned)W * 1024u), (int)((unsigned)(ty0 * W + tx0) * 256u), 0, 0);
                    }
                    // rolling transform of row TR + the patch rows of step S + 1.  Branch kernels: the transform BEFORE this chunk's patch reads
                    // -- the rows it frees (chunk 0: the old rows 1 and 3) are dead by the time new ones arrive, at most three patch rows live
                    // instead of four; its results pinned where they are computed (with branch chunks in the way -- other basic blocks --
                    // LLVM's IR-level sinking moves the adds next to their first use, into the join block behind the next step's branches,
                    // where nothing overlaps them: 96 VALU instructions per branch chunk in the first PAR build).  Plain kernels: the
                    // same source order makes hipcc keep `acc` and `V` in scratch MEMORY (private_seg_size 1616, ten times slower), and
                    // the pin costs the back half 8 %: they keep reads first, transform second, unpinned.
                    //   The branch kernels do the 32 adds in TWO gaps (16 + 16): a gap with any vector-ALU instruction in it costs ~9-20
                    // cycles of matrix time before the 4 per instruction (tools/ubench/ub_valu_gap.hip; front half 427 -> 423 us).  The
                    // plain kernels keep one float4 per gap: lumped, the residual kernel spills 22 instead of 13 registers (+3 %).
                    if constexpr (PAR) {
                        if (XF && WINO_JIT_ROWS && g >= 8 && g < 16) {   // (beside the halo pieces: LDS reads cost the matrix pipe nothing)
                            const int c = g & 3;
                            if (g < 12) da[c] = patch(SR{}, RA, c);
                            else db[c] = patch(SR{}, RB, c);
                        }
                        if (XF && g == 16) {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                tt[c] = WINO_JIT_ROWS ? (TR == 1 ? da[c] + db[c] : da[c] - db[c])
                                                      : TR == 0 ? d0[c] - d2[c] : (TR == 1 ? d1[c] + d2[c] : (TR == 2 ? d2[c] - d1[c] : d1[c] - d3[c]));
                        }
                        if (XF && g == 17) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                V[4 * TR + c] = c == 0 ? tt[0] - tt[2] : (c == 1 ? tt[1] + tt[2] : (c == 2 ? tt[2] - tt[1] : tt[1] - tt[3]));
                                asm volatile("" : "+v"(V[4 * TR + c]));
                            }
                        }
                        if (!WINO_JIT_ROWS && g >= 28 && g < 36) {                 // patch rows of step S + 1: rows 0, 2 | 1 | 3 | none
                            const int c = (g - 28) & 3;
                            if (PG == 0 && g < 32) d0[c] = patch(SN{}, 0, c);
                            if (PG == 0 && g >= 32) d2[c] = patch(SN{}, 2, c);
                            if (PG == 1 && g < 32) d1[c] = patch(SN{}, 1, c);
                            if (PG == 2 && g < 32) d3[c] = patch(SN{}, 3, c);
                        }
                    } else {
                        if (XF && WINO_JIT_ROWS && g >= 20 && g < 28) {
                            const int c = (g - 20) & 3;
                            if (g < 24) da[c] = patch(SR{}, RA, c);
                            else db[c] = patch(SR{}, RB, c);
                        }
                        if (!WINO_JIT_ROWS && g >= 20 && g < 28) {                 // patch rows of step S + 1: rows 0, 2 | 1 | 3 | none
                            const int c = (g - 20) & 3;
                            if (PG == 0 && g < 24) d0[c] = patch(SN{}, 0, c);
                            if (PG == 0 && g >= 24) d2[c] = patch(SN{}, 2, c);
                            if (PG == 1 && g < 24) d1[c] = patch(SN{}, 1, c);
                            if (PG == 2 && g < 24) d3[c] = patch(SN{}, 3, c);
                        }
                        // first the row combination, then the column combination: WINO_LUMP gaps in all (a gap that holds any vector-ALU
                        // instruction costs ~9 cycles of matrix time before the 4 per instruction: 8 gaps 200 cycles per chunk, 2 gaps 146)
                        constexpr int LPG = 8 / WINO_LUMP;       // float4 sums per gap
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (XF && g == 36 + c / LPG)
                                tt[c] = WINO_JIT_ROWS ? (TR == 1 ? add4(da[c], db[c]) : sub4(da[c], db[c]))
                                                      : TR == 0 ? sub4(d0[c], d2[c]) : (TR == 1 ? add4(d1[c], d2[c]) : (TR == 2 ? sub4(d2[c], d1[c]) : sub4(d1[c], d3[c])));
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (XF && g == 36 + (4 + c) / LPG)
                                V[4 * TR + c] = c == 0 ? sub4(tt[0], tt[2]) : (c == 1 ? add4(tt[1], tt[2]) : (c == 2 ? sub4(tt[2], tt[1]) : sub4(tt[1], tt[3])));
                    }
                    if constexpr (FO) {
                        // the plane's 1x1 fragments of this step: requested in position row 0, parked (N tiles 0-2) / kept (3) in row 1
                        if (PG == 0 && g >= 14 && g < 18) wjl[g - 14] = bload4(r_up, bl, fo_so + S * 12288 + (g - 14) * 1024);
                        if (PG == 1 && g >= 2 && g < 5) *reinterpret_cast<f32x4*>(smem + PV_B + (g - 2) * 4096 + tq16) = wjl[g - 2];
                        if (PG == 1 && g == 5) wj3 = wjl[3];
                    }
                    if constexpr (PAR || FO) {
                        if ((PG == 1 || PG == 2) && (g == 11 || g == 12 || g == 13 || g == 27 || g == 28 || g == 29)) {
                            // (unconditional, like the FMAs below with a zero factor when nothing is folded: a branch per gap costs the
                            //  K loop its straight-line schedule; the rows hold finite partition values then)
                            wjt[(g & 15) - 11] = lds4(PV_B + ((g & 15) - 11) * 4096 + tq16);
                        }
                        if ((PG == 1 || PG == 2) && (g == 15 || g == 31)) {
                            // positions (1,1) (1,2) | (2,1) (2,2) are next: their fragments (read a position ago) take the folded plane
                            {
                                const int pjn = (g + 1) >> 4;
                                const float cs = ((PG == 1) == (pjn == 1)) ? foldc : -foldc;
#if WINO_PK_FOLD
                                const f32x2 c2 = {cs, cs};
#pragma unroll
                                for (int n = 0; n < 3; ++n) bf[pjn & 1][n] = pk_fma4(c2, wjt[n], bf[pjn & 1][n]);
                                bf[pjn & 1][3] = pk_fma4(c2, wj3, bf[pjn & 1][3]);
#else
                                const f32x4 c4 = {cs, cs, cs, cs};
#pragma unroll
                                for (int n = 0; n < 3; ++n) bf[pjn & 1][n] = __builtin_elementwise_fma(c4, wjt[n], bf[pjn & 1][n]);
                                bf[pjn & 1][3] = __builtin_elementwise_fma(c4, wj3, bf[pjn & 1][3]);
#endif
                            }
                        }
                    }
                    if (!RDMA && g >= 52 && g < 52 + NRING) {
                        if (MSF || !(MS && C >= 13 && last_seg && g > 52)) *reinterpret_cast<f32x4*>(smem + (NC & 3) * 16384 + (g - 52) * 4096 + tq16) = breg[g - 52];
                    }
                    if (RDMA && g == 63) {
                        // The ONE wait of a chunk: everything requested BEFORE this chunk has landed -- i.e. the weight chunk C + 2 (requested a
                        // chunk ago, first read behind the barrier at the top of the next chunk) and every halo piece older than this chunk.
                        // vmcnt retires in order, so "at most the requests of THIS chunk outstanding" says exactly that: 4 weight pieces + the
                        // halo pieces (3 | 2 + the sixth piece's register load) + the residual warm-up (2) + the folded plane's fragments (4).
                        // (The epilogue's stores in front of a tile's first chunk are not counted: that wait then covers them too, as the ring
                        //  write's wait did.)
                        // (MS: + the two pieces of the next tile's RGB halo)
                        constexpr int NVM = 4 + (WINO_HALO_DMA ? (PG == 1 ? 3 : (PG == 2 ? 3 : 0)) : ((PG == 0 || PG == 1) ? 3 : 0))
                                          + ((RES && S == 3 && PG == 0) ? 2 : 0) + ((FO && PG == 0) ? 4 : 0) + ((MS && S == 1 && PG == 0) ? 2 : 0);
                        wait_vm<NVM>();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (KIND == 1) {
                static_assert(!NBR, "a branch chunk requests a position chunk (four pieces), written by the position chunk behind it");
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto step = [&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            if constexpr (PAR) {
                chunk(I<5 * S>{}, s_c, I<0>{}, I<1>{});
                chunk(I<5 * S + 1>{}, s_c, I<0>{}, I<0>{});
                chunk(I<5 * S + 2>{}, s_c, I<1>{}, I<0>{});
                chunk(I<5 * S + 3>{}, s_c, I<2>{}, I<0>{});
                chunk(I<5 * S + 4>{}, s_c, I<3>{}, I<0>{});
            } else {
                chunk(I<4 * S>{}, s_c, I<0>{}, I<0>{});
                chunk(I<4 * S + 1>{}, s_c, I<1>{}, I<0>{});
                chunk(I<4 * S + 2>{}, s_c, I<2>{}, I<0>{});
                chunk(I<4 * S + 3>{}, s_c, I<3>{}, I<0>{});
            }
        };
        unsigned long long dbg_a = 0, dbg_b = 0;
        if (a.dbg) dbg_a = __builtin_amdgcn_s_memtime();
        if constexpr (MS) {
            // ---- the frame itself first: 3 (+1 zero) channels are ONE k-step of the MFMA (channel = lane quarter), so its Winograd
            //      contraction is 4 chunks of 16 MFMAs; they start the accumulators (zero C operand; the bias at position (1,1))
            {
                const unsigned rb = PV_B + ((2 * ty) * HP + tx) * 16 + kq * 4;
                float dr[16];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) dr[r * 4 + c] = *reinterpret_cast<const float*>(smem + rb + (r * HP + (c & 1) * 9 + (c >> 1)) * 16);
                float tr[16];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    tr[0 * 4 + c] = dr[0 * 4 + c] - dr[2 * 4 + c];
                    tr[1 * 4 + c] = dr[1 * 4 + c] + dr[2 * 4 + c];
                    tr[2 * 4 + c] = dr[2 * 4 + c] - dr[1 * 4 + c];
                    tr[3 * 4 + c] = dr[1 * 4 + c] - dr[3 * 4 + c];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    Vr[i * 4 + 0] = tr[i * 4 + 0] - tr[i * 4 + 2];
                    Vr[i * 4 + 1] = tr[i * 4 + 1] + tr[i * 4 + 2];
                    Vr[i * 4 + 2] = tr[i * 4 + 2] - tr[i * 4 + 1];
                    Vr[i * 4 + 3] = tr[i * 4 + 1] - tr[i * 4 + 3];
                }
            }
            const f32x4 bgv = *reinterpret_cast<const f32x4*>(smem + BG_B + (tq & 15) * 16);
            auto rgb_chunk = [&](auto pg_c) {
                constexpr int PG = decltype(pg_c)::value;         // chunk PG of the tile, ring slot PG
                lds_bar();
                f32x4 br[4];
#pragma unroll
                for (int pj = 0; pj < 4; ++pj) br[pj] = lds4(PG * 16384 + (pj * 64 + (tq & 63)) * 16);
                // three ahead: RGB chunk 3 (one piece), then the first source's chunks 0..2
                constexpr bool RD = WINO_RING_DMA && WINO_HALO_DMA;
                if constexpr (RD) {
                    // (the slot of the RGB chunk in front: every wave took its fragments out of it before the barrier above)
                    auto* dst = (__attribute__((address_space(3))) void*)(smem + ((PG + 3) & 3) * 16384 + wave_s * 1024);
                    if (PG == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_urgb, dst, 16, (int)tq16, 3 * 4096, 0, 0);
                    else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            auto* di = (__attribute__((address_space(3))) void*)(smem + ((PG + 3) & 3) * 16384 + i * 4096 + wave_s * 1024);
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_u, di, 16, (int)tq16, (int)(a.u_off[0] + (PG - 1) * 16384 + i * 4096), 0, 0);
                        }
                    }
                } else if (PG == 0) breg[0] = bload4(r_urgb, tq16, 3 * 4096);
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) breg[i] = bload4(r_u, tq16, a.u_off[0] + (PG - 1) * 16384 + i * 4096);
                }
#pragma unroll
                for (int pj = 0; pj < 4; ++pj)
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const int p = PG * 4 + pj;
                        const f32x4 c0 = p == 5 ? f32x4{bgv[n], bgv[n], bgv[n], bgv[n]} : f32x4{0.f, 0.f, 0.f, 0.f};
                        acc[p][n] = mfma16(Vr[p], br[pj][n], c0);
                    }
                if (PG == 3) {                                     // the first fragments of the first source's first chunk
#pragma unroll
                    for (int n = 0; n < 4; ++n) bf[0][n] = lds4(bl + n * 1024);
                }
                if constexpr (RD) wait_vm<(PG == 0 ? 1 : 4)>();       // what the chunks in front requested has landed (this chunk's own may fly)
                else {
#pragma unroll
                    for (int i = 0; i < (PG == 0 ? 1 : 4); ++i) *reinterpret_cast<f32x4*>(smem + ((PG + 3) & 3) * 16384 + i * 4096 + tq16) = breg[i];
                }
            };
            rgb_chunk(I<0>{});
            rgb_chunk(I<1>{});
            rgb_chunk(I<2>{});
            rgb_chunk(I<3>{});
            // ---- then one segment of 16 chunks per 64-channel source
            const int nw = a.nsrc;
            for (int ks = 0; ks < nw; ++ks) {
                last_seg = ks + 1 >= nw;
                u_so = a.u_off[ks];
                if (MSF) r_nx = rsrc_of(last_seg ? a.Urgb : a.ubase, last_seg ? 4u * 4096u : OOBW);
                nx_base = (MSF && last_seg) ? 0u : a.u_off[last_seg ? 0 : ks + 1];
                nx_cs = (MSF && last_seg) ? 4096u : 16384u;
                nx_ps = (MSF && last_seg) ? 0u : 4096u;
                // the slabs this segment refills belong to the next source of this tile, or to the first source of the next tile
                r_src = rsrc_of(reinterpret_cast<const char*>(a.srcs[last_seg ? 0 : ks + 1]) - ((long)W + 1) * 256, OOBW);
                ref_so = last_seg ? nso : (unsigned)(ty0 * W + tx0) * 256u;
                // (the thread id laundered per segment: the eighteen per-lane constants behind the six offsets are recomputed here instead
                //  of living -- seven of them in scratch, each reload an s_waitcnt vmcnt(0) -- across the whole tile)
                int tqs = tq;
                if (WINO_MS_LAUNDER >= 1) asm volatile("" : "+v"(tqs));
                if (WINO_MS_LAUNDER >= 2) {
                    tqk = tqs;
                    tq16 = (unsigned)tqs * 16u;
                }
                halo_offsets(tqs, last_seg ? nty0 : ty0, last_seg ? ntx0 : tx0);
                step(I<0>{});
                step(I<1>{});
                step(I<2>{});
                step(I<3>{});
            }
        } else {
            step(I<0>{});
            step(I<1>{});
            step(I<2>{});
            step(I<3>{});
        }
        if (a.dbg) {
            dbg_b = __builtin_amdgcn_s_memtime();
            dbg_k += dbg_b - dbg_a;
        }

        // ---- epilogue: Y = A^T M A per (tile, channel) in the accumulator layout (+ bias (* gamma), activation), then one N tile at a
        //      time through LDS -- the wave's 4 x 16 pixel strip x 16 channels = 4 KiB of ring slot 3, free now: the next tile's chunks
        //      0..2 sit in slots 0..2 -- so that residual loads and stores are 16 B per lane (64-B channel runs per pixel)
        const unsigned so = (unsigned)(ty0 * W + tx0) * 256u;
        const bool partial = ty0 + 16 > H || tx0 + 16 > W;
        lds_bar();                       // every wave has read its last fragments out of slot 3
        // (next = the block's quadrant unit: every wave fetches quadrant qquad's values)
        const int tqp = (!has_next && qtile >= 0) ? ((qquad << 6) | (tq & 63)) : tq;
        pv_request(tqp, nty0, ntx0);
        fo_request(nty0, ntx0, (!has_next && qtile >= 0) ? qquad : -1);
        auto epilogue = [&](auto partial_c) {
            constexpr bool PARTIAL = decltype(partial_c)::value;
            // (the thread id laundered once more: everything the epilogue derives from it is computed HERE -- hoisted to the top of the
            //  tile it lives through the K loop at the register limit, i.e. in scratch, and each reload is an s_waitcnt vmcnt(0))
            int te = tq;
            asm volatile("" : "+v"(te));
            const int lq = te & 63, wq = te >> 6, kqq = lq >> 4, mq = lq & 15;
            char* tr = smem + 3 * 16384 + wq * 4096;
            // write side: accumulator register r of this lane is tile (row kq, column r) of the wave's 4x4 tiles; value (q = 2 a + b, r)
            // is pixel (row 2 kq + a, column 2 r + b) of its 8x8 block, channel m
            const unsigned wbase = (unsigned)((2 * kqq) * 8 * 64 + mq * 4);
            // read side: float4 j of this lane = block pixel lp = 16 j + (lane >> 2) (row lp >> 3, column lp & 7), channels 4 (lane & 3) .. + 3
            const unsigned rbase = (unsigned)((lq >> 2) * 64 + (lq & 3) * 16);
            unsigned go[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 8 * (wq >> 1) + 2 * j + (lq >> 5), col = 8 * (wq & 1) + ((lq >> 2) & 7);
                go[j] = (unsigned)(row * W + col) * 256u + (unsigned)(lq & 3) * 16u;
                if (PARTIAL) go[j] = (ty0 + row < H && tx0 + col < W) ? go[j] : OOBW;
            }
            f32x4 rs[4];
            if (RES) {
#pragma unroll
                for (int j = 0; j < 4; ++j) rs[j] = bload4(r_res, go[j], so);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                f32x4 w0[4], w1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    w0[i] = acc[i * 4 + 0][n] + acc[i * 4 + 1][n] + acc[i * 4 + 2][n];
                    w1[i] = acc[i * 4 + 1][n] - acc[i * 4 + 2][n] - acc[i * 4 + 3][n];
                }
                f32x4 y[4];
                y[0] = w0[0] + w0[1] + w0[2];
                y[1] = w1[0] + w1[1] + w1[2];
                y[2] = w0[1] - w0[2] - w0[3];
                y[3] = w1[1] - w1[2] - w1[3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // (the bias came in through the accumulator of position (1,1))
#pragma unroll
                    for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(tr + wbase + ((q >> 1) * 8 + 2 * r + (q & 1)) * 64) = y[q][r];
                }
                f32x4 o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = *reinterpret_cast<const f32x4*>(tr + rbase + j * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // none: max(v, 1 v) | relu: max(v, 0 v) | leaky-relu: max(v, 0.1 v).  The residual bodies take no activation (the
                    // reference adds the residual to a bare conv, sr_backbone_utils.py:313,329; launch_conv3x3_wino refuses anything else):
                    // 128 vector-ALU instructions per tile less where the matrix pipe stands idle
                    if constexpr (!RES) o[j] = __builtin_elementwise_max(o[j], act_lo * o[j]);
                    if (RES) {
                        o[j] += rs[j];
                        if (n < 3) rs[j] = bload4(r_res, go[j], so + (n + 1) * 64);       // requested one N tile ahead
                    }
                    bstore4(r_out, go[j], so + n * 64, o[j]);      // (the N tile's 64 B go into the scalar offset: go[j] may be the OOB marker)
                }
            }
        };
        if (partial) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        pv_finish(tqp);
        fo_finish();
        ++dbg_n;
        if (a.dbg) dbg_e += __builtin_amdgcn_s_memtime() - dbg_b;
        if (!has_next) break;
        tile = ntile;
        ty0 = nty0;
        tx0 = ntx0;
        if constexpr (SEAM) {
            // V rows 0-2 of the next tile's first step (its slab 0 has been in place since this tile's step 1): twelve patch reads and 24
            // float4 sums with no MFMA beside them -- the vector ALU is the matrix pipe, so only the LDS latency (a few hundred cycles per
            // tile) is new -- in exchange for 48 registers that no longer live across the epilogue
            f32x4 e0[4], e1[4], e2[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                e0[c] = patch(I<0>{}, 0, c);
                e1[c] = patch(I<0>{}, 1, c);
                e2[c] = patch(I<0>{}, 2, c);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32x4 tr[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if constexpr (PAR) tr[c] = i == 0 ? e0[c] - e2[c] : (i == 1 ? e1[c] + e2[c] : e2[c] - e1[c]);
                    else tr[c] = i == 0 ? sub4(e0[c], e2[c]) : (i == 1 ? add4(e1[c], e2[c]) : sub4(e2[c], e1[c]));
                }
                if constexpr (PAR) {
                    V[4 * i + 0] = tr[0] - tr[2];
                    V[4 * i + 1] = tr[1] + tr[2];
                    V[4 * i + 2] = tr[2] - tr[1];
                    V[4 * i + 3] = tr[1] - tr[3];
                } else {
                    V[4 * i + 0] = sub4(tr[0], tr[2]);
                    V[4 * i + 1] = add4(tr[1], tr[2]);
                    V[4 * i + 2] = sub4(tr[2], tr[1]);
                    V[4 * i + 3] = sub4(tr[1], tr[3]);
                }
            }
        }
    }
    // (LDS-DMA loads of "the next tile's" first weight chunks may still be in flight behind the block's last tile: they must have landed
    //  before the block ends and its LDS goes to the next one)
    if constexpr (WINO_RING_DMA && WINO_HALO_DMA && !PAR) wait_vm<0>();
    // ---- quadrant unit (see the strip assignment): 8x8 pixels of tile qtile, wave w = output channels 16 w .. + 15.  Same arithmetic
    // in the same order as a whole tile -- per accumulator: branches, then the position's 4 k-steps, step by step; bias through the C
    // operand of position (1,1) -- so a pixel's value does not depend on which form computed it (bit for bit; tested).  Straight-line code:
    // the patch comes out of the slabs (the last whole tile's K loop fetched this tile's halo as its "next tile"), the B fragments straight
    // from L2 into registers, a step ahead (a wave reads only its own N tile's).
    if constexpr (!MS) {
        const int qy0 = (qtile / tiles_x) * 16 + 8 * (qquad >> 1), qx0 = (qtile % tiles_x) * 16 + 8 * (qquad & 1);
        if (qtile >= 0 && qy0 < H && qx0 < W) {
            // (its tile's halo is in the slabs already: the last whole tile's K loop fetched it as "the next tile")
            if (a.dbg) dbg_q0 = __builtin_amdgcn_s_memtime();
            const int tyq = m >> 2, txq = m & 3;
            const unsigned wq16 = (unsigned)lane * 16u + (unsigned)wave * 1024u;
            const unsigned qb0 = RING_B + ((2 * (4 * (qquad >> 1) + tyq)) * HP + 4 * (qquad & 1) + txq) * 64 + kq * 16, qb1 = qb0 + 2 * SLAB_B;
            // B fragments of step 0 first (the first MFMAs wait for these only), then the residual values
            f32x4 Bq[2][16], Bp[2][3];
#pragma unroll
            for (int p = 0; p < 16; ++p) Bq[0][p] = bload4(r_u, wq16, (unsigned)((p >> 2) * 16384 + (p & 3) * 4096));
            if constexpr (PAR) {
#pragma unroll
                for (int j = 0; j < 3; ++j) Bp[0][j] = bload4(r_up, wq16, (unsigned)(j * 4096));
            }
            // FO: the unit's quadrant is all zero or one constant plane (fo_request looked at quadrant qquad for every wave)
            f32x4 Bf[2];
            const int foq = FO ? __builtin_amdgcn_readfirstlane(fo_next) : -1;
            const unsigned foq_so = (unsigned)(foq > 0 ? foq : 0) * 4096u;
            if constexpr (FO) Bf[0] = bload4(r_up, wq16, foq_so);
            // output offsets of the lane's 16 values: value (q = 2 a + b, r) is pixel (2 kq + a, 2 r + b) of the quadrant, channel 16 w + m
            const unsigned qo = (unsigned)((qy0 + 2 * kq) * W + qx0) * 256u + (unsigned)(wave * 16 + m) * 4u;
            float resq[16];
            if constexpr (RES) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool inq = qy0 + 2 * kq + (q >> 1) < H && qx0 + 2 * r + (q & 1) < W;
                        resq[q * 4 + r] = bload1(r_res, inq ? qo : OOBW, (unsigned)((q >> 1) * W + 2 * r + (q & 1)) * 256u);
                    }
            }
            const float bgq = *reinterpret_cast<const float*>(smem + BG_B + (m * 4 + wave) * 4);
            if constexpr (PAR) {
#pragma unroll
                for (int n = 0; n < 1; ++n) acc[0][0] = acc[3][0] = acc[12][0] = acc[15][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            lds_bar();       // (the slabs are complete and every wave is out of the last epilogue; the requests above stay in flight)
            if (a.dbg) dbg_q1 = __builtin_amdgcn_s_memtime();
            // the partition values and the branch decision of this quadrant came through the tile pipeline (pv_request / pv_finish with
            // every wave on quadrant qquad)
            int needq = (PAR && a.par_flags) ? __builtin_amdgcn_readfirstlane(need_next) : 7;
            // (a folded plane as in the tiles: the same FMAs on the same fragments, so the values stay the tile form's bit for bit)
            const int foldq = PAR ? __builtin_amdgcn_readfirstlane(fold_next) : -1;
            const float foldcq = PAR ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, foldc_next)))
                                     : (FO ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, foc_next))) : 0.f);
            if (foldq >= 0) needq = 0;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                if (s4 < 3) {
#pragma unroll
                    for (int p = 0; p < 16; ++p) Bq[(s4 + 1) & 1][p] = bload4(r_u, wq16, (unsigned)(((s4 + 1) * 4 + (p >> 2)) * 16384 + (p & 3) * 4096));
                    if constexpr (PAR) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) Bp[(s4 + 1) & 1][j] = bload4(r_up, wq16, (unsigned)((s4 + 1) * 12288 + j * 4096));
                    }
                    if constexpr (FO) Bf[(s4 + 1) & 1] = bload4(r_up, wq16, (unsigned)((s4 + 1) * 12288) + foq_so);
                }
                f32x4 dq[4][4], tq4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        dq[r][c] = lds4((s4 < 2 ? qb0 : qb1) + (s4 & 1) * SLAB_B + (r * HP + (c & 1) * 9 + (c >> 1)) * 64);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) tq4[c] = i == 0 ? dq[0][c] - dq[2][c] : (i == 1 ? dq[1][c] + dq[2][c] : (i == 2 ? dq[2][c] - dq[1][c] : dq[1][c] - dq[3][c]));
                    V[4 * i + 0] = tq4[0] - tq4[2];
                    V[4 * i + 1] = tq4[1] + tq4[2];
                    V[4 * i + 2] = tq4[2] - tq4[1];
                    V[4 * i + 3] = tq4[1] - tq4[3];
                }
                // the whole transform first, then the MFMAs back to back: interleaved (what the scheduler does by itself) every MFMA gap
                // holds 2-3 vector-ALU instructions and costs ~25 cycles of matrix time (tools/ubench/ub_valu_gap.hip)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(V[i]));
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (PAR) {
                    auto qbranch = [&](auto j_c) {
                        constexpr int J = decltype(j_c)::value;
                        const f32x4 pv = lds4(PV_B + J * 4096 + (unsigned)((qquad << 6) | lane) * 16);
                        f32x4 ax[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            ax[q] = (q == 0 ? dq[1][1] : (q == 1 ? dq[1][2] : (q == 2 ? dq[2][1] : dq[2][2]))) * pv[q];
                            asm volatile("" : "+v"(ax[q]));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        // (k outermost: four independent accumulators in a row; each one still sees its k-steps in order)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                f32x4& ac = q == 0 ? acc[0][0] : (q == 1 ? acc[3][0] : (q == 2 ? acc[12][0] : acc[15][0]));
                                ac = mfma16(ax[q][k], Bp[s4 & 1][J][k], ac);
                            }
                    };
                    if (needq & 1) qbranch(I<0>{});
                    if (needq & 2) qbranch(I<1>{});
                    if (needq & 4) qbranch(I<2>{});
                    if (foldq >= 0) {
                        const f32x4 wq = foldq == 0 ? Bp[s4 & 1][0] : (foldq == 1 ? Bp[s4 & 1][1] : Bp[s4 & 1][2]);
                        const f32x4 cp = {foldcq, foldcq, foldcq, foldcq}, cm = {-foldcq, -foldcq, -foldcq, -foldcq};
                        Bq[s4 & 1][5] = __builtin_elementwise_fma(cp, wq, Bq[s4 & 1][5]);
                        Bq[s4 & 1][6] = __builtin_elementwise_fma(cm, wq, Bq[s4 & 1][6]);
                        Bq[s4 & 1][9] = __builtin_elementwise_fma(cm, wq, Bq[s4 & 1][9]);
                        Bq[s4 & 1][10] = __builtin_elementwise_fma(cp, wq, Bq[s4 & 1][10]);
                    }
                }
                if constexpr (FO) {
                    const f32x4 wq = Bf[s4 & 1];
                    const f32x4 cp = {foldcq, foldcq, foldcq, foldcq}, cm = {-foldcq, -foldcq, -foldcq, -foldcq};
                    Bq[s4 & 1][5] = __builtin_elementwise_fma(cp, wq, Bq[s4 & 1][5]);
                    Bq[s4 & 1][6] = __builtin_elementwise_fma(cm, wq, Bq[s4 & 1][6]);
                    Bq[s4 & 1][9] = __builtin_elementwise_fma(cm, wq, Bq[s4 & 1][9]);
                    Bq[s4 & 1][10] = __builtin_elementwise_fma(cp, wq, Bq[s4 & 1][10]);
                }
#pragma unroll
                for (int pr = 0; pr < 4; ++pr)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int pc = 0; pc < 4; ++pc) {
                            const int p = pr * 4 + pc;
                            const bool fresh = s4 == 0 && k == 0 && !(PAR && (p == 0 || p == 3 || p == 12 || p == 15));
                            const f32x4 c0 = p == 5 ? f32x4{bgq, bgq, bgq, bgq} : f32x4{0.f, 0.f, 0.f, 0.f};
                            acc[p][0] = mfma16(V[p][k], Bq[s4 & 1][p][k], fresh ? c0 : acc[p][0]);
                        }
            }
            if (a.dbg) dbg_q2 = __builtin_amdgcn_s_memtime();
            // output transform, activation, residual, 4-B stores (a wave instruction writes 64 B of four pixels)
            f32x4 w0[4], w1[4], yq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w0[i] = acc[i * 4 + 0][0] + acc[i * 4 + 1][0] + acc[i * 4 + 2][0];
                w1[i] = acc[i * 4 + 1][0] - acc[i * 4 + 2][0] - acc[i * 4 + 3][0];
            }
            yq[0] = w0[0] + w0[1] + w0[2];
            yq[1] = w1[0] + w1[1] + w1[2];
            yq[2] = w0[1] - w0[2] - w0[3];
            yq[3] = w1[1] - w1[2] - w1[3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                yq[q] = __builtin_elementwise_max(yq[q], act_lo * yq[q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool inq = qy0 + 2 * kq + (q >> 1) < H && qx0 + 2 * r + (q & 1) < W;
                    float v = yq[q][r];
                    if (RES) v += resq[q * 4 + r];
                    bstore1(r_out, inq ? qo : OOBW, (unsigned)((q >> 1) * W + 2 * r + (q & 1)) * 256u, v);
                }
            }
        }
    }
    if (a.dbg && t == 0) {
        unsigned long long* d = a.dbg + (size_t)blockIdx.x * 16;
        d[0] = dbg_t0;
        d[1] = dbg_k;
        d[2] = dbg_e;
        d[3] = __builtin_amdgcn_s_memtime();
        d[4] = dbg_q0;         // quadrant unit: start, first step, output transform (0: the block had none)
        d[5] = dbg_q1;
        d[6] = dbg_q2;
        d[7] = dbg_n;
        d[13] = dbg_r0;
        d[14] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---- small frames: every tile as four quadrant units -------------------------------------------------------------------------------
// Below 512 tiles the persistent kernel above cannot fill the chip (128x128: 64 tiles for 256 CUs) and the direct kernel runs one
// 4x16-pixel tile per CU (a 64 -> 64 conv = 9.6 us of MFMA in a 16 us launch).  Here block u works on quadrant u & 3 of tile u >> 2:
// 8x8 pixels, the four waves splitting the output channels -- the arithmetic of the big kernel's tail units (bit for bit the same
// values as a whole tile), as a kernel of its own: the 10x10x64 halo goes to LDS first, the B fragments come straight from L2 a step
// ahead, 2.25x fewer matrix FLOPs than the direct form and four times as many blocks.
template <bool PAR, bool RES, bool MS, bool FO = false>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_kernel(const WinoArgs a) {
    wino_tile_body<PAR, RES, MS, FO>(a);
}

#ifndef WINO_MS_TU
// A front half behind the device-side gate as ONE launch: the frame's partition word (ConvArgs::par_any, bit 3 = every 8x8 quadrant of the
// frame is all zero or carries one constant plane) picks the fold-only body or the branch body -- block-uniform, one scalar load.  Round 5
// launched both kernels and let one return (112 launches of 5.2 us per 720p clip step = 0.7 %); the bodies are the standalone kernels'
// (same registers, same LDS; only the taken one ever enters the instruction cache).
template <bool RES>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_gated_kernel(const WinoArgs a) {
    WinoArgs b = a;
    b.gate = nullptr;
    if (__builtin_nontemporal_load(a.gate) & 8) {
        b.par_flags = nullptr;
        wino_tile_body<false, RES, false, true>(b);
    } else {
        wino_tile_body<true, RES, false, false>(b);
    }
}
#endif

#ifndef WINO_MS_TU
// FO ("fold only", as in the tile kernels): the frame passed the gate -- every 8x8 quadrant is all zero or carries one constant
// partition plane -- so the unit reads its plane and factor from its first pixel and folds it into the B fragments: no per-lane
// partition values, no branch code, one 1x1 fragment per step instead of three.  Same values as the PAR body bit for bit (the same
// FMAs on the same fragments; accumulators started from an inline zero either way).
template <bool PAR, bool RES, bool FO>
__device__ __forceinline__ void wino_quad_body(const WinoArgs& a) {
    static_assert(!(PAR && FO), "the fold-only body has no branch code");
    constexpr int QSTR = 272;                                // bytes per halo pixel in LDS (64 channels + 16: patch reads spread over banks)
    __shared__ __attribute__((aligned(16))) char smem[100 * QSTR];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + 15) >> 4;
    const int qtile = blockIdx.x >> 2, qquad = blockIdx.x & 3;
    const int qy0 = (qtile / tiles_x) * 16 + 8 * (qquad >> 1), qx0 = (qtile % tiles_x) * 16 + 8 * (qquad & 1);
    if (qy0 >= H || qx0 >= W) return;
    const unsigned map_bytes = (unsigned)H * (unsigned)W * 256u;
    const __amdgpu_buffer_rsrc_t r_u = rsrc_of(a.U, 16u * 16384u);
    const __amdgpu_buffer_rsrc_t r_up = rsrc_of((PAR || FO) ? a.Upar : a.U, 4u * 12288u);
    const __amdgpu_buffer_rsrc_t r_out = rsrc_of(a.out, map_bytes);
    const __amdgpu_buffer_rsrc_t r_res = rsrc_of(RES ? a.residual : a.src, RES ? map_bytes : 0u);
    const float act_lo = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.1f);
    const int tyq = m >> 2, txq = m & 3;
    // Every request of the unit's prologue goes out before the first wait (a launch of these is one latency chain): halo, the lane's
    // partition values, then (below) the B fragments of the first steps; the halo's LDS writes wait for the halo only (in-order vmcnt)
    float praw[3][4];
    f32x4 hv[7];
    {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int e = t + 256 * i, pe = e >> 4, ry = pe / 10, rx = pe - ry * 10, gy = qy0 - 1 + ry, gx = qx0 - 1 + rx;
            hv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e < 1600 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                hv[i] = *reinterpret_cast<const f32x4*>(a.src + ((long)gy * W + gx) * 64 + (e & 15) * 4);
        }
        if constexpr (PAR) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int gy = min(qy0 + 2 * tyq + (q >> 1), H - 1), gx = min(qx0 + 2 * txq + (q & 1), W - 1);      // (clamped: see pv_request)
                    praw[j][q] = a.par[(long)j * a.par_plane + (long)gy * W + gx];
                }
        }
        if constexpr (FO) {
#pragma unroll
            for (int j = 0; j < 3; ++j) praw[j][0] = a.par[(long)j * a.par_plane + (long)qy0 * W + qx0];      // (block-uniform)
        }
    }
    const unsigned wq16 = (unsigned)lane * 16u + (unsigned)wave * 1024u;
    // B fragments of step s4 live in set s4 % QB, requested QA steps ahead (QA = 2: the first two steps' fragments are requested before
    // the halo has landed)
    constexpr int QA = WINO_QUAD_AHEAD, QB = QA + 1;
    f32x4 Bq[QB][16], Bp[QB][FO ? 1 : 3];
#pragma unroll
    for (int s0 = 0; s0 < QA; ++s0) {
#pragma unroll
        for (int p = 0; p < 16; ++p) Bq[s0][p] = bload4(r_u, wq16, (unsigned)((s0 * 4 + (p >> 2)) * 16384 + (p & 3) * 4096));
        if constexpr (PAR) {
#pragma unroll
            for (int j = 0; j < 3; ++j) Bp[s0][j] = bload4(r_up, wq16, (unsigned)(s0 * 12288 + j * 4096));
        }
    }
    int needq = 7, foldq = -1;
    float foldcq = 0.f;
    if constexpr (FO) {
        // (the 1x1 fragments wait for the plane index: one dependent request, behind everything else of the prologue)
        needq = 0;
#pragma unroll
        for (int j = 2; j >= 0; --j) {
            const float uv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, praw[j][0])));
            if (uv != 0.f) {
                foldq = j;
                foldcq = 0.25f * uv;
            }
        }
#pragma unroll
        for (int s0 = 0; s0 < QA; ++s0) Bp[s0][0] = bload4(r_up, wq16, (unsigned)(s0 * 12288 + (foldq < 0 ? 0 : foldq) * 4096));
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int e = t + 256 * i;
        if (e < 1600) *reinterpret_cast<f32x4*>(smem + (e >> 4) * QSTR + (e & 15) * 16) = hv[i];
    }
    // partition values of the lane's tile, signed as the output transform wants them (positions (0,3), (3,0) negated), and the branches
    // the unit needs at all (a plane that is zero on all 64 pixels adds exact zeros)
    float pq[3][4];
    if constexpr (PAR) {
        int nz_any = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            bool nz = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = praw[j][q];
                nz = nz || v != 0.f;
                pq[j][q] = (q == 1 || q == 2) ? -v : v;
            }
            if (__builtin_amdgcn_ballot_w64(nz) != 0) nz_any |= 1 << j;
        }
        if (a.par_flags) needq = __builtin_amdgcn_readfirstlane(nz_any);
        // exactly one plane live on the unit and constant there: folded into the B fragments of positions (1,1) (1,2) (2,1) (2,2), as in
        // the tile kernel (pv_finish there)
        if (WINO_FOLD) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float uv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pq[j][0])));
                const bool same = pq[j][0] == uv && -pq[j][1] == uv && -pq[j][2] == uv && pq[j][3] == uv;
                if (nz_any == (1 << j) && __builtin_amdgcn_ballot_w64(!same) == 0) {
                    foldq = j;
                    foldcq = 0.25f * uv;
                }
            }
            if (foldq >= 0) needq = 0;
        }
    }
    const unsigned qo = (unsigned)((qy0 + 2 * kq) * W + qx0) * 256u + (unsigned)(wave * 16 + m) * 4u;
    float resq[16];
    if constexpr (RES) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool inq = qy0 + 2 * kq + (q >> 1) < H && qx0 + 2 * r + (q & 1) < W;
                resq[q * 4 + r] = bload1(r_res, inq ? qo : OOBW, (unsigned)((q >> 1) * W + 2 * r + (q & 1)) * 256u);
            }
    }
    const int co = wave * 16 + m;
    const float bgq = (a.bias ? a.bias[co] : 0.f) * (a.gamma ? a.gamma[co] : 1.f);
    f32x4 acc[16], V[16];
    if constexpr (PAR) acc[0] = acc[3] = acc[12] = acc[15] = f32x4{0.f, 0.f, 0.f, 0.f};
    lds_bar();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        if (s4 + QA < 4) {
#pragma unroll
            for (int p = 0; p < 16; ++p) Bq[(s4 + QA) % QB][p] = bload4(r_u, wq16, (unsigned)(((s4 + QA) * 4 + (p >> 2)) * 16384 + (p & 3) * 4096));
            if constexpr (PAR) {
#pragma unroll
                for (int j = 0; j < 3; ++j) Bp[(s4 + QA) % QB][j] = bload4(r_up, wq16, (unsigned)((s4 + QA) * 12288 + j * 4096));
            }
            if constexpr (FO) Bp[(s4 + QA) % QB][0] = bload4(r_up, wq16, (unsigned)((s4 + QA) * 12288 + (foldq < 0 ? 0 : foldq) * 4096));
        }
        f32x4 dq[4][4], tq4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                dq[r][c] = *reinterpret_cast<const f32x4*>(smem + ((2 * tyq + r) * 10 + 2 * txq + c) * QSTR + (16 * s4 + 4 * kq) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int c = 0; c < 4; ++c) tq4[c] = i == 0 ? dq[0][c] - dq[2][c] : (i == 1 ? dq[1][c] + dq[2][c] : (i == 2 ? dq[2][c] - dq[1][c] : dq[1][c] - dq[3][c]));
            V[4 * i + 0] = tq4[0] - tq4[2];
            V[4 * i + 1] = tq4[1] + tq4[2];
            V[4 * i + 2] = tq4[2] - tq4[1];
            V[4 * i + 3] = tq4[1] - tq4[3];
        }
        // (the whole transform first, then the MFMAs back to back: see the tail units above)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(V[i]));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PAR) {
            auto qbranch = [&](auto j_c) {
                constexpr int J = decltype(j_c)::value;
                f32x4 ax[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ax[q] = (q == 0 ? dq[1][1] : (q == 1 ? dq[1][2] : (q == 2 ? dq[2][1] : dq[2][2]))) * pq[J][q];
                    asm volatile("" : "+v"(ax[q]));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4& ac = q == 0 ? acc[0] : (q == 1 ? acc[3] : (q == 2 ? acc[12] : acc[15]));
                        ac = mfma16(ax[q][k], Bp[s4 % QB][J][k], ac);
                    }
            };
            if (needq & 1) qbranch(I<0>{});
            if (needq & 2) qbranch(I<1>{});
            if (needq & 4) qbranch(I<2>{});
            if (foldq >= 0) {
                const f32x4 wq = foldq == 0 ? Bp[s4 % QB][0] : (foldq == 1 ? Bp[s4 % QB][1] : Bp[s4 % QB][2]);
                const f32x4 cp = {foldcq, foldcq, foldcq, foldcq}, cm = {-foldcq, -foldcq, -foldcq, -foldcq};
                Bq[s4 % QB][5] = __builtin_elementwise_fma(cp, wq, Bq[s4 % QB][5]);
                Bq[s4 % QB][6] = __builtin_elementwise_fma(cm, wq, Bq[s4 % QB][6]);
                Bq[s4 % QB][9] = __builtin_elementwise_fma(cm, wq, Bq[s4 % QB][9]);
                Bq[s4 % QB][10] = __builtin_elementwise_fma(cp, wq, Bq[s4 % QB][10]);
            }
        }
        if constexpr (FO) {
            if (foldq >= 0) {
                const f32x4 wq = Bp[s4 % QB][0];
                const f32x4 cp = {foldcq, foldcq, foldcq, foldcq}, cm = {-foldcq, -foldcq, -foldcq, -foldcq};
                Bq[s4 % QB][5] = __builtin_elementwise_fma(cp, wq, Bq[s4 % QB][5]);
                Bq[s4 % QB][6] = __builtin_elementwise_fma(cm, wq, Bq[s4 % QB][6]);
                Bq[s4 % QB][9] = __builtin_elementwise_fma(cm, wq, Bq[s4 % QB][9]);
                Bq[s4 % QB][10] = __builtin_elementwise_fma(cp, wq, Bq[s4 % QB][10]);
            }
        }
#pragma unroll
        for (int pr = 0; pr < 4; ++pr)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int pc = 0; pc < 4; ++pc) {
                    const int p = pr * 4 + pc;
                    const bool fresh = s4 == 0 && k == 0 && !(PAR && (p == 0 || p == 3 || p == 12 || p == 15));
                    const f32x4 c0 = p == 5 ? f32x4{bgq, bgq, bgq, bgq} : f32x4{0.f, 0.f, 0.f, 0.f};
                    acc[p] = mfma16(V[p][k], Bq[s4 % QB][p][k], fresh ? c0 : acc[p]);
                }
    }
    f32x4 w0[4], w1[4], yq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w0[i] = acc[i * 4 + 0] + acc[i * 4 + 1] + acc[i * 4 + 2];
        w1[i] = acc[i * 4 + 1] - acc[i * 4 + 2] - acc[i * 4 + 3];
    }
    yq[0] = w0[0] + w0[1] + w0[2];
    yq[1] = w1[0] + w1[1] + w1[2];
    yq[2] = w0[1] - w0[2] - w0[3];
    yq[3] = w1[1] - w1[2] - w1[3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        yq[q] = __builtin_elementwise_max(yq[q], act_lo * yq[q]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool inq = qy0 + 2 * kq + (q >> 1) < H && qx0 + 2 * r + (q & 1) < W;
            float v = yq[q][r];
            if (RES) v += resq[q * 4 + r];
            bstore1(r_out, inq ? qo : OOBW, (unsigned)((q >> 1) * W + 2 * r + (q & 1)) * 256u, v);
        }
    }
}

template <bool PAR, bool RES>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_quad_kernel(const WinoArgs a) {
    wino_quad_body<PAR, RES, false>(a);
}
// A small frame's front half behind the device-side gate as ONE launch, like conv3x3_wino_gated_kernel: bit 3 of the frame's partition
// word picks the fold-only body, the branch body otherwise (block-uniform: a scalar load)
template <bool RES>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_quad_gated_kernel(const WinoArgs a) {
    if (__builtin_nontemporal_load(a.gate) & 8) wino_quad_body<false, RES, true>(a);
    else wino_quad_body<true, RES, false>(a);
}
#endif      // !WINO_MS_TU
#ifdef WINO_MS_TU

// The input conv over the virtual concat [frame, wide sources ...] as quadrant units (the multi-source kernel's arithmetic, block u =
// quadrant u & 3 of tile u >> 2, wave w = output channels 16 w .. + 15): the frame's RGB0 halo and the first source's halo go to LDS,
// the frame's one k-step starts the accumulators (bias at position (1,1)), then one 4-step pass per 64-channel source; the NEXT
// source's halo is fetched into registers during a pass and lands in the other LDS buffer behind it.
__global__ __launch_bounds__(256, 1) void conv3x3_wino_quad_ms_kernel(const WinoArgs a) {
    constexpr int QSTR = 272, HB = 100 * QSTR;
    __shared__ __attribute__((aligned(16))) char smem[2 * HB + 100 * 16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + 15) >> 4;
    const int qtile = blockIdx.x >> 2, qquad = blockIdx.x & 3;
    const int qy0 = (qtile / tiles_x) * 16 + 8 * (qquad >> 1), qx0 = (qtile % tiles_x) * 16 + 8 * (qquad & 1);
    if (qy0 >= H || qx0 >= W) return;
    const __amdgpu_buffer_rsrc_t r_u = rsrc_of(a.ubase, OOBW);
    const __amdgpu_buffer_rsrc_t r_urgb = rsrc_of(a.Urgb, 4u * 4096u);
    const __amdgpu_buffer_rsrc_t r_out = rsrc_of(a.out, (unsigned)H * (unsigned)W * 256u);
    const float act_lo = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.1f);
    f32x4 hv[7];
    auto halo_request = [&](const float* src) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int e = t + 256 * i, pe = e >> 4, ry = pe / 10, rx = pe - ry * 10, gy = qy0 - 1 + ry, gx = qx0 - 1 + rx;
            hv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e < 1600 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                hv[i] = *reinterpret_cast<const f32x4*>(src + ((long)gy * W + gx) * 64 + (e & 15) * 4);
        }
    };
    auto halo_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int e = t + 256 * i;
            if (e < 1600) *reinterpret_cast<f32x4*>(smem + buf * HB + (e >> 4) * QSTR + (e & 15) * 16) = hv[i];
        }
    };
    halo_request(a.srcs[0]);
    if (t < 100) {
        const int ry = t / 10, rx = t - ry * 10, gy = qy0 - 1 + ry, gx = qx0 - 1 + rx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = *reinterpret_cast<const f32x4*>(a.rgb + ((long)gy * W + gx) * 4);
        *reinterpret_cast<f32x4*>(smem + 2 * HB + t * 16) = v;
    }
    const int tyq = m >> 2, txq = m & 3;
    const unsigned wq16 = (unsigned)lane * 16u + (unsigned)wave * 1024u;
    f32x4 Bq[2][16], Br[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) Br[p] = bload4(r_urgb, (unsigned)lane * 16u, (unsigned)((p >> 2) * 4096 + (p & 3) * 1024));
#pragma unroll
    for (int p = 0; p < 16; ++p) Bq[0][p] = bload4(r_u, wq16, a.u_off[0] + (unsigned)((p >> 2) * 16384 + (p & 3) * 4096));
    halo_store(0);
    const float bgq = a.bias ? a.bias[wave * 16 + m] : 0.f;
    f32x4 acc[16], V[16];
    lds_bar();
    {   // the frame: channel kq of the lane's 4x4 patch, one float per position
        float dr[16], tr[16], Vr[16];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) dr[r * 4 + c] = *reinterpret_cast<const float*>(smem + 2 * HB + ((2 * tyq + r) * 10 + 2 * txq + c) * 16 + kq * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            tr[0 * 4 + c] = dr[0 * 4 + c] - dr[2 * 4 + c];
            tr[1 * 4 + c] = dr[1 * 4 + c] + dr[2 * 4 + c];
            tr[2 * 4 + c] = dr[2 * 4 + c] - dr[1 * 4 + c];
            tr[3 * 4 + c] = dr[1 * 4 + c] - dr[3 * 4 + c];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Vr[i * 4 + 0] = tr[i * 4 + 0] - tr[i * 4 + 2];
            Vr[i * 4 + 1] = tr[i * 4 + 1] + tr[i * 4 + 2];
            Vr[i * 4 + 2] = tr[i * 4 + 2] - tr[i * 4 + 1];
            Vr[i * 4 + 3] = tr[i * 4 + 1] - tr[i * 4 + 3];
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const f32x4 c0 = p == 5 ? f32x4{bgq, bgq, bgq, bgq} : f32x4{0.f, 0.f, 0.f, 0.f};
            const float bw = wave == 0 ? Br[p][0] : (wave == 1 ? Br[p][1] : (wave == 2 ? Br[p][2] : Br[p][3]));
            acc[p] = mfma16(Vr[p], bw, c0);
        }
    }
    for (int ks = 0; ks < a.nsrc; ++ks) {
        const bool more = ks + 1 < a.nsrc;
        const int buf = ks & 1;
        if (more) halo_request(a.srcs[ks + 1]);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            if (s4 < 3) {
#pragma unroll
                for (int p = 0; p < 16; ++p)
                    Bq[(s4 + 1) & 1][p] = bload4(r_u, wq16, a.u_off[ks] + (unsigned)(((s4 + 1) * 4 + (p >> 2)) * 16384 + (p & 3) * 4096));
            } else if (more) {
#pragma unroll
                for (int p = 0; p < 16; ++p) Bq[0][p] = bload4(r_u, wq16, a.u_off[ks + 1] + (unsigned)((p >> 2) * 16384 + (p & 3) * 4096));
            }
            f32x4 dq[4][4], tq4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    dq[r][c] = *reinterpret_cast<const f32x4*>(smem + buf * HB + ((2 * tyq + r) * 10 + 2 * txq + c) * QSTR + (16 * s4 + 4 * kq) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int c = 0; c < 4; ++c) tq4[c] = i == 0 ? dq[0][c] - dq[2][c] : (i == 1 ? dq[1][c] + dq[2][c] : (i == 2 ? dq[2][c] - dq[1][c] : dq[1][c] - dq[3][c]));
                V[4 * i + 0] = tq4[0] - tq4[2];
                V[4 * i + 1] = tq4[1] + tq4[2];
                V[4 * i + 2] = tq4[2] - tq4[1];
                V[4 * i + 3] = tq4[1] - tq4[3];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(V[i]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pr = 0; pr < 4; ++pr)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int pc = 0; pc < 4; ++pc) {
                        const int p = pr * 4 + pc;
                        acc[p] = mfma16(V[p][k], Bq[s4 & 1][p][k], acc[p]);
                    }
        }
        if (more) {
            halo_store(buf ^ 1);       // (last read a whole pass ago: every wave is past the barrier behind that pass)
            lds_bar();
        }
    }
    f32x4 w0[4], w1[4], yq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w0[i] = acc[i * 4 + 0] + acc[i * 4 + 1] + acc[i * 4 + 2];
        w1[i] = acc[i * 4 + 1] - acc[i * 4 + 2] - acc[i * 4 + 3];
    }
    yq[0] = w0[0] + w0[1] + w0[2];
    yq[1] = w1[0] + w1[1] + w1[2];
    yq[2] = w0[1] - w0[2] - w0[3];
    yq[3] = w1[1] - w1[2] - w1[3];
    const unsigned qo = (unsigned)((qy0 + 2 * kq) * W + qx0) * 256u + (unsigned)(wave * 16 + m) * 4u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        yq[q] = __builtin_elementwise_max(yq[q], act_lo * yq[q]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool inq = qy0 + 2 * kq + (q >> 1) < H && qx0 + 2 * r + (q & 1) < W;
            bstore1(r_out, inq ? qo : OOBW, (unsigned)((q >> 1) * W + 2 * r + (q & 1)) * 256u, yq[q][r]);
        }
    }
}
#endif      // WINO_MS_TU
#ifndef WINO_MS_TU

// ---- weight images ------------------------------------------------------------------------------------------------------------
// U = G g G^T per (output channel, input channel), times gamma[co] when given (see launch_wino_images), from a packed direct-conv
// B image (common.h: 9 chunks, chunk = tap; float index ((q * 2 + nt32) * 64 + h * 32 + n32) * 4 + j  <->  ci = 8 q + 4 h + j,
// co = 32 nt32 + n32) into 16 chunks (chunk = 4 * (ci >> 4) + position row i; float index
// ((pj * 4 + (co >> 4)) * 64 + ((ci >> 2) & 3) * 16 + (co & 15)) * 4 + (ci & 3)).  Computed in double, rounded once.
struct WinoImgArgs {
    const float* src[16];
    float* dst[16];
    const float* gamma;
};
__global__ __launch_bounds__(256) void wino_image_kernel(const WinoImgArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // [ci >> 4 | co >> 4 | (ci >> 2) & 3 | co & 15 | ci & 3]
    const int ci = ((idx >> 10) & 3) * 16 + ((idx >> 6) & 3) * 4 + (idx & 3), co = ((idx >> 8) & 3) * 16 + ((idx >> 2) & 15);
    const float* s = a.src[blockIdx.y];
    float* d = a.dst[blockIdx.y];
    double g[3][3];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
        g[tap / 3][tap % 3] = s[tap * 4096 + (((ci >> 3) * 2 + (co >> 5)) * 64 + ((ci >> 2) & 1) * 32 + (co & 31)) * 4 + (ci & 3)];
    const double gm = a.gamma ? (double)a.gamma[co] : 1.0;
    double tmp[4][3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        tmp[0][x] = g[0][x];
        tmp[1][x] = 0.5 * (g[0][x] + g[1][x] + g[2][x]);
        tmp[2][x] = 0.5 * (g[0][x] - g[1][x] + g[2][x]);
        tmp[3][x] = g[2][x];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double u[4] = {tmp[i][0], 0.5 * (tmp[i][0] + tmp[i][1] + tmp[i][2]), 0.5 * (tmp[i][0] - tmp[i][1] + tmp[i][2]), tmp[i][2]};
#pragma unroll
        for (int pj = 0; pj < 4; ++pj)
            d[((ci >> 4) * 4 + i) * 4096 + ((pj * 4 + (co >> 4)) * 64 + ((ci >> 2) & 3) * 16 + (co & 15)) * 4 + (ci & 3)] = (float)(u[pj] * gm);
    }
}

// the three 1x1 branch images (PACK_1X1 chunks, same float index as above with tap = branch) -> [step s][branch][N tile][lane][j]
__global__ __launch_bounds__(256) void wino_par_image_kernel(const float* __restrict__ src, float* __restrict__ dst) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // 3 * 4096
    const int br = idx >> 12, e = idx & 4095;
    const int ci = ((e >> 10) & 3) * 16 + ((e >> 6) & 3) * 4 + (e & 3), co = ((e >> 8) & 3) * 16 + ((e >> 2) & 15);
    const float v = src[br * 4096 + (((ci >> 3) * 2 + (co >> 5)) * 64 + ((ci >> 2) & 1) * 32 + (co & 31)) * 4 + (ci & 3)];
    dst[(((ci >> 4) * 3 + br) * 4 + (co >> 4)) * 256 + (((ci >> 2) & 3) * 16 + (co & 15)) * 4 + (ci & 3)] = v;
}

// the RGB frame's chunk (PACK_RGB4, common.h: float index ((q * 2 + nt32) * 64 + h * 32 + n32) * 4 + j  <->  tap 2 q + h, channel j,
// co = 32 nt32 + n32) -> four position-row chunks of 1024 floats: float index (pj * 64 + kq * 16 + n16) * 4 + nt  <->  position
// (i, pj), input channel kq (the MFMA's k index: one k-step covers R, G, B and the zero channel), co = 16 nt + n16
__global__ __launch_bounds__(256) void wino_rgb_image_kernel(const float* __restrict__ src, float* __restrict__ dst) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // 256 = 4 channels x 64 output channels
    const int c = idx >> 6, co = idx & 63;
    double g[3][3];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
        g[tap / 3][tap % 3] = src[(((tap >> 1) * 2 + (co >> 5)) * 64 + (tap & 1) * 32 + (co & 31)) * 4 + c];
    double tmp[4][3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        tmp[0][x] = g[0][x];
        tmp[1][x] = 0.5 * (g[0][x] + g[1][x] + g[2][x]);
        tmp[2][x] = 0.5 * (g[0][x] - g[1][x] + g[2][x]);
        tmp[3][x] = g[2][x];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double u[4] = {tmp[i][0], 0.5 * (tmp[i][0] + tmp[i][1] + tmp[i][2]), 0.5 * (tmp[i][0] - tmp[i][1] + tmp[i][2]), tmp[i][2]};
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) dst[i * 1024 + (pj * 64 + c * 16 + (co & 15)) * 4 + (co >> 4)] = (float)u[pj];
    }
}

#endif      // !WINO_MS_TU
}  // namespace

// The multi-source instantiation (the input convs) is its own translation unit, conv_wino_ms.hip = this file with WINO_MS_TU defined:
// its K loop is a run-time loop over the sources with all 64 accumulator quads carried around it, and the register-allocation flag that
// pays for the straight-line kernels (build_native.py: -greedy-reverse-local-assignment, fewer accumulator quads parked in arch VGPRs)
// costs it spill slots.  WinoArgs is the same struct in both units (same source); it crosses the boundary as an opaque pointer.
int launch_conv3x3_wino_ms_raw(const void* wino_args, int grid, int units, int ntiles, hipStream_t stream);

#ifdef WINO_MS_TU
int launch_conv3x3_wino_ms_raw(const void* wino_args, int grid, int units, int ntiles, hipStream_t stream) {
    static PnpPerDevice once;
    int unused = 0;
    const hipError_t attr_err = once.run([](int, int&) {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino_kernel<false, false, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS);
    }, &unused);
    if (attr_err != hipSuccess) return (int)attr_err;
    const WinoArgs& w = *static_cast<const WinoArgs*>(wino_args);
    if (units) hipLaunchKernelGGL(conv3x3_wino_quad_ms_kernel, dim3(4 * ntiles), dim3(256), 0, stream, w);
    else hipLaunchKernelGGL((conv3x3_wino_kernel<false, false, true>), dim3(grid), dim3(256), WINO_LDS, stream, w);
    return (int)hipGetLastError();
}
#else

int launch_wino_rgb_image(const float* src, float* dst, hipStream_t stream) {
    hipLaunchKernelGGL(wino_rgb_image_kernel, dim3(1), dim3(256), 0, stream, src, dst);
    return (int)hipGetLastError();
}

int launch_wino_images(const float* const* src, float* const* dst, int n, const float* gamma, hipStream_t stream) {
    if (n < 1 || n > 16) return PNP_ERR_BAD_ARG;
    WinoImgArgs a;
    for (int i = 0; i < 16; ++i) {
        a.src[i] = src[i < n ? i : 0];
        a.dst[i] = dst[i < n ? i : 0];
    }
    a.gamma = gamma;
    hipLaunchKernelGGL(wino_image_kernel, dim3(16, n), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}

int launch_wino_par_image(const float* src, float* dst, hipStream_t stream) {
    hipLaunchKernelGGL(wino_par_image_kernel, dim3(48), dim3(256), 0, stream, src, dst);
    return (int)hipGetLastError();
}

// the input conv: source 0 the RGB frame, then 1..3 64-channel sources, every one with its Winograd image (and the frame's)
bool conv_wino_ms_eligible(const ConvArgs& a, int cfg, int grid_y) {
    if (!a.wwino_rgb || a.prec != 0 || cfg == CONV_CFG_RGB || grid_y != 1 || a.out_mode != 0) return false;
    if (a.nsrc < 2 || a.nsrc > 4 || a.src_c[0] != 4 || a.src_f16 || a.out_f16 || a.out16) return false;
    if (a.wpar || a.residual || a.gamma) return false;
    for (int s = 1; s < a.nsrc; ++s)
        if (a.src_c[s] != 64 || !a.wwino_src[s]) return false;
    return (long)a.H * a.W * 256 < ((long)1 << 32) - 65536;
}

bool conv_wino_eligible(const ConvArgs& a, int cfg, int grid_y) {
    if (!a.wwino || a.prec != 0 || cfg == CONV_CFG_RGB || grid_y != 1 || a.out_mode != 0) return false;
    if (a.nsrc != 1 || a.src_c[0] != 64 || a.src_f16 || a.out_f16 || a.out16) return false;
    if (a.wpar && (!a.wwino_par || !a.par)) return false;
    return (long)a.H * a.W * 256 < ((long)1 << 32) - 65536;
}

int launch_conv3x3_wino(const ConvArgs& a, hipStream_t stream) {
    static PnpPerDevice once;
    int cus = 256;
    const hipError_t attr_err = once.run([](int dev, int& g) {
        hipError_t e = hipSuccess;
        const void* fo = reinterpret_cast<const void*>(conv3x3_wino_gated_kernel<false>);
        e = hipFuncSetAttribute(fo, hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS);
        const void* fo_res = reinterpret_cast<const void*>(conv3x3_wino_gated_kernel<true>);
        if (e == hipSuccess) e = hipFuncSetAttribute(fo_res, hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS);
        const void* fns[4] = {reinterpret_cast<const void*>(conv3x3_wino_kernel<false, false, false>),
                              reinterpret_cast<const void*>(conv3x3_wino_kernel<false, true, false>),
                              reinterpret_cast<const void*>(conv3x3_wino_kernel<true, false, false>),
                              reinterpret_cast<const void*>(conv3x3_wino_kernel<true, true, false>)};
        for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS);
        g = 256;
        (void)hipDeviceGetAttribute(&g, hipDeviceAttributeMultiprocessorCount, dev);
        return e;
    }, &cus);
    if (attr_err != hipSuccess) return (int)attr_err;
    WinoArgs w;
    memset(&w, 0, sizeof(w));
    w.src = a.src[0];
    w.U = a.wwino;
    w.Upar = a.wpar ? a.wwino_par : nullptr;
    w.par = a.par;
    w.par_plane = a.par_plane;
    w.par_flags = a.par_flags;
    w.bias = a.bias;
    w.gamma = a.gamma;
    w.residual = a.residual;
    w.out = a.out;
    w.H = a.H;
    w.W = a.W;
    w.act = a.act;
    w.dbg = a.dbg;
    w.quad = WINO_QUAD;
    const int ntiles = ((a.W + 15) / 16) * ((a.H + 15) / 16);
    if (a.wino_units && a.wwino) {                          // small frames: one block per quadrant unit
        const dim3 gq(4 * ntiles), bq(256);
        if (a.wpar && a.par_any) {
            w.gate = a.par_any;
            if (a.residual) hipLaunchKernelGGL((conv3x3_wino_quad_gated_kernel<true>), gq, bq, 0, stream, w);
            else hipLaunchKernelGGL((conv3x3_wino_quad_gated_kernel<false>), gq, bq, 0, stream, w);
        } else if (a.wpar && a.residual) hipLaunchKernelGGL((conv3x3_wino_quad_kernel<true, true>), gq, bq, 0, stream, w);
        else if (a.wpar) hipLaunchKernelGGL((conv3x3_wino_quad_kernel<true, false>), gq, bq, 0, stream, w);
        else if (a.residual) hipLaunchKernelGGL((conv3x3_wino_quad_kernel<false, true>), gq, bq, 0, stream, w);
        else hipLaunchKernelGGL((conv3x3_wino_quad_kernel<false, false>), gq, bq, 0, stream, w);
        return (int)hipGetLastError();
    }
    if (a.residual && a.act != 0) return PNP_ERR_UNSUPPORTED;      // (tile kernels: residual bodies have no activation; the unit kernels above do)
    int grid = ntiles < cus ? ntiles : cus;                 // one resident block per CU
    if (grid >= 8) grid -= grid % 8;
    if (conv_wino_ms_eligible(a, CONV_CFG_BIG, 1) && !a.wwino) {
        // the input conv: the frame + the 64-channel sources; the images must sit within 4 GiB of the lowest one (one descriptor)
        const float* lo = a.wwino_src[1];
        for (int s = 2; s < a.nsrc; ++s)
            if (a.wwino_src[s] < lo) lo = a.wwino_src[s];
        w.ubase = lo;
        w.nsrc = a.nsrc - 1;
        for (int s = 1; s < a.nsrc; ++s) {
            const long off = (const char*)a.wwino_src[s] - (const char*)lo;
            if (off < 0 || off >= ((long)1 << 32) - PNP_WINO_IMG_FLOATS * 4) return PNP_ERR_UNSUPPORTED;
            w.srcs[s - 1] = a.src[s];
            w.u_off[s - 1] = (unsigned)off;
        }
        w.rgb = a.src[0];
        w.Urgb = a.wwino_rgb;
        return launch_conv3x3_wino_ms_raw(&w, grid, a.wino_units ? 1 : 0, ntiles, stream);
    } else if (a.wpar && a.par_any) {
        // ONE launch behind a device-side gate on the frame's partition word (launch_par_frame_any): the fold-only body when EVERY 8x8
        // quadrant of the frame is all zero or carries one constant plane (one-hot maps on >= 8x8 codec blocks, frames without records),
        // the branch body otherwise.  Bit-identical results either way.
        w.gate = a.par_any;
        if (a.residual) hipLaunchKernelGGL((conv3x3_wino_gated_kernel<true>), dim3(grid), dim3(256), WINO_LDS, stream, w);       // channel-last blocks
        else hipLaunchKernelGGL((conv3x3_wino_gated_kernel<false>), dim3(grid), dim3(256), WINO_LDS, stream, w);
    } else if (a.wpar && a.residual) hipLaunchKernelGGL((conv3x3_wino_kernel<true, true, false>), dim3(grid), dim3(256), WINO_LDS, stream, w);
    else if (a.wpar) hipLaunchKernelGGL((conv3x3_wino_kernel<true, false, false>), dim3(grid), dim3(256), WINO_LDS, stream, w);
    else if (a.residual) hipLaunchKernelGGL((conv3x3_wino_kernel<false, true, false>), dim3(grid), dim3(256), WINO_LDS, stream, w);
    else hipLaunchKernelGGL((conv3x3_wino_kernel<false, false, false>), dim3(grid), dim3(256), WINO_LDS, stream, w);
    return (int)hipGetLastError();
}
#endif      // !WINO_MS_TU
