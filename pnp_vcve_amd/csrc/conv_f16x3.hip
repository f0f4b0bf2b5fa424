// Split-fp16 ("f16x3") variant of the fused 3x3 conv: fp32-LEVEL results from the fp16 matrix pipe.
//
// Every operand is split into two fp16 numbers, x = hi + lo / 2048 with hi = fp16(x), lo = fp16((x - hi) * 2048), and a product
// is contracted as three MFMAs -- hi*hi into one fp32 accumulator, lo*hi + hi*lo into a second one that is scaled by 1/2048 at
// the end (the scaling keeps the low parts in fp16's normal range; the dropped lo*lo term is 2^-22 relative).  hi + lo carries
// 22 of fp32's 24 significand bits and every fp16 x fp16 product is exact in fp32, so a conv differs from the exact-fp32 MFMA
// kernels by ~1e-6 relative (1.2e-7 on a clip, the fp32 path's own distance from the reference; plain fp16 operands,
// PNP_PREC_F16: 4e-5) -- at 3 fp16 MFMAs per product against fp32 MFMAs at 1/16 of the rate: ~5x fewer matrix-pipe cycles than the
// fp32 path.  Feature maps stay fp32 in HBM (reader and writer both): the mode
// changes no data layout, only how a conv multiplies.  Opt-in (PNP_PREC_F16X3); the default and the headline stay exact fp32.
//
// Kernel (the r04 form; details in front of the kernel below and in DESIGN.md 3.6): persistent 4-wave blocks, two per CU, each
// drawing 8x16 tiles from a per-XCD queue (X3Args::queue): the fp32 halo -- requested one tile ahead, riding in registers through
// the end of the K loop -- is split into the (hi, lo) A tile in LDS; weight chunks stream from L2 through a 3-slot ring.  The
// weight image interleaves the two halves of the split so that every 8 KiB chunk (tap, k-half) = [hi N0..3, lo N0..3] units is
// self-contained: one 32-deep k-step of v_mfma_f32_16x16x32_f16, 4 A + 8 B fragment reads for 24 MFMAs per wave, the reads dealt
// one per MFMA gap (a wave issues in order: a burst of reads in front of the MFMAs costs a lone wave 48-60 cycles per MFMA, dealt
// reads 33).  Chunks are requested four ahead into register sets (L2 latency) and written into the ring two ahead, in the middle
// of a chunk, so that the per-chunk barrier waits for that write only (counted lgkmcnt) while the next k-step's fragments,
// fetched before the barrier, stay in flight.  75 KiB of LDS -> two blocks per CU.  A partition branch contracts a MASKED A
// fragment with weights scaled by 1/255 at pack time on tiles whose partition values are all 0 or 1/255 (what the reference's
// loader writes), and re-splits par_j(pixel) * x per fragment otherwise.  An input conv over the virtual concat is ONE launch (MS).
// Pixel-shuffle (out_mode 1) and channel-block (out_mode 4) convs are one launch per 64-channel output block with an affine
// output mapping (o_sy, o_sx, o_c0).
#include "conv_mfma.h"
#include "f16_util.h"
#include <string.h>
#include <utility>

template <int V> struct X3Tag { static constexpr int v = V; };                  // a compile-time int as a lambda argument
template <int... I, class F> __device__ __forceinline__ void x3_static_for(std::integer_sequence<int, I...>, F&& f) { (f(X3Tag<I>{}), ...); }

namespace {

constexpr int X3_RING = 3;
constexpr int X3_CHUNK = 8 * UNIT;                            // 4 k-steps x 2 N tiles
static_assert(2 * ((TH + 2) * PW * 288 + 272 + X3_RING * X3_CHUNK) <= 160 * 1024, "two blocks per CU");
constexpr int X3_AIT = (NPIX * 16 + 255) / 256;               // 12 16-byte halo loads per thread (fp32 source)
constexpr float X3_SCALE = 2048.f, X3_INV = 1.f / 2048.f;

struct X3Args {
    const float* src;            // NHWC64 fp32
    const _Float16* w;           // split image (launch_f16x3_image): 18 chunks of 8 units
    const _Float16* wpar;        // 6 chunks (branch, k-half) or nullptr; wpar_scaled: 6 more, the same images x PNP_PAR_UNIT
    int wpar_scaled;
    const float* par;
    long par_plane;
    const int* par_flags;
    const float *bias, *gamma, *residual;
    float* out;
    int res_pre;                 // residual is added BEFORE the activation (partial sum of a source chain)
    unsigned o_sy, o_sx, o_c0;   // output addressing: byte offset of pixel (gy, gx) = gy * o_sy + gx * o_sx + o_c0
    unsigned out_bytes;
    int H, W, act;
    unsigned long long* dbg;     // timeline: 8 u64 per block or nullptr
    // MS kernels (an input conv in ONE launch): further 64-channel sources of the same size, accumulated into the same MFMA chains
    // source after source (src / w are source 0), and optionally the RGB frame in front of them
    int nsrc;                    // 64-channel sources: 1..3
    const float* src2[2];
    const _Float16* w2[2];
    const float* lr4;            // (H, W, 4) fp32, 4th channel zero, or nullptr
    const _Float16* wlr;         // its split image: 2 chunks (k = 4 tap + channel)
    // Tile queue (r04) or nullptr: 8 per-XCD ticket counters + a count of finished blocks, all zero between launches.  A block's first
    // tile is static (band_lo + blockIdx / 8); every further one is band_lo + per + ticket.  Tiles differ in cost (a front-half tile runs
    // 18..24 chunks) and so do blocks (the one that reached its CU first wins the issue arbitration: 19.6 k against 24.6 k cycles per
    // tile): on a static split the launch lasts as long as its unluckiest block
    int* queue;
};

// Halo geometry: 10 rows x 18 pixels x 16 float4.  Requests 0..9: row k, pixels 0..15 (thread t: pixel t >> 4, float4 t & 15);
// requests 10, 11: the two right-hand pixel columns (item j = t + 256 (k - 10): row j >> 5, pixel 16 + ((j >> 4) & 1)) -- so the
// global offset and the LDS address of request k are one per-thread base plus k times a constant (no per-request registers).
// S4: 4x16-pixel tiles for frames with at most 256 8x16 tiles (half the CUs would otherwise idle): the four waves are 2 pixel-row
// pairs x 2 N halves, a wave owns 32 pixels x 32 channels; halo 6 rows = requests 0..5 + one side request (t < 192).
//
// MFMA shape (r04): v_mfma_f32_16x16x32_f16.  The kernel is POWER-limited -- in r04's A/B 10 % fewer cycles per tile came back as
// a 5 % lower clock -- and the clock the chip holds under this load depends on the shape: the same LDS-fed split loop sustains
// 1.67 PFLOP/s at 1.70 GHz on 16x16x32 against 1.43 PFLOP/s at 1.50 GHz on 32x32x16 (tools/ubench/ub_shape16.hip,
// profiles/r04_ub_shape16.txt: +17 %).  A wave's 32 pixels are two M tiles (its two pixel rows: MFMA row = pixel column), its
// channels 16-wide N tiles; a chunk (tap, k-half) is ONE 32-deep k-step: 4 A fragments (2 rows x hi / lo) and per N tile a
// (hi, lo) pair of B fragments for 6 MFMAs of 16 cycles -- the same 1 KiB of fragment reads per 32 matrix cycles as before.
// LDS image of the halo: pixel = [64 ch hi | 64 ch lo | 32 B pad] = 288 B.  A lane group of a ds_read_b128 holds 8 pixels of
// k-group g and the 8 other pixels of k-group g ^ 1 (16 B apart): with a pixel stride of 18 (any even number of) 16-B slots
// the first set lands on the even slots and the second on the odd ones for every tap -- conflict-free; the 144-B stride of the
// 32x32x16 image (one plane per pixel) cannot do that for any lane -> pixel mapping.
constexpr int XPSB = 288, XRSB = PW * XPSB;                  // pixel / row stride of the split A tile
constexpr int XPARK = 256 + 16;                              // where idle lanes of the side requests put their (unused) halves; + the queue mailbox

// a pointer chosen at run time by wave-uniform values, made PROVABLY uniform (the descriptor built from it must live in scalar
// registers: otherwise every buffer access becomes a waterfall loop)
template <class T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
}

template <bool PAR, bool DBG, bool S4, bool MS = false>
__global__ __launch_bounds__(256, 2) void conv3x3_f16x3_kernel(const X3Args a) {
    static_assert(!MS || (!PAR && !DBG), "the multi-source kernel is the plain conv");
    constexpr int NC = 18 + (PAR ? 6 : 0);                   // chunks: two k-halves per tap, then two per partition branch
    // register sets of weight chunks (chunk c travels in set c % NSET, requested NSET chunks ahead, written to the ring two ahead).
    constexpr int NSET = 4;                                  // (3 starves the ring: K loop 13.6 k -> 18.8 k cycles per tile)
    constexpr int WPT = 2;
    constexpr int THX = S4 ? 4 : TH, ROWS = THX + 2;                               // tile rows, halo rows
    constexpr int LBY = MS ? ROWS * PW * 16 : 0;                                   // MS: the RGB halo, [4 ch hi | 4 ch lo] = 16 B per pixel
    constexpr int ABY = ROWS * XRSB + XPARK + LBY;                                 // bytes in front of the weight ring
    constexpr int NTW = S4 ? 2 : 4;                          // 16-channel N tiles per wave
    constexpr int NREQ = S4 ? ROWS + 1 : X3_AIT;             // 16-byte halo requests per thread
    constexpr int CW = NTW * 4, PPI = 64 / CW, EIT = 32 / PPI;   // epilogue: float4 per pixel in the wave's N range, pixels per instruction
    constexpr int RQR = 13;                              // chunk at whose top the residual rows are requested
    constexpr int QW = 5, QR = 8;                        // chunks at whose top the queue ticket enters / leaves the mailbox (two barriers apart)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned long long dbg_t0 = 0, dbg_p = 0, dbg_k = 0, dbg_e = 0;
    int dbg_n = 0;
    unsigned long long dbg_r0 = 0;
    if (DBG) {
        dbg_t0 = __builtin_amdgcn_s_memtime();
        dbg_r0 = __builtin_amdgcn_s_memrealtime();
    }
    const int lm = lane & 15, lg = lane >> 4;                // MFMA row / column (pixel column, output channel) and k-group of the lane
    const int wrow = S4 ? wave >> 1 : wave, wn = S4 ? wave & 1 : 0;      // the wave's pixel-row pair and N half
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TW - 1) / TW;
    const int ntiles = tiles_x * ((H + THX - 1) / THX);
    // Persistent blocks, two per CU.  XCD x (blockIdx & 7) owns the contiguous band of tiles [ntiles*x/8, ntiles*(x+1)/8); its
    // blocks walk the band interleaved (block i takes tiles i, i + per, ...), so at any moment an XCD works on one window of
    // consecutive tiles whose shared halo rows meet in its L2.
    const int per = gridDim.x >> 3, xcd = blockIdx.x & 7;
    const int band_lo = (int)((long)ntiles * xcd / 8), band_hi = (int)((long)ntiles * (xcd + 1) / 8);
    int tile = band_lo + (blockIdx.x >> 3);
    int* const queue = a.queue;                              // (uniform) nullptr: the static walk tile, tile + per, ...
    auto leave_queue = [&]() {                               // the last block to leave zeroes the counters for the next launch
        if (queue && t == 0) {
            if (atomicAdd(queue + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) __hip_atomic_store(queue + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    if (tile >= band_hi) {
        leave_queue();
        return;
    }
    char* const sR = smem + ABY;

    const unsigned map_bytes = (unsigned)H * (unsigned)W * 256u;
    const unsigned row_bytes = (unsigned)W * 256u;
    __amdgpu_buffer_rsrc_t r_src = make_rsrc(a.src, map_bytes);      // the source whose halo is requested next (MS: changes per pass)
    const __amdgpu_buffer_rsrc_t r_res = make_rsrc(a.residual ? (const void*)a.residual : (const void*)a.src, a.residual ? map_bytes : 0);
    const __amdgpu_buffer_rsrc_t r_par = make_rsrc(PAR ? (const void*)a.par : (const void*)a.src, PAR ? (unsigned)(3 * a.par_plane * 4) : 0);
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(a.out, a.out_bytes);
    const unsigned o_sy = (unsigned)__builtin_amdgcn_readfirstlane((int)a.o_sy), o_sx = (unsigned)__builtin_amdgcn_readfirstlane((int)a.o_sx);
    const __amdgpu_buffer_rsrc_t r_flags = make_rsrc((PAR && a.par_flags) ? (const void*)a.par_flags : (const void*)a.src,
                                                     (PAR && a.par_flags) ? (unsigned)(tiles_x * ((H + TH - 1) / TH)) * 4u : 0);   // per 8x16 tile
    // weight chunks through descriptors too: voffset = 16 t for every load, the chunk in the SCALAR offset -- no per-chunk
    // 64-bit address pairs for the compiler to hoist out of the tile loop and spill
    __amdgpu_buffer_rsrc_t r_w = make_rsrc(a.w, 18u * X3_CHUNK);       // the weights whose chunks are requested next (MS: per pass)
    const __amdgpu_buffer_rsrc_t r_wp = make_rsrc(PAR ? (const void*)a.wpar : (const void*)a.w, (PAR && a.wpar_scaled ? 12u : 6u) * X3_CHUNK);
    const bool has_lr = MS && a.lr4 != nullptr;
    const __amdgpu_buffer_rsrc_t r_lr = make_rsrc(has_lr ? (const void*)a.lr4 : (const void*)a.src, has_lr ? (unsigned)H * (unsigned)W * 16u : 0);
    const __amdgpu_buffer_rsrc_t r_wlr = make_rsrc(has_lr ? (const void*)a.wlr : (const void*)a.w, 2u * X3_CHUNK);
    char* const sL = smem + ROWS * XRSB + XPARK;
    char* const mailbox = smem + ROWS * XRSB + 256;
    f32x4 lrreg = (f32x4)(0.f);                              // the thread's pixel of the next tile's RGB halo (t < ROWS * PW)
    // RGB A fragments: k = 32 kh + 8 lg + jj = 4 tap + channel -> the lane's 8 k values are taps 8 kh + 2 lg, + 1 (4 channels each);
    // taps beyond 8 carry zero weights and re-read tap 8 (finite values)
    int rgb_off[2][2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int tap = 8 * kh + 2 * lg + e;
            tap = tap > 8 ? 8 : tap;
            rgb_off[kh][e] = ((tap / 3) * PW + tap % 3) * 16;
        }
    const int ec = lane % CW, ep = lane / CW;
    const float neg_slope = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.1f);
    const float k_pre = a.res_pre ? 1.f : 0.f, k_post = 1.f - k_pre;
    float bco[NTW], gco[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {                          // the lane's output channel in N tile j of the wave
        bco[j] = a.bias ? a.bias[wn * 32 + j * 16 + lm] : 0.f;
        gco[j] = a.gamma ? a.gamma[wn * 32 + j * 16 + lm] : 1.f;
    }
    const int a_off = 2 * wrow * XRSB + lm * XPSB + 16 * lg;         // + row r, tap (dy, dx), plane (128), k-half (64)
    // per-thread halo bases (see above), from a thread index tq
    struct HaloBases {
        int hp, hcs, h2r, h2x;
        unsigned g_main, g_side;
    };
    auto halo_bases = [&](int tq) {
        HaloBases b;
        b.hp = tq >> 4, b.hcs = tq & 15;
        b.h2r = tq >> 5, b.h2x = 16 + ((tq >> 4) & 1);
        b.g_main = (unsigned)b.hp * 256u + (unsigned)b.hcs * 16u;                                        // + (row k) * row_bytes
        b.g_side = (unsigned)b.h2r * row_bytes + (unsigned)b.h2x * 256u + (unsigned)b.hcs * 16u;         // + 8 rows for request 11
        return b;
    };

    // ---- requests that travel ahead of their tile: the fp32 halo (12 x 16 B per thread) and its partition values / flags
    f32x4 areg[NREQ];
    float pvn[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};  // partition planes at the lane's two pixels (rows 2 wrow, 2 wrow + 1; column lm)
    int pfn = 0;
    // part 0..2: halo rows [0,4), [4,6), [6,8) (S4: [0,2), [2,3), [3,4)); part 3: the remaining rows, the side columns and the
    // partition values / flags.  The parts are issued where registers are free (see the K loop).
    bool want_lr = false;                                    // the tile being requested starts with the RGB pass
    auto request_tile_part = [&](int tl, bool live, int part) {     // !live: every offset out of range (loads return 0, no branch)
        const int ty0 = (tl / tiles_x) * THX, tx0 = (tl % tiles_x) * TW;
        const unsigned hbase = (unsigned)((ty0 - 1) * W + (tx0 - 1)) * 256u;
        const HaloBases hb = halo_bases(t);
        const unsigned g_main = hb.g_main, g_side = hb.g_side;
        const bool ok_main = live & ((unsigned)(tx0 - 1 + hb.hp) < (unsigned)W);    // rows outside the image leave the descriptor by themselves
        const bool ok_side = live & ((unsigned)(tx0 - 1 + hb.h2x) < (unsigned)W);
        constexpr int K1 = S4 ? 2 : 4, K2 = S4 ? 3 : 6, K3 = S4 ? 4 : 8;
        const int k0 = part == 0 ? 0 : (part == 1 ? K1 : (part == 2 ? K2 : K3));
        const int k1 = part == 0 ? K1 : (part == 1 ? K2 : (part == 2 ? K3 : ROWS));
#pragma unroll
        for (int k = 0; k < ROWS; ++k)
            if (k >= k0 && k < k1) areg[k] = buf_load4(r_src, ok_main ? hbase + g_main + (unsigned)k * row_bytes : OOB);
        if (part != 3) return;
        if (MS && want_lr) {
            const int ry = t / PW, rx = t - ry * PW;
            const bool ok = live & (t < ROWS * PW) & ((unsigned)(tx0 - 1 + rx) < (unsigned)W);
            lrreg = buf_load4(r_lr, ok ? (unsigned)((ty0 - 1 + ry) * W + (tx0 - 1 + rx)) * 16u : OOB);
        }
        if (S4) {
            areg[ROWS] = buf_load4(r_src, (ok_side & (t < 192)) ? hbase + g_side : OOB);
        } else {
            areg[NREQ - 2] = buf_load4(r_src, ok_side ? hbase + g_side : OOB);
            areg[NREQ - 1] = buf_load4(r_src, (ok_side & (t < 64)) ? hbase + g_side + 8u * row_bytes : OOB);
        }
        if (PAR) {
            const int gx = tx0 + lm;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int gy = ty0 + 2 * wrow + r;
#pragma unroll
                for (int jj = 0; jj < 3; ++jj)
                    pvn[jj][r] = buf_load1(r_par, (live & (gy < H) & (gx < W)) ? (unsigned)(jj * a.par_plane + (long)gy * W + gx) * 4u : OOB);
            }
            const int fidx = S4 ? ((tl / tiles_x) >> 1) * tiles_x + tl % tiles_x : tl;       // the 8x16 tile this one lies in
            pfn = __builtin_bit_cast(int, buf_load1(r_flags, live ? (unsigned)fidx * 4u : OOB));
        }
    };
    auto request_tile = [&](int tl, bool live) {
#pragma unroll
        for (int part = 0; part < 4; ++part) request_tile_part(tl, live, part);
    };
    int ncr = NC, bs0 = 0, bs1 = 1, bs2 = 2;
    // PAR, per tile: every partition value of the needed planes is 0 or exactly PNP_PAR_UNIT (flag bits 3..5: what the reference's loader
    // writes) and the scaled branch images are there -> the branch chunks contract a MASKED A operand with weights scaled at pack time
    // instead of re-splitting par_j(pixel) * x per fragment (~180 vector instructions per chunk and wave: the front half's bound)
    bool fast = false;
    auto bsel = [&](int j) { return j == 0 ? bs0 : (j == 1 ? bs1 : bs2); };
    f32x4 wreg[NSET][WPT];
    auto request_chunk = [&](int c) {                       // c < 18 compile-time, branch chunks via bsel; into set c % NSET
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int so = (c < 18 ? c : bsel((c - 18) >> 1) * 2 + (c & 1) + (fast ? 6 : 0)) * X3_CHUNK + i * 4096;
            wreg[c % NSET][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c < 18 ? r_w : r_wp, t * 16, so, 0));
        }
    };
    auto request_first_chunks = [&](bool rgb_first) {       // chunks 0..NSET-1 of a pass: the same images for every tile
        if (MS && rgb_first) {                              // the RGB pass's two chunks travel in sets 2, 3; source 0's chunks 2, 3 follow
            request_chunk(0);
            request_chunk(1);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < WPT; ++i)
                    wreg[2 + c][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_wlr, t * 16, c * X3_CHUNK + i * 4096, 0));
            return;
        }
#pragma unroll
        for (int c = 0; c < NSET; ++c) request_chunk(c);
    };
    want_lr = has_lr;
    request_tile(tile, true);
    request_first_chunks(has_lr);

    for (;;) {
        unsigned long long dbg_a = 0, dbg_b = 0, dbg_c = 0;
        if (DBG) dbg_a = __builtin_amdgcn_s_memtime();
        const int ty0 = (tile / tiles_x) * THX, tx0 = (tile % tiles_x) * TW;
        // the tile after this one: static, or (queue) a ticket drawn now by thread 0, passed through the mailbox at chunks QW -> QR and
        // known to every wave from chunk QR on (first needed at chunk 14)
        int next = tile + per;
        bool has_next = !queue && next < band_hi;
        int ticket = 0;
        if (queue && t == 0) ticket = atomicAdd(queue + xcd, 1);
        // ---- K loop: chunk c (one 32-deep k-step) from ring slot c % 3: hi*hi -> acc_hi, lo*hi + hi*lo -> acc_lo
        f32x4 acc_hi[2][NTW], acc_lo[2][NTW];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                acc_hi[r][j] = (f32x4)(0.f);
                acc_lo[r][j] = (f32x4)(0.f);
            }
        auto fold = [&](bool with_bias) {          // acc_hi <- ((acc_hi + acc_lo / 2048) [+ bias]) [* gamma]; acc_lo <- 0
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const f32x4 v = acc_hi[r][j] + acc_lo[r][j] * X3_INV;
                    acc_hi[r][j] = with_bias ? (v + bco[j]) * gco[j] : v;
                    acc_lo[r][j] = (f32x4)(0.f);
                }
        };
        f32x4 res4[EIT];                                     // set at chunk RQR of every pass (zeros but in the last one): not live across passes
        // ---- MS with the RGB frame: its 9 x 4 (3 + a zero) channels are K = 36 of two 32-deep chunks, contracted in front of the 64-channel
        //      sources from a 16-byte-per-pixel halo of its own.  Short (48 MFMAs per wave): no software pipeline.
        if (MS && has_lr) {
            if (t < ROWS * PW) {
                const f32x4 xc = clamp_h(lrreg);
                const h4 hi = __builtin_convertvector(xc, h4);
                *reinterpret_cast<h4*>(sL + t * 16) = hi;
                *reinterpret_cast<h4*>(sL + t * 16 + 8) = __builtin_convertvector((xc - __builtin_convertvector(hi, f32x4)) * X3_SCALE, h4);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(sR + c * X3_CHUNK + (t + 256 * i) * 16) = wreg[2 + c][i];
            lds_barrier();
            request_chunk(2);            // sets 2, 3 are free again: source 0's chunks 2, 3 (r_w is source 0's image here)
            request_chunk(3);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                h8 xr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const char* bp = sL + ((2 * wrow + (u & 1)) * PW + lm) * 16 + (u >> 1) * 8;
                    const h4 t0 = *reinterpret_cast<const h4*>(bp + rgb_off[kh][0]);
                    const h4 t1 = *reinterpret_cast<const h4*>(bp + rgb_off[kh][1]);
                    xr[u] = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const char* bl = sR + kh * X3_CHUNK + (S4 ? 2 * wn + j : j) * UNIT + lane * 16;
                    const h8 bh = *reinterpret_cast<const h8*>(bl), bo = *reinterpret_cast<const h8*>(bl + 4 * UNIT);
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        acc_hi[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xr[r], bh, acc_hi[r][j], 0, 0, 0);
                        acc_lo[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xr[2 + r], bh, acc_lo[r][j], 0, 0, 0);
                        acc_lo[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xr[r], bo, acc_lo[r][j], 0, 0, 0);
                    }
                }
            }
            lds_barrier();               // ring slots 0, 1 are read: source 0's first chunks may overwrite them
        }
        // ---- MS: one pass per 64-channel source, all into the same accumulators (a conv over the virtual concat, iconvsr_ipb_par.py:90,125,
        //      as ONE K = nsrc x 576 contraction); a single pass otherwise.  During pass p the halo of pass p + 1 (or of the next
        //      tile's pass 0) and its first weight chunks are requested, exactly as a single-source tile requests the next tile's.
        const int npass = MS ? a.nsrc : 1;
#pragma nounroll
        for (int pass = 0;; ++pass) {
        const bool last = pass + 1 == npass;
        int nx_tile = last ? (has_next ? next : tile) : tile;
        bool nx_live = last ? has_next : true;
        if (MS) {       // both descriptors are rebuilt in every pass (nothing loop-carried: a descriptor phi would leave the scalar registers)
            // (static indices only: a run-time index into the kernel arguments would move them to scratch)
            r_src = make_rsrc(uniform_ptr(last ? a.src : (pass == 0 ? a.src2[0] : a.src2[1])), map_bytes);        // whose halo this pass requests
            r_w = make_rsrc(uniform_ptr(pass == 0 ? a.w : (pass == 1 ? a.w2[0] : a.w2[1])), 18u * X3_CHUNK);  // whose chunks it streams
        }
        want_lr = has_lr && last;
        // ---- fp32 halo -> the split A tile: hi = fp16(x) (saturating), lo = fp16((x - hi) * 2048); chunks 0, 1 -> ring
        const HaloBases hs = halo_bases(t);
        char* const l_main = smem + hs.hp * XPSB + hs.hcs * 8;                                        // + k * XRSB
        // request 11 covers rows 8, 9 only (t < 64): the other threads park their (zero) value behind the tile
        char* const l_side = smem + hs.h2r * XRSB + hs.h2x * XPSB + hs.hcs * 8;
        char* const l_park = smem + ROWS * XRSB + hs.hcs * 8;
        char* const l_side11 = t < 64 ? l_side + 8 * XRSB : l_park;
        char* const l_side6 = t < 192 ? l_side : l_park;          // S4: the one side request covers rows 0..5
#pragma unroll
        for (int k = 0; k < NREQ; ++k) {
            // x saturates at +-65504 as a WHOLE: the remainder is taken from the clamped value, so it is at most half an fp16 ulp
            // (<= 16) and its scaled form (<= 32768) needs no clamp of its own; in-range values are untouched
            char* d = k < ROWS ? l_main + k * XRSB : (S4 ? l_side6 : (k == ROWS ? l_side : l_side11));
#ifdef X3_EXP_VERBATIM        /* upper bound of producer-split maps: the 16 bytes go to LDS as they are (wrong results) */
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<f32x2*>(d) = areg[k].xy;
            *reinterpret_cast<f32x2*>(d + 128) = areg[k].zw;
#else
            const f32x4 xc = clamp_h(areg[k]);
            const h4 hi = __builtin_convertvector(xc, h4);
            *reinterpret_cast<h4*>(d) = hi;
            *reinterpret_cast<h4*>(d + 128) = __builtin_convertvector((xc - __builtin_convertvector(hi, f32x4)) * X3_SCALE, h4);
#endif
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(sR + c * X3_CHUNK + (t + 256 * i) * 16) = wreg[c][i];
        float pv[3][2];
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) pv[jj][0] = pvn[jj][0], pv[jj][1] = pvn[jj][1];
        if (PAR) {
            ncr = NC;
            bs0 = 0, bs1 = 1, bs2 = 2;
            fast = false;
            if (a.par_flags) {
                const int fraw = __builtin_amdgcn_readfirstlane(pfn);
                const int f0 = fraw & 7;
                fast = a.wpar_scaled != 0 && ((fraw >> 3) & f0) == f0;
                const int f1 = f0 & (f0 - 1), f2 = f1 & (f1 - 1);
                ncr = 18 + 2 * __builtin_popcount(f0);
                bs0 = f0 ? __builtin_ctz(f0) : 0;
                bs1 = f1 ? __builtin_ctz(f1) : 0;
                bs2 = f2 ? __builtin_ctz(f2) : 0;
            }
        }
        lds_barrier();
        if (DBG) dbg_b = __builtin_amdgcn_s_memtime();

        // fragments: A of a chunk = [hi row 0, hi row 1, lo row 0, lo row 1]; B of (chunk, N tile j) = (hi, lo); chunk units in the
        // ring slot: [hi N0..N3, lo N0..N3]
        auto load_a = [&](int c, int u) {                   // c, u compile-time after unrolling
            const int tap = c >> 1, kh = c & 1, dy = c < 18 ? tap / 3 : 1, dx = c < 18 ? tap % 3 : 1;
            return *reinterpret_cast<const h8*>(smem + a_off + ((u & 1) + dy) * XRSB + dx * XPSB + (u >> 1) * 128 + kh * 64);
        };
        auto load_b = [&](int c, int j, int plane) {
            return *reinterpret_cast<const h8*>(sR + (c % X3_RING) * X3_CHUNK + (plane * 4 + (S4 ? 2 * wn + j : j)) * UNIT + lane * 16);
        };
        h8 fa[4], fan[4], fb[2], fbn[2];
#pragma unroll
        for (int u = 0; u < 4; ++u) fa[u] = load_a(0, u), fan[u] = fa[u];
        fb[0] = load_b(0, 0, 0);
        fb[1] = load_b(0, 0, 1);
        fbn[0] = fb[0], fbn[1] = fb[1];
        // one chunk, as a generic lambda over compile-time (chunk, FAST): the 3x3 chunks run once, the branch chunks exist in TWO copies
        // picked per tile -- a run-time "fast" inside the chunk would put the operand preparation into blocks of its own in front of the
        // quarter's MFMAs instead of into their gaps
        auto chunk = [&](auto c_tag, auto fast_tag) __attribute__((always_inline)) {
            constexpr int c = decltype(c_tag)::v;
            constexpr bool FAST = decltype(fast_tag)::v != 0;
            // every wave writes its own mailbox word (no run-time guard inside a chunk: it would cut the dealt block); wave 0's is the ticket
            if (c == QW) *reinterpret_cast<volatile int*>(mailbox + wave * 4) = __builtin_amdgcn_readfirstlane(ticket);
            if (c == QR) {
                const int drawn = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(mailbox));
                next = queue ? band_lo + per + drawn : tile + per;
                // (a ticket outside [0, tiles left) ends the walk: counters that were NOT zero on entry must not send a block through
                //  2^31 tickets -- wrong results then, but no five-minute launch)
                has_next = queue ? (unsigned)drawn < (unsigned)(band_hi - band_lo - per) : next < band_hi;
                nx_tile = last ? (has_next ? next : tile) : tile;
                nx_live = last ? has_next : true;
            }
            // (branch chunks are requested and written to the ring whether or not the tile runs them: a run-time guard here would cut
            //  chunks 14.. into basic blocks and the dealt schedule with them; an unneeded chunk is 8 KiB from L2 into a free slot)
            if (c + NSET < NC) request_chunk(c + NSET);
            // After the last 3x3 chunk request (chunk 17, at the top of chunk 13): memory returns in order, so a tile-data request
            // (HBM, ~2.5 us) ahead of a weight chunk (L2) would hold the chunk back.  Residual rows / partial sums of THIS tile first
            // (the epilogue needs them); the halo of the NEXT tile is requested behind the loop.
            if (c == RQR) {
#pragma unroll
                for (int i = 0; i < EIT; ++i) {           // the wave's pixel p = ep + PPI i: row p >> 4, column p & 15
                    const int p = ep + PPI * i, gx = tx0 + (p & 15);
                    const unsigned ro = ((unsigned)(ty0 + 2 * wrow + (p >> 4)) * (unsigned)W + (unsigned)gx) * 256u +
                                        (unsigned)wn * 128u + (unsigned)ec * 16u;
                    if (PAR) res4[i] = (f32x4)(0.f);
                    else res4[i] = buf_load4(r_res, (last & (gx < W)) ? ro : OOB);     // (not the last pass: out of range, zeros)
                }
            }
            // the next tile's halo: requested where registers are free -- the weight sets of chunks 14..17 are dead once written to
            // the ring (two chunks ahead), so rows [0,4) can leave at chunk 14, [4,6) at 15, [6,8) at 16; the rest follows behind the
            // loop.  (All of it behind the loop: the epilogue's stores queue behind 20 loads in the CU's memory pipe, +2 k cycles.)
            if (!PAR && c >= 14 && c <= 16) request_tile_part(nx_tile, nx_live, c - 14);
            h8 xa[4] = {fa[0], fa[1], fa[2], fa[3]};
            if (PAR && c >= 18) {
                if (c == 18) fold(true);                   // (conv + bias) * gamma BEFORE the 1x1 partition branches
                const int bi = bsel((c - 18) >> 1);
                // the branch's plane at the lane's two pixels, picked by ARITHMETIC (x 1 / x 0 on wave-uniform factors: exact for
                // finite maps): a select of vector registers on a uniform condition is lowered to branches, which cut the chunk into
                // basic blocks and the dealt schedule with them
                const float m0 = bi == 0 ? 1.f : 0.f, m1 = bi == 1 ? 1.f : 0.f, m2 = bi == 2 ? 1.f : 0.f;
                const float pjr[2] = {pv[0][0] * m0 + pv[1][0] * m1 + pv[2][0] * m2, pv[0][1] * m0 + pv[1][1] * m1 + pv[2][1] * m2};
                // par_j(pixel) * x as a split number again: (hi + lo / 2048) is exact in fp32 (22 bits), one fp32 rounding for the
                // product, then the same split as the halo (saturating as a whole)
                if (FAST) {
                    // pj is 0 or 1/255 and the weights carry the 1/255: the operand is x or nothing.  As a bit mask (one v_cndmask + 8
                    // v_and per row): a per-lane select on the fragments compiles to divergent branches, which cut the chunk into
                    // basic blocks and the dealt schedule with it (measured: K loop 24.5 k -> 23.3 k instead of -> 17 k)
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const unsigned mk = pjr[r] != 0.f ? 0xffffffffu : 0u;
                        xa[r] = __builtin_bit_cast(h8, __builtin_bit_cast(u32x4, fa[r]) & mk);
                        xa[2 + r] = __builtin_bit_cast(h8, __builtin_bit_cast(u32x4, fa[2 + r]) & mk);
                    }
                } else {
                    // general maps: par_j(pixel) * x as a split number again -- (hi + lo / 2048) is exact in fp32 (22 bits), one fp32
                    // rounding for the product, then the same split as the halo (saturating as a whole).  SCALAR arithmetic and the file
                    // built with -fno-slp-vectorize: the conservative form.  An intermediate r04 build (fast / general chosen inside the
                    // chunk) with this formula on float vectors gave run-to-run varying results; in this structure all four forms are
                    // bit-stable (tools/repro/f16x3_resplit_hazard.py, DESIGN.md 3.6 finding 5).
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const float pj = pjr[r];
                        h8 nh, nl;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float v = ((float)fa[r][e] + (float)fa[2 + r][e] * X3_INV) * pj;
                            v = fminf(fmaxf(v, -65504.f), 65504.f);
                            const _Float16 hh = (_Float16)v;
                            nh[e] = hh;
                            nl[e] = (_Float16)((v - (float)hh) * X3_SCALE);
                        }
                        xa[r] = nh;
                        xa[2 + r] = nl;
                    }
                }
            }
            const bool more = c + 1 < NC;                   // (a branch tile that stops early reads a stale ring slot: unused)
#pragma unroll
            for (int q = 0; q < NTW; ++q) {
                // quarter q: N tile q of the chunk -- 6 MFMAs; meanwhile the (hi, lo) B pair of the next quarter and 1 (S4: 2) of the
                // next chunk's 4 A fragments are fetched, DEALT into the MFMA gaps: a wave issues in order, a burst of 1-KiB reads in
                // front of the MFMAs holds its issue slot while only a partner wave can feed the matrix pipe (a lone wave then runs
                // 48-60 cycles per 32-cycle MFMA), one read per gap hides inside it (tools/ubench/ub_mfma_issue.hip).
                const int cn = q + 1 < NTW ? c : c + 1, qn = q + 1 < NTW ? q + 1 : 0;
                if (q + 1 < NTW || more) {
                    fbn[0] = load_b(cn, qn, 0);
                    fbn[1] = load_b(cn, qn, 1);
                }
                if (more) {
#pragma unroll
                    for (int u = q * (4 / NTW); u < (q + 1) * (4 / NTW); ++u) fan[u] = load_a(c + 1, u);
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) acc_hi[r][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[r], fb[0], acc_hi[r][q], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; ++r) acc_lo[r][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[2 + r], fb[0], acc_lo[r][q], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; ++r) acc_lo[r][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[r], fb[1], acc_lo[r][q], 0, 0, 0);
                if (q == (S4 ? 0 : 1) && c + 2 < NC) {
                    // ring write of chunk c + 2 (into the slot of chunk c - 1, which every wave left before the previous barrier)
                    // in the MIDDLE of the chunk: the barrier below then waits for it, not for the fragment reads behind it
                    char* d = sR + ((c + 2) % X3_RING) * X3_CHUNK;
#pragma unroll
                    for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(d + (t + 256 * i) * 16) = wreg[(c + 2) % NSET][i];
                }
#define X3_GAP(NREAD, NVMEM, NWRITE)                                                              \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      /* one MFMA */                \
                if (NREAD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   /* a fragment read */ \
                if (NVMEM) __builtin_amdgcn_sched_group_barrier(0x020, NVMEM, 0);  /* memory requests */ \
                if (NWRITE) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   /* ring write */
                if (S4) {                   // 4 reads
                    X3_GAP(1, 4, 0) X3_GAP(1, 4, 0) X3_GAP(1, 4, 0) X3_GAP(1, 4, 0) X3_GAP(0, 0, 1) X3_GAP(0, 0, 1)
                } else {                    // 3 reads
                    X3_GAP(1, 3, 0) X3_GAP(0, 3, 0) X3_GAP(1, 3, 0) X3_GAP(0, 3, 1) X3_GAP(1, 0, 0) X3_GAP(0, 0, 1)
                }
#undef X3_GAP
                __builtin_amdgcn_sched_barrier(0);
                fb[0] = fbn[0], fb[1] = fbn[1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) fa[u] = fan[u];
            // LDS operations complete in order: the fragment reads of the quarters behind the ring write (non-S4: quarters 2, 3 = 6
            // reads; S4: quarter 1 = 4) were issued after it, so "at most that many outstanding" means the write has landed -- and
            // those reads stay in flight across the barrier
            if (c + 1 >= NC) lds_barrier();
            else if (S4) asm volatile("s_waitcnt lgkmcnt(4)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(6)\n\ts_barrier" ::: "memory");
        };
        x3_static_for(std::make_integer_sequence<int, 18>{}, [&](auto c_tag) __attribute__((always_inline)) { chunk(c_tag, X3Tag<0>{}); });
        if constexpr (PAR) {
            auto branches = [&](auto fast_tag) __attribute__((always_inline)) {        // ncr = 18 + 2 x (branches this tile runs)
                if (ncr > 18) {
                    chunk(X3Tag<18>{}, fast_tag);
                    chunk(X3Tag<19>{}, fast_tag);
                    if (ncr > 20) {
                        chunk(X3Tag<20>{}, fast_tag);
                        chunk(X3Tag<21>{}, fast_tag);
                        if (ncr > 22) {
                            chunk(X3Tag<22>{}, fast_tag);
                            chunk(X3Tag<23>{}, fast_tag);
                        }
                    }
                }
            };
            if (fast) branches(X3Tag<1>{});
            else branches(X3Tag<0>{});
        }
        if (DBG) dbg_c = __builtin_amdgcn_s_memtime();
        // the next halo only now: inside the loop its 48 registers do not fit beside the fragments of the 16x16x32 pipeline (23 spills),
        // and the epilogue is long enough to cover the latency (the branch variant always requested it here: same prologue time)
        if (PAR) request_tile(has_next ? next : tile, has_next);
        else request_tile_part(nx_tile, nx_live, 3);
        if (MS) r_w = make_rsrc(uniform_ptr(last ? a.w : (pass == 0 ? a.w2[0] : a.w2[1])), 18u * X3_CHUNK);        // the next pass's weights
        request_first_chunks(has_lr && last);     // every set is free again; the latency hides behind the epilogue (or the next pass's split)
        if (last) break;
        }                                // pass loop
        fold(!PAR || ncr == 18);         // PAR with branches: bias / gamma went in before them; otherwise here

        // ---- epilogue: transpose through the dead A tile, [+ partial sum], activation, [+ residual], whole pixel rows to HBM
        // accumulator element e of lane (lm, lg), M tile r, N tile j = pixel (row r, column 4 lg + e), channel 16 j + lm of the wave
        // pixel stride 16 NTW + 4 floats: a write instruction's four lane groups (pixels 4 lg + e) then start 16 banks apart -- with the
        // bare stride all four met in the same 16 banks (SQ_LDS_BANK_CONFLICT per 720p launch: 1.84e6 -> 9.2e5)
        constexpr int TPS = 16 * NTW + 4;
        static_assert(4 * 32 * TPS * 4 <= ROWS * XRSB, "the transposition slices fit the dead A tile");
        float* sT = reinterpret_cast<float*>(smem + wave * (32 * TPS * 4));      // [32 pixels][16 NTW channels + 4] fp32 per wave
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) sT[(16 * r + 4 * lg + e) * TPS + j * 16 + lm] = acc_hi[r][j][e];
        asm volatile("" ::: "memory");
        f32x4 rows[EIT];
#pragma unroll
        for (int i = 0; i < EIT; ++i) rows[i] = *reinterpret_cast<const f32x4*>(sT + (ep + PPI * i) * TPS + ec * 4);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < EIT; ++i) {
            f32x4 v = rows[i] + k_pre * res4[i];
            v = __builtin_elementwise_max(v, (f32x4)(0.f)) + neg_slope * __builtin_elementwise_min(v, (f32x4)(0.f));
            v += k_post * res4[i];
            const int p = ep + PPI * i, gx = tx0 + (p & 15);
            const unsigned o = (unsigned)(ty0 + 2 * wrow + (p >> 4)) * o_sy + (unsigned)gx * o_sx + a.o_c0 + (unsigned)wn * 128u + (unsigned)ec * 16u;
            buf_store4(r_out, gx < W ? o : OOB, v);
        }
        if (DBG) {
            dbg_p += dbg_b - dbg_a;
            dbg_k += dbg_c - dbg_b;
            dbg_e += __builtin_amdgcn_s_memtime() - dbg_c;
            ++dbg_n;
        }
        if (!has_next) break;
        tile = next;
        lds_barrier();                   // the transposition rows are read: the next tile may overwrite the A tile
    }
    leave_queue();
    if (DBG && t == 0) {
        unsigned long long* d = a.dbg + (size_t)blockIdx.x * 8;
        d[0] = dbg_t0;
        d[1] = dbg_p;
        d[2] = dbg_k;
        d[3] = __builtin_amdgcn_s_memtime();
        d[4] = dbg_e;
        d[5] = dbg_n;
        d[6] = __builtin_amdgcn_s_memrealtime() - dbg_r0;       // 100 MHz
        d[7] = __builtin_amdgcn_s_getreg(6 | (31 << 11));       // HW_REG_LDS_ALLOC: LDS base 0 = the block that reached its CU first
    }
}

// fp32 B image -> the split image: per 64-deep chunk of the fp32 image (a tap / a 1x1 branch) two 8 KiB chunks (k-halves = one
// 32-deep k-step of v_mfma_f32_16x16x32_f16) of [hi N0..N3, lo N0..N3] fragment units; unit = 64 lanes x 8 halfs, lane (n, g) holds
// input channels 32 kh + 8 g .. + 7 of output channel 16 j + n; hi = fp16(w) (saturating), lo = fp16((w - hi) * 2048)
__global__ __launch_bounds__(256) void f16x3_image_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one (input channel, output channel) of a 64-deep chunk
    if (idx >= total) return;
    constexpr int per_chunk = PNP_CHUNK_Q * 2 * 256;                   // 4096
    const long chunk = idx / per_chunk;
    const int rem = (int)(idx - chunk * per_chunk);
    const int jj = rem & 7, lane = (rem >> 3) & 63, j = (rem >> 9) & 3, kh = rem >> 11;
    const int n = lane & 15, g = lane >> 4;
    const int k = 32 * kh + 8 * g + jj, co = 16 * j + n;
    const float v = src[chunk * per_chunk + (((k >> 3) * 2 + (co >> 5)) * 64 + ((k >> 2) & 1) * 32 + (co & 31)) * 4 + (k & 3)];
    const float vc = fminf(fmaxf(v, -65504.f), 65504.f);         // saturates as a whole (see the halo split)
    const _Float16 hi = (_Float16)vc;
    const _Float16 lo = (_Float16)((vc - (float)hi) * X3_SCALE);
    _Float16* d = dst + chunk * (2 * per_chunk) + kh * per_chunk + j * 512 + lane * 8 + jj;
    d[0] = hi;
    d[4 * 512] = lo;
}

template <bool PAR, bool DBG, bool S4, bool MS = false>
int launch_x3_t(const X3Args& xa, hipStream_t stream) {
    auto kern = conv3x3_f16x3_kernel<PAR, DBG, S4, MS>;
    constexpr int lds = (S4 ? 4 + 2 : TH + 2) * XRSB + XPARK + (MS ? (S4 ? 4 + 2 : TH + 2) * PW * 16 : 0) + X3_RING * X3_CHUNK;
    static_assert(2 * lds <= 160 * 1024, "two blocks per CU");
    static PnpPerDevice once;
    const hipError_t attr_err = once.run([&](int, int&) {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    if (attr_err != hipSuccess) return (int)attr_err;
    const int th = S4 ? 4 : TH;
    const int tiles = ((xa.W + TW - 1) / TW) * ((xa.H + th - 1) / th);
    int per_xcd = (tiles + 7) / 8;                // blocks per XCD: two per CU at most (32 CUs), one tile each on small frames
    if (per_xcd > 64) per_xcd = 64;
    X3Args xq = xa;
    if (tiles <= 8 * per_xcd) xq.queue = nullptr;       // one tile per block: nothing to hand out (128x128: 2870 -> 2230 frames/s with it)
    hipLaunchKernelGGL(kern, dim3(8 * per_xcd), dim3(256), lds, stream, xq);
    return (int)hipGetLastError();
}

// frames with at most one 8x16 tile per CU run on 4x16 tiles -- twice the blocks, half the work each.  Measured: 128x128 (128
// tiles) 2240 -> 2888 frames/s; 180x320 (460 tiles, three clips) loses 9 % on them and stays on 8x16.
template <bool PAR, bool DBG, bool MS = false>
int launch_x3(const X3Args& xa, hipStream_t stream) {
    const int tiles8 = ((xa.W + TW - 1) / TW) * ((xa.H + TH - 1) / TH);
    return tiles8 <= 256 ? launch_x3_t<PAR, DBG, true, MS>(xa, stream) : launch_x3_t<PAR, DBG, false, MS>(xa, stream);
}

}  // namespace

int launch_f16x3_image(const float* src, void* dst, int nchunks, hipStream_t stream) {
    if (nchunks < 1) return PNP_ERR_BAD_ARG;
    const long total = (long)nchunks * pnp_chunk_floats(2);
    hipLaunchKernelGGL(f16x3_image_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src,
                       reinterpret_cast<_Float16*>(dst), total);
    return (int)hipGetLastError();
}

// A conv over several sources runs as a chain that accumulates through `out` (fp32 partial sums, added before the activation
// of the last link): the RGB frame first, on the exact fp32 kernel, then one split launch per 64-channel source.
// A pixel-shuffle conv (out_mode 1, upsample.py:49-50) is four launches, one per sub-pixel weight image, each scattering its 64
// channels to pixel (2y + dy, 2x + dx) of the 2H x 2W map.
int launch_conv3x3_f16x3(const ConvArgs& a, int cfg, hipStream_t stream) {
    if (a.out_mode == 1 || a.out_mode == 4) {
        // out_mode 4 (the DCN aligner's conv_offset[2], iconvsr_mv.py:30-36): grid_y blocks of 64 output channels into a map of
        // out_cstride channels per pixel
        const int ny = a.out_mode == 1 ? 4 : a.out_cstride / 64;
        const unsigned pix = (unsigned)a.out_cstride * 4u;
        for (int y = 0; y < ny; ++y) {
            X3Args x;
            memset(&x, 0, sizeof(x));
            x.nsrc = 1;
            x.src = a.src[0];
            x.w = reinterpret_cast<const _Float16*>(a.wsrc_h[0]) + 2 * (long)y * a.w_ystride;     // split image: 2 halfs per float
            x.wpar = nullptr;
            x.par = nullptr;
            x.par_plane = 0;
            x.par_flags = nullptr;
            x.bias = a.bias ? a.bias + y * a.bias_ystride : nullptr;
            x.gamma = nullptr;
            x.residual = nullptr;
            x.res_pre = 0;
            x.out = a.out;
            x.H = a.H;
            x.W = a.W;
            x.act = a.act;
            x.queue = a.tile_queue;
            x.dbg = nullptr;
            if (a.out_mode == 1) {
                x.o_sy = (unsigned)a.W * 1024u;
                x.o_sx = 512u;
                x.o_c0 = (unsigned)(y >> 1) * (unsigned)a.W * 512u + (unsigned)(y & 1) * 256u;
                x.out_bytes = (unsigned)a.H * (unsigned)a.W * 1024u;
            } else {
                x.o_sy = (unsigned)a.W * pix;
                x.o_sx = pix;
                x.o_c0 = 256u * (unsigned)y;
                x.out_bytes = (unsigned)a.H * (unsigned)a.W * pix;
            }
            const int rc = launch_x3<false, false>(x, stream);
            if (rc) return rc;
        }
        return PNP_OK;
    }
    int lr_idx = -1, wide[4], nwide = 0;
    for (int s = 0; s < a.nsrc; ++s) {
        if (a.src_c[s] == 4) lr_idx = s;
        else wide[nwide++] = s;
    }
    bool have_partial = false;
    const bool fold_lr = lr_idx >= 0 && a.wsrc_h[lr_idx] && !a.dbg;       // the RGB frame inside the MS launch (its split image exists)
    if (lr_idx >= 0 && !fold_lr) {
        ConvArgs r = a;
        r.prec = 0;
        r.nsrc = 1;
        r.src[0] = a.src[lr_idx];
        r.src_c[0] = 4;
        r.wsrc[0] = a.wsrc[lr_idx];
        r.act = 0;
        r.residual = nullptr;
        r.dbg = nullptr;          // the fp32 tile kernel's trace is 16 u64 per TILE; pnp_conv3x3_f16x3_ex's buffer holds 8 per split block
        const int rc = launch_conv3x3(r, cfg, 1, stream);
        if (rc) return rc;
        have_partial = true;
    }
    if ((nwide > 1 || fold_lr) && !a.dbg) {
        // all 64-channel sources in ONE launch (MS kernel): one K = nwide x 576 contraction per tile, no partial sums through HBM
        // between the sources; the RGB link's partial sums (if any) come in as the pre-activation residual
        X3Args x;
        memset(&x, 0, sizeof(x));
        x.nsrc = nwide;
        x.src = a.src[wide[0]];
        x.w = reinterpret_cast<const _Float16*>(a.wsrc_h[wide[0]]);
        for (int k = 1; k < nwide; ++k) {
            x.src2[k - 1] = a.src[wide[k]];
            x.w2[k - 1] = reinterpret_cast<const _Float16*>(a.wsrc_h[wide[k]]);
        }
        if (fold_lr) {
            x.lr4 = a.src[lr_idx];
            x.wlr = reinterpret_cast<const _Float16*>(a.wsrc_h[lr_idx]);
        }
        x.bias = have_partial ? nullptr : a.bias;
        x.residual = have_partial ? a.out : (nwide == 1 ? a.residual : nullptr);
        x.res_pre = have_partial ? 1 : 0;
        x.out = a.out;
        x.H = a.H;
        x.W = a.W;
        x.act = a.act;
        x.queue = a.tile_queue;
        x.o_sy = (unsigned)a.W * 256u;
        x.o_sx = 256u;
        x.o_c0 = 0;
        x.out_bytes = (unsigned)a.H * (unsigned)a.W * 256u;
        return launch_x3<false, false, true>(x, stream);
    }
    for (int k = 0; k < nwide; ++k) {
        const bool last = k == nwide - 1;
        X3Args x;
        memset(&x, 0, sizeof(x));
        x.nsrc = 1;
        x.src = a.src[wide[k]];
        x.w = reinterpret_cast<const _Float16*>(a.wsrc_h[wide[k]]);
        x.wpar = reinterpret_cast<const _Float16*>(a.wpar_h);
        x.wpar_scaled = a.wpar_h_scaled;
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
        x.queue = a.tile_queue;
        x.dbg = a.dbg;
        x.o_sy = (unsigned)a.W * 256u;
        x.o_sx = 256u;
        x.o_c0 = 0;
        x.out_bytes = (unsigned)a.H * (unsigned)a.W * 256u;
        const int rc = x.wpar ? (x.dbg ? launch_x3<true, true>(x, stream) : launch_x3<true, false>(x, stream))
                              : (x.dbg ? launch_x3<false, true>(x, stream) : launch_x3<false, false>(x, stream));
        if (rc) return rc;
        have_partial = true;
    }
    return PNP_OK;
}
