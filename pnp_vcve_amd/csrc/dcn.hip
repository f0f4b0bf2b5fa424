// Flow-guided modulated deformable alignment: the reference's optional aligners
// BasiceformableAlignment / FVCDeformableAlignment (mmedit/models/backbones/sr_backbones/iconvsr_mv.py:21-84)
// call mmcv.ops.modulated_deform_conv2d(x, offset, mask, W(64,64,3,3), bias, stride 1, pad 1, dil 1,
// groups 1, deform_groups 16).  mmcv-full is not vendored in the reference: the arithmetic below restates
// its published semantics (modulated_deformable_im2col + dmcn_im2col_bilinear: a sample is taken only if
// -1 < h_im < H and -1 < w_im < W, each of the four corners contributes only if it lies inside the image;
// out = W . (mask * sample) + bias) -- PARITY UNPINNED against mmcv itself, checked against oracle/cpu_ref.py
// (whose degenerate cases are pinned to ATen ops, tests/test_oracle_dcn.py).
//
// Structure (round 2).  The deformable im2col never exists in memory, and the 9 x 16 x 4 corner reads per pixel no
// longer go to global memory:
//   * LDS WINDOWS.  A block owns an 8x16 pixel tile = two 8x8 halves.  Codec motion is constant over >= 8x8
//     partitions, so each half gets its own 16x16 pixel window of `x` (fp32, 256 B + 16 B pad per pixel, 68 KiB),
//     positioned at the half's origin + its rounded flow - 4: it holds every corner of every sample whose learned
//     offset stays within about +-2.5 px of the flow.  A corner read is then ONE ds_read_b128; a sample whose corners
//     leave the window falls back to the same arithmetic on global memory (exact same result, wave-uniform branch).
//   * NO GATHER BUFFER.  Lane (m, h) of a wave is MFMA row m / k-half h, and v_mfma_f32_32x32x2_f32 consumes, at
//     q-step q, channels 8q + 4h .. +3 of pixel m = the four channels of deform group 2q + h.  So the lane gathers
//     exactly the (pixel, group) samples its own A operand needs -- 8 per tap -- and feeds them to the MFMAs from
//     registers.  The conv_offset output is laid out for that (prep.h, pnp_dcn_ref_channel_impl): a lane's 8
//     (dy, dx) pairs of a tap are 64 contiguous bytes, its 8 mask logits 32.
//   * TWO WAVES PER SIMD IN ANTI-PHASE.  A persistent 512-thread block per CU: waves w and w + 4 work on the same 32
//     pixels, each contracting half of every tap's K (q-steps 0-3 / 4-7 = deform groups 0-7 / 8-15); in every tap
//     interval one gathers tap k+1 and then contracts tap k, the other contracts first.  The offsets of tap k+2 and the
//     weights of tap k+1 are in flight meanwhile; the two partial tiles are summed in the epilogue.
//   * WHAT BOUNDS IT (tools/bench_dcn.py timeline, 720p): per tile 37 k cycles of fp32 MFMA (the 64x576 contraction,
//     0.5 ms per call on its own) + ~25 k cycles of gather arithmetic + 15 k of window fill / weight hand-over / epilogue.
//     v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate: the gather's ~55 VALU per sample and the fp32 MFMAs of the
//     partner wave do not overlap (measured: a wave's 250-instruction gather takes 2.5 k cycles beside the partner's
//     MFMAs and slows those from 64 to 120 cycles each), so the two add up.  1.19 ms per 720p call (was 1.89 ms).
#include "dcn.h"
#include <mutex>

namespace {

constexpr int WIN = 16;                  // window side: 8 + 2 * WR
constexpr int WR = 4;
constexpr int PSTR = 17;                 // float4 per window pixel (256 B + 16 B pad: 16 consecutive pixels tile the 64 banks)
constexpr int WIN_F4 = WIN * WIN * PSTR;
constexpr int CH4 = PNP_CHUNK_Q * 2 * 64;
constexpr int WIN1 = WIN_F4 + 8;          // window 1 starts 8 slots (32 banks) later: the two halves of a pixel row land on disjoint banks
constexpr int LDS_BYTES = (WIN1 + WIN_F4 + CH4) * 16;      // 155,776
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(2 * WIN_F4 >= 8 * 32 * 16, "the epilogue transposes 8 x 8 KiB through the window region");

// sigmoid of the mask logit (torch.sigmoid, iconvsr_mv.py:79) on the hardware exp / rcp (1-2 ulp)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// One (pixel, deform group) sample read from GLOBAL memory: mmcv's dmcn_im2col_bilinear verbatim (corner by corner
// validity).  Rare path (a corner outside the half's LDS window).
__device__ __forceinline__ f32x4 sample_global(const f32x4* __restrict__ x4, float hi, float wi, float mask, int g,
                                                         int H, int W) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (!(hi > -1.f && wi > -1.f && hi < (float)H && wi < (float)W)) return z;
    const float hl = floorf(hi), wl = floorf(wi);
    const int h0 = (int)hl, w0 = (int)wl, h1 = h0 + 1, w1 = w0 + 1;
    const float lh = hi - hl, lw = wi - wl, hh = 1.f - lh, hw = 1.f - lw;
    const long b = ((long)h0 * W + w0) * 16 + g;
    const f32x4 v1 = (h0 >= 0 && w0 >= 0) ? x4[b] : z;
    const f32x4 v2 = (h0 >= 0 && w1 <= W - 1) ? x4[b + 16] : z;
    const f32x4 v3 = (h1 <= H - 1 && w0 >= 0) ? x4[b + (long)W * 16] : z;
    const f32x4 v4 = (h1 <= H - 1 && w1 <= W - 1) ? x4[b + (long)W * 16 + 16] : z;
    return (v1 * (hh * hw) + v2 * (hh * lw) + v3 * (lh * hw) + v4 * (lh * lw)) * mask;
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// F16 (PNP_PREC_F16): the gathered samples and the weights are rounded to fp16 as MFMA operands
// (v_mfma_f32_32x32x16_f16, fp32 accumulation), like every other 64-channel contraction of that mode.  The fp16 matrix
// pipe is separate from the vector ALUs, so here the partner wave's gather does run under the MFMAs.
template <bool F16>
__global__ __launch_bounds__(512) void dcn_window_kernel(const DcnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* sW = reinterpret_cast<f32x4*>(smem_raw);      // two 16x16 pixel windows of x
    f32x4* sB = sW + WIN1 + WIN_F4;                       // weights of the current tap
    // 8 waves = two groups of 4 on the SAME 128 pixels: group A (waves 0-3) contracts q-steps 0..3 of every tap
    // (deform groups 0..7), group B (waves 4-7) q-steps 4..7; the two partial tiles are summed at the end.
    const int t = threadIdx.x, lane = t & 63;
    const int grp = __builtin_amdgcn_readfirstlane(t >> 8), wave = __builtin_amdgcn_readfirstlane((t >> 6) & 3);
    const int m = lane & 31, h = lane >> 5;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + 15) / 16;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x);
    const f32x4* wimg = reinterpret_cast<const f32x4*>(F16 ? a.w16 : (const void*)a.w);
    constexpr int WCH4 = F16 ? CH4 / 2 : CH4;     // float4 per tap of the weight image (16 KiB fp32, 8 KiB fp16)
    // Persistent: one block per CU (LDS-limited) walks a strip of tiles.  XCD x (blocks with blockIdx.x % 8 == x) owns a
    // contiguous band of tiles, dealt round-robin to its blocks, so neighbouring windows meet in that XCD's L2.
    const int ntiles = tiles_x * ((H + 7) / 8);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int bq = ntiles >> 3, br = ntiles & 7;
    const int xbeg = xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq;
    const int xend = xbeg + bq + (xcd < br ? 1 : 0);
    unsigned long long d_fill = 0, d_pro = 0, d_g = 0, d_m = 0, d_b = 0, d_epi = 0, d_t0 = 0, d_x = 0;
    int d_n = 0;
    auto stamp = [&](unsigned long long& acc_) {     // adds the time since the previous stamp to acc_
        if (a.dbg) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            acc_ += now - d_x;
            d_x = now;
        }
    };
    if (a.dbg) d_t0 = d_x = __builtin_amdgcn_s_memtime();
    for (int tile = xbeg + slot; tile < xend; tile += nslots) {
    const int ty0 = (tile / tiles_x) * 8, tx0 = (tile % tiles_x) * 16;

    // ---- window origins: half s = columns tx0 + 8 s .. + 7; its window starts at (origin + rounded flow - WR)
    int oy[2], ox[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int by = 0, bx = 0;
        if (a.fx) {
            const long c = (long)min(ty0, H - 1) * W + min(tx0 + 8 * s, W - 1);
            by = (int)floorf(a.fy[c] + 0.5f);
            bx = (int)floorf(a.fx[c] + 0.5f);
        }
        oy[s] = __builtin_amdgcn_readfirstlane(ty0 + by - WR);
        ox[s] = __builtin_amdgcn_readfirstlane(tx0 + 8 * s + bx - WR);
    }
    // ---- fill the windows: 2 x 256 pixels x 16 float4, 16 consecutive lanes = one pixel (256 B); 8 loads per thread in
    //      flight at a time (64 KiB per CU)
#pragma unroll 1
    for (int jb = 0; jb < 16; jb += 8) {
        f32x4 wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = t + 512 * (jb + j);
            const int slot = i & 15, wp = i >> 4, s = wp >> 8, wq = wp & 255;
            const int iy = oy[s] + (wq >> 4), ix = ox[s] + (wq & 15);
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            wv[j] = ok ? x4[((long)iy * W + ix) * 16 + slot] : z;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = t + 512 * (jb + j);
            const int slot = i & 15, wp = i >> 4, s = wp >> 8, wq = wp & 255;
            sW[s * WIN1 + wq * PSTR + slot] = wv[j];
        }
    }

    stamp(d_fill);
    // ---- this lane's pixel, its offset record and its window
    const int py = 2 * wave + (m >> 4), px = m & 15;
    const int gy = ty0 + py, gx = tx0 + px;
    const bool pin = gy < H && gx < W;
    const long pidx = pin ? (long)gy * W + gx : 0;
    const f32x4* rec = reinterpret_cast<const f32x4*>(a.om + pidx * 448);
    float fyv = 0.f, fxv = 0.f;
    if (a.fx) {                                  // 'basic': offset + flow.flip(1) (iconvsr_mv.py:77)
        fyv = a.fy[pidx];
        fxv = a.fx[pidx];
    }
    const int half = px >> 3;
    const f32x4* win = sW + half * WIN1;
    const int woy = oy[half], wox = ox[half];

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // offsets (2 float4 = 4 (dy, dx) pairs) and mask logits (1 float4) of this lane's 4 samples of one tap
    auto load_om = [&](int k, f32x4* o, f32x4& mk) {
        o[0] = rec[k * 8 + h * 4 + grp * 2];
        o[1] = rec[k * 8 + h * 4 + grp * 2 + 1];
        mk = rec[72 + k * 4 + h * 2 + grp];
    };
    // The 4 samples of a tap.  Window pixels outside the image hold zeros, so inside the window mmcv's per-corner validity
    // tests are implied (an invalid corner contributes 0 * weight) and a sample is four ds_read_b128 + 20 VALU.  If any
    // lane of the WAVE has a sample with a corner outside its window (wave-uniform test), the wave redoes the tap's
    // out-of-window samples from global memory.
    auto gather = [&](int ky, int kx, const f32x4* o, const f32x4& mk, f32x4* out) {
        float hi[4], wi[4], mask[4];
        int outside = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            hi[q] = (float)(gy + ky) + (o[q >> 1][(q & 1) * 2] + fyv);
            wi[q] = (float)(gx + kx) + (o[q >> 1][(q & 1) * 2 + 1] + fxv);
            mask[q] = pin ? fast_sigmoid(mk[q]) : 0.f;
            const float hl = floorf(hi[q]), wl = floorf(wi[q]);
            const float lh = hi[q] - hl, lw = wi[q] - wl, hh = 1.f - lh, hw = 1.f - lw;
            const int ry = (int)hl - woy, rx = (int)wl - wox;
            const bool inw = (unsigned)ry <= (unsigned)(WIN - 2) && (unsigned)rx <= (unsigned)(WIN - 2);
            // a sample that is not taken at all (outside (-1, H) x (-1, W)) has every corner outside the image: zeros
            const bool far = !(hi[q] > -1.f && wi[q] > -1.f && hi[q] < (float)H && wi[q] < (float)W);
            outside |= (!inw && !far && pin) ? (1 << q) : 0;
            const f32x4* p = win + ((inw ? (ry * WIN + rx) * PSTR : 0) + 2 * (4 * grp + q) + h);
            const float m = (inw && !far) ? mask[q] : 0.f;
            out[q] = (p[0] * (hh * hw) + p[PSTR] * (hh * lw) + p[WIN * PSTR] * (lh * hw) + p[WIN * PSTR + PSTR] * (lh * lw)) * m;
        }
        if (__any(outside)) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (outside & (1 << q)) out[q] = sample_global(x4, hi[q], wi[q], mask[q], 2 * (4 * grp + q) + h, H, W);
        }
    };
    auto mfma_tap = [&](const f32x4* av) {
        if (F16) {
            // k-step s' of the fp16 image pairs q-steps 2s', 2s'+1 (dcn_f16_image_kernel): lane (m, h) supplies the 4 + 4
            // channels of its two samples; this wave's k-steps are 2 grp, 2 grp + 1
            const f32x4* bb = sB + lane + grp * (4 * 64);
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const h4 lo = __builtin_convertvector(av[2 * sp], h4), hi = __builtin_convertvector(av[2 * sp + 1], h4);
                const h8 af = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                const h8 b0 = __builtin_bit_cast(h8, bb[(sp * 2) * 64]), b1 = __builtin_bit_cast(h8, bb[(sp * 2 + 1) * 64]);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b1, acc[1], 0, 0, 0);
            }
            return;
        }
        const f32x4* bb = sB + lane + grp * (8 * 64);       // q-steps 4 grp .. 4 grp + 3 of the tap's chunk
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b0 = bb[(q * 2) * 64], b1 = bb[(q * 2 + 1) * 64];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][kk], b0[kk], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][kk], b1[kk], acc[1], 0, 0, 0);
            }
        }
    };

    // ---- prologue: weights of tap 0 -> LDS, samples of tap 0 -> registers, offsets of tap 1 in flight
    constexpr int NB = WCH4 / 512;               // float4 of a tap's weights per thread
    f32x4 breg[NB], av[4], nav[4], oo[2], no[2], om, nm;
    load_om(0, oo, om);
#pragma unroll
    for (int i = 0; i < NB; ++i) breg[i] = wimg[t + 512 * i];
#pragma unroll
    for (int i = 0; i < NB; ++i) sB[t + 512 * i] = breg[i];
    __syncthreads();                             // windows + weights of tap 0 visible
    load_om(1, no, nm);
    gather(-1, -1, oo, om, av);
    stamp(d_pro);

    // Interval k: sB holds tap k's weights and `av` tap k's samples.  The two waves that share a SIMD (w, w + 4) run the
    // interval in opposite order -- group A gathers tap k+1 and then contracts tap k, group B contracts first -- so one
    // wave's LDS reads / vector arithmetic run under the other's MFMAs.
    int ky = -1, kx = 0;                         // tap k + 1 = (ky, kx)
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int i = 0; i < NB; ++i) breg[i] = wimg[(long)(k + 1) * WCH4 + t + 512 * i];
        oo[0] = no[0];
        oo[1] = no[1];
        om = nm;
        if (k < 7) load_om(k + 2, no, nm);
        // fp16: the contraction is 4 MFMAs per tap -- nothing to hide behind, so both groups gather first (compile-time choice).
        // Round 2 found the fp16 instantiation giving corrupted, run-to-run varying sums with the run-time two-armed order.
        // Round 3 (tools/repro/dcn_f16_hazard.py + four micro-repros, profiles/r03_dcn_hazard_report.txt): what gets corrupted is
        // the A operand -- gathered samples of lanes 16..31 / 48..63 -- not the accumulators; it needs the divergent global-memory
        // fallback of gather() to be present and hipcc's SLP packing of the scalar gather arithmetic into v_pk_*_f32 pairs; it is
        // timing dependent and also hits THIS code shape (rarely: 732 of 59 M elements in one of ~10 runs at 720p; always with
        // -amdgpu-waitcnt-forcezero).  Built with -fno-slp-vectorize (pnp_vcve_amd/build_native.py) every shape, including the
        // run-time order, is correct and bit-stable under every perturbation tried.  The hardware itself was cleared: MFMA
        // destination registers become readable 4 / 8 / 12 wait states after issue (dst[0] / [8] / [15]; hipcc waits 12), SrcA /
        // SrcB may be overwritten at once, SrcC (C != D) after 4, narrowing EXEC behind an MFMA is harmless, and VALU / trans /
        // packed / DPP / v_mov_b64 / LDS work beside MFMAs of the same or the partner wave is exact.
        if (F16 || grp == 0) {
            gather(ky, kx, oo, om, nav);
            stamp(d_g);
            mfma_tap(av);
            stamp(d_m);
        } else {
            mfma_tap(av);
            stamp(d_m);
            gather(ky, kx, oo, om, nav);
            stamp(d_g);
        }
        if (++kx > 1) {
            kx = -1;
            ++ky;
        }
        __syncthreads();                         // every wave is done with sB
#pragma unroll
        for (int i = 0; i < NB; ++i) sB[t + 512 * i] = breg[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = nav[q];
        __syncthreads();
        stamp(d_b);
    }
    mfma_tap(av);                                // tap 8
    stamp(d_m);
    __syncthreads();                             // every wave is done with the windows

    // ---- the two partial tiles (transposed to pixel rows) through the now free window region, summed + bias, stored
    const int n0 = lane & 31;
    float* sT = reinterpret_cast<float*>(sW) + (grp * 4 + wave) * (32 * 64);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + j * 32 + n0] = acc[j][r];
    __syncthreads();
    const f32x4* sA4 = reinterpret_cast<const f32x4*>(sW);
    const f32x4 bias4 = reinterpret_cast<const f32x4*>(a.bias)[t & 15];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = t + 512 * i;                       // float4 #e of the 128 x 16 tile: pixel e >> 4 (wave-major order)
        const int p = e >> 4, ec = e & 15;
        const int oyy = ty0 + (p >> 4), oxx = tx0 + (p & 15);
        const f32x4 v = (sA4[e] + sA4[e + 4 * 32 * 16]) + bias4;
        if (oyy < H && oxx < W) *reinterpret_cast<f32x4*>(a.out + ((long)oyy * W + oxx) * 64 + ec * 4) = v;
    }
    __syncthreads();                             // the partial tiles are read: the next tile may refill the windows
    stamp(d_epi);
    ++d_n;
    }   // strip
    if (a.dbg && lane == 0) {
        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 8 + (t >> 6)) * 8;
        d[0] = d_fill;
        d[1] = d_pro;
        d[2] = d_g;
        d[3] = d_m;
        d[4] = d_b;
        d[5] = d_epi;
        d[6] = __builtin_amdgcn_s_memtime() - d_t0;
        d[7] = d_n;
    }
}

// fp32 B image of deform_align.weight (9 chunks, common.h) -> the fp16 image dcn_window_kernel<true> contracts:
// 1-KiB units [tap][k-step s'][n-tile][lane (h, n)][8 halfs], half j' = fp32 element (q = 2 s' + (j' >> 2), h, j = j' & 3),
// i.e. input channel 8 q + 4 h + j: the k order in which a lane's two gathered samples (deform groups 2q+h) arrive.
__global__ __launch_bounds__(256) void dcn_f16_image_kernel(const float* __restrict__ src, _Float16* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // over 9 * 4096 halfs
    if (idx >= 9 * 4096) return;
    const int tap = idx >> 12, rem = idx & 4095;
    const int jp = rem & 7, lane = (rem >> 3) & 63, nt = (rem >> 9) & 1, sp = rem >> 10;
    const int n = lane & 31, hh = lane >> 5, q = 2 * sp + (jp >> 2), j = jp & 3;
    const float v = src[tap * 4096 + ((q * 2 + nt) * 64 + hh * 32 + n) * 4 + j];
    dst[idx] = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
}

}  // namespace

int launch_dcn_f16_image(const float* packed_w, void* dst, hipStream_t stream) {
    hipLaunchKernelGGL(dcn_f16_image_kernel, dim3(9 * 4096 / 256), dim3(256), 0, stream, packed_w,
                       reinterpret_cast<_Float16*>(dst));
    return (int)hipGetLastError();
}

// blocks of a launch on the current device (one resident block per CU) + the one-time kernel attributes
static hipError_t dcn_setup(int* grid_out) {
    static PnpPerDevice once;
    int grid = 256;
    const hipError_t attr_err = once.run([](int dev, int& g) {
        g = 256;
        (void)hipDeviceGetAttribute(&g, hipDeviceAttributeMultiprocessorCount, dev);      // one resident block per CU
        g -= g % 8;
        if (g < 8) g = 8;        // a device / partition exposing fewer than 8 CUs: the strips are dealt over blockIdx.x % 8
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_window_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_window_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        return e;
    }, &grid);
    *grid_out = grid;
    return attr_err;
}

int dcn_trace_u64s() {
    int grid = 0;
    return dcn_setup(&grid) == hipSuccess ? grid * 8 * 8 : -1;      // 8 u64 per wave, 8 waves per block
}

int launch_dcn(const DcnArgs& a, hipStream_t stream) {
    int grid = 256;
    const hipError_t attr_err = dcn_setup(&grid);
    if (attr_err != hipSuccess) return (int)attr_err;
    if (a.w16) hipLaunchKernelGGL(dcn_window_kernel<true>, dim3(grid), dim3(512), LDS_BYTES, stream, a);
    else hipLaunchKernelGGL(dcn_window_kernel<false>, dim3(grid), dim3(512), LDS_BYTES, stream, a);
    return (int)hipGetLastError();
}
