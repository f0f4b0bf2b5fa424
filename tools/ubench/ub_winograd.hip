// Upper bound for a Winograd F(2x2,3x3) 64 -> 64 channel 3x3 conv on the fp32 matrix pipe of MI355X.
//
// Structure measured (the K loop the real kernel would run, on L2/LDS-resident data):
//   block = 4 waves, ONE block per CU (256 accumulator registers per lane), block tile = 16x16 output pixels = 8x8 Winograd tiles.
//   wave w owns tile rows 2w, 2w+1 (16 tiles = the M of v_mfma_f32_16x16x4_f32) and ALL 64 output channels (4 N tiles) and all 16
//   transform positions: 16 x 4 accumulators of 4 registers.  Lane (m = lane & 15, kq = lane >> 4) is tile m, k-quarter kq.
//   K loop: 4 steps s of 16 channels (lane: channels 16 s + 4 kq + j, j = 0..3 = four MFMA k-steps); per step the lane reads its 4x4
//   patch (16 ds_read_b128), transforms it (32 float4 adds: V = B^T d B) and runs 16 positions x 4 N tiles x 4 k-steps = 256 MFMAs.
//   Transformed weights (256 KB per conv) stream L2 -> registers -> a 3-slot LDS ring in 16 chunks of 16 KB (chunk = step s, position
//   row i); the halo tile (18 x 18 x 64 ch = 81 KB) lives in LDS as four 16-channel slabs, and the NEXT tile's slab s replaces this
//   tile's slab s as soon as step s has read it (K-outer order: no second halo buffer).
//   Epilogue: output transform Y = A^T M A in registers, direct global stores.
// Modes: 0 full (ring + halo refill from global, barriers) | 1 static LDS (no global traffic) | 2 as 1 without the input transform
//        (VALU cost) | 3 MFMAs on register operands only (the ceiling of the loop shape).
// Prints effective direct-conv TFLOP/s (2 * 576 * 64 FLOP per pixel) next to the executed MFMA TFLOP/s; checks the result of one
// tile against a direct convolution on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HP = 18, NPIX = HP * HP, SLAB4 = NPIX * 4, CH4 = 1024;
constexpr int LDS_BYTES = (4 * SLAB4 + 3 * CH4) * 16;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int MODE>
__global__ __launch_bounds__(256, 1) void wino(const f32x4* __restrict__ src, const f32x4* __restrict__ U, float* __restrict__ out,
                                               int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    f32x4* sH = smem;
    f32x4* sB = smem + 4 * SLAB4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int ty = 2 * wave + (m >> 3), tx = m & 7;
    for (int i = t; i < 4 * SLAB4; i += 256) {
        const int s = i / SLAB4, r = i - s * SLAB4;
        sH[i] = src[(r >> 2) * 16 + s * 4 + (r & 3)];
    }
    for (int i = t; i < 2 * CH4; i += 256) sB[i] = U[i];
    __syncthreads();
    const f32x4* dbase = sH + ((2 * ty) * HP + 2 * tx) * 4 + kq;

    f32x4 acc[16][4];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Rolling input transform: V row i (positions 4i..4i+3) is consumed by chunk (s, i) and rewritten for step s+1 right after:
    //   chunk (s,0): row 3 of step s from d1, d3;  then read d rows 0, 2 of step s+1
    //   chunk (s,1): row 0 of step s+1 (d0 - d2);  read d row 1
    //   chunk (s,2): row 1 of step s+1 (d1 + d2);  read d row 3
    //   chunk (s,3): row 2 of step s+1 (d2 - d1)
    // 8 float4 adds per chunk of 64 MFMAs; live: V (64 registers) + at most 16 d float4.
    f32x4 V[16], d0[4], d1[4], d2[4], d3[4];
    auto read_row = [&](f32x4* dr, int s, int r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) dr[c] = dbase[s * SLAB4 + (r * HP + c) * 4];
    };
    auto col_tf = [&](f32x4* v, const f32x4* tt) {
        if (MODE == 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = tt[c];
            return;
        }
        v[0] = tt[0] - tt[2];
        v[1] = tt[1] + tt[2];
        v[2] = tt[2] - tt[1];
        v[3] = tt[1] - tt[3];
    };
    auto row_tf = [&](int i) {
        f32x4 tt[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (MODE == 2) tt[c] = i == 0 ? d0[c] : (i == 1 ? d1[c] : (i == 2 ? d2[c] : d3[c]));
            else tt[c] = i == 0 ? d0[c] - d2[c] : (i == 1 ? d1[c] + d2[c] : (i == 2 ? d2[c] - d1[c] : d1[c] - d3[c]));
        }
        col_tf(V + 4 * i, tt);
    };
    if (MODE == 3) {
#pragma unroll
        for (int i = 0; i < 16; ++i) V[i] = src[t + 256 * i];
    } else {
        read_row(d0, 0, 0);
        read_row(d1, 0, 1);
        read_row(d2, 0, 2);
        read_row(d3, 0, 3);
        row_tf(0);
        row_tf(1);
        row_tf(2);
    }

    int slot_r = 0, slot_w = 2, gchunk = 2;
    f32x4 breg[4], hreg[3];
    f32x4 bf[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) bf[n] = MODE == 3 ? U[t + 256 * n] : sB[lane + n * 64];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();

    int tq = t;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(tq));      // per-tile index arithmetic is recomputed, not hoisted into ~100 loop-invariant registers
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) {
                if (MODE != 3) __syncthreads();
                if (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) breg[i] = U[(gchunk & 15) * CH4 + tq + 256 * i];
                    // next tile's slab s in two halves: requested in chunks 0 / 1, written one chunk later
                    if (pg == 1 || pg == 2) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            int r = tq + 256 * (i + 3 * (pg - 1));
                            r = r < SLAB4 ? r : SLAB4 - 1;          // clamped like the load: duplicates rewrite the same value, no branch
                            sH[s * SLAB4 + r] = hreg[i];
                        }
                    }
                    if (pg == 0 || pg == 1) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            int r = tq + 256 * (i + 3 * pg);
                            r = r < SLAB4 ? r : SLAB4 - 1;
                            hreg[i] = src[(r >> 2) * 16 + s * 4 + (r & 3)];
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // the requests stay at the top of the chunk (hipcc sinks them to their use)
                if (MODE != 3) {
                    if (pg == 0) {
                        row_tf(3);
                        read_row(d0, (s + 1) & 3, 0);
                        read_row(d2, (s + 1) & 3, 2);
                    } else if (pg == 1) {
                        row_tf(0);
                        read_row(d1, (s + 1) & 3, 1);
                    } else if (pg == 2) {
                        row_tf(1);
                        read_row(d3, (s + 1) & 3, 3);
                    } else {
                        row_tf(2);
                    }
                }
                // B fragments run one group (16 MFMAs: one position, 4 N tiles x 4 k-steps) ahead of the MFMAs, across chunk
                // boundaries too: chunk c+1 has been visible since the barrier at the top of chunk c (it was written during c-1).
                const int slot_n = slot_r == 2 ? 0 : slot_r + 1;
#pragma unroll
                for (int pj = 0; pj < 4; ++pj) {
                    f32x4 bfn[4];
                    if (MODE != 3) {
                        const f32x4* bn = pj < 3 ? sB + slot_r * CH4 + lane + (pj + 1) * 256 : sB + slot_n * CH4 + lane;
#pragma unroll
                        for (int n = 0; n < 4; ++n) bfn[n] = bn[n * 64];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int n = 0; n < 4; ++n) acc[pg * 4 + pj][n] = mfma16(V[pg * 4 + pj][j], bf[n][j], acc[pg * 4 + pj][n]);
                    if (MODE != 3) {
#pragma unroll
                        for (int n = 0; n < 4; ++n) bf[n] = bfn[n];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) sB[slot_w * CH4 + tq + 256 * i] = breg[i];
                }
                slot_r = slot_r == 2 ? 0 : slot_r + 1;
                slot_w = slot_w == 2 ? 0 : slot_w + 1;
                ++gchunk;
            }
        }
        // ---- epilogue: Y = A^T M A per (tile, channel), store, clear
        float* ob = out + (size_t)blockIdx.x * 256 * 64;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x4 w0[4], w1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w0[i] = acc[i * 4 + 0][n] + acc[i * 4 + 1][n] + acc[i * 4 + 2][n];
                w1[i] = acc[i * 4 + 1][n] - acc[i * 4 + 2][n] - acc[i * 4 + 3][n];
            }
            const f32x4 y00 = w0[0] + w0[1] + w0[2], y01 = w1[0] + w1[1] + w1[2];
            const f32x4 y10 = w0[1] - w0[2] - w0[3], y11 = w1[1] - w1[2] - w1[3];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mo = 4 * ((tq >> 4) & 3) + r, oy = 2 * (2 * (tq >> 6) + (mo >> 3)), ox = 2 * (mo & 7);
                float* o = ob + (oy * 16 + ox) * 64 + n * 16 + (tq & 15);
                o[0] = y00[r];
                o[64] = y01[r];
                o[16 * 64] = y10[r];
                o[17 * 64] = y11[r];
            }
#pragma unroll
            for (int p = 0; p < 16; ++p) acc[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

static void host_U(const std::vector<float>& g, std::vector<float>& U) {   // g[n][c][3][3] -> chunk images
    const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    U.assign(16 * CH4 * 4, 0.f);
    for (int n = 0; n < 64; ++n)
        for (int c = 0; c < 64; ++c) {
            double u[4][4];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    double a = 0;
                    for (int y = 0; y < 3; ++y)
                        for (int x = 0; x < 3; ++x) a += G[i][y] * g[((n * 64 + c) * 3 + y) * 3 + x] * G[j][x];
                    u[i][j] = a;
                }
            const int s = c >> 4, kq = (c >> 2) & 3, jj = c & 3, nt = n >> 4, nn = n & 15;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
                    U[(((size_t)(s * 4 + i) * CH4) + (j * 4 + nt) * 64 + kq * 16 + nn) * 4 + jj] = (float)u[i][j];
        }
}

template <int MODE>
static void run(const char* name, const f32x4* src, const f32x4* U, float* out, unsigned long long* cyc, const std::vector<float>* ref) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(wino<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int blocks = 256, iters = 400;
    hipLaunchKernelGGL(wino<MODE>, dim3(blocks), dim3(256), LDS_BYTES, 0, src, U, out, 1, cyc);
    hipDeviceSynchronize();
    if (ref) {
        std::vector<float> h(256 * 64);
        hipMemcpy(h.data(), out + (size_t)7 * 256 * 64, h.size() * 4, hipMemcpyDeviceToHost);
        double e = 0, mx = 0;
        for (size_t i = 0; i < h.size(); ++i) {
            e = fmax(e, fabs((double)h[i] - (*ref)[i]));
            mx = fmax(mx, fabs((*ref)[i]));
        }
        printf("%-44s max |winograd - direct(fp64)| = %.3e  (max |ref| %.3f)\n", name, e, mx);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 4; ++l) hipLaunchKernelGGL(wino<MODE>, dim3(blocks), dim3(256), LDS_BYTES, 0, src, U, out, iters, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hc(blocks);
        hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double cs = 0;
        for (auto v : hc) cs += (double)v;
        cs /= blocks;
        const double tiles = 4.0 * blocks * iters;
        const double eff = tiles * 256 * 64 * 576 * 2 / (ms * 1e-3) / 1e12, exe = tiles * 4 * 1024 * 2048 / (ms * 1e-3) / 1e12;
        printf("%-44s rep %d: %7.2f ms  effective %6.1f TFLOP/s  executed %6.1f TFLOP/s  %6.0f cycles/tile (MFMA floor 32768)  clock %.2f GHz\n",
               name, rep, ms, eff, exe, cs / iters, cs / (ms / 4 * 1e-3) / 1e9);
    }
}

int main() {
    std::mt19937 gen(1);
    std::uniform_real_distribution<float> dist(-1.f, 1.f);
    std::vector<float> hs(NPIX * 64), g(64 * 64 * 9), hU;
    for (auto& v : hs) v = dist(gen);
    for (auto& v : g) v = dist(gen) * 0.05f;
    host_U(g, hU);
    std::vector<float> ref(256 * 64);
    for (int y = 0; y < 16; ++y)
        for (int x = 0; x < 16; ++x)
            for (int n = 0; n < 64; ++n) {
                double a = 0;
                for (int dy = 0; dy < 3; ++dy)
                    for (int dx = 0; dx < 3; ++dx)
                        for (int c = 0; c < 64; ++c) a += (double)hs[((y + dy) * HP + x + dx) * 64 + c] * g[((n * 64 + c) * 3 + dy) * 3 + dx];
                ref[(y * 16 + x) * 64 + n] = (float)a;
            }
    f32x4 *src, *U;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&src, (hs.size() + 4096 * 4) * 4);
    hipMalloc(&U, hU.size() * 4);
    hipMalloc(&out, (size_t)256 * 256 * 64 * 4);
    hipMalloc(&cyc, 256 * 8);
    hipMemset(src, 0, (hs.size() + 4096 * 4) * 4);
    hipMemcpy(src, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(U, hU.data(), hU.size() * 4, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) {
        run<0>("winograd F(2x2,3x3) full (ring+halo refill)", src, U, out, cyc, round == 0 ? &ref : nullptr);
        run<1>("  static LDS, no global traffic", src, U, out, cyc, round == 0 ? &ref : nullptr);
        run<2>("  static LDS, no input transform", src, U, out, cyc, nullptr);
        run<3>("  MFMA on register operands only", src, U, out, cyc, nullptr);
    }
    return 0;
}
