// Host-only run of the clip scheduler (pnp_vcve_amd/csrc/generator.hip) under AddressSanitizer / UBSan.
//
// TEST INFRASTRUCTURE.  Built by tests/test_host_scheduler.py with a plain host compiler:
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -DPNP_HOST_STUB -x c++ tests/host/sched_stub.cpp
// The HIP runtime is replaced by csrc/host_stub/hip_stub.h (events and streams are heap objects, a kernel launch is a record) and
// every launch_* entry of the other .hip files by a RECORDING launcher below that
//   * checks that each byte range the real kernel would read or write is addressable (ASan shadow) -- the buffers are plain
//     heap blocks of exactly the sizes the C ABI asks for, so a carve / size bug lands in a red zone;
//   * keeps a set of written ranges and refuses a read of bytes nothing has written (a schedule that consumes a map before its
//     producer ran, e.g. an fp16 mirror that was never made);
//   * records what the schedule did: which slot was aligned for which frame, which expert mixture each block conv used, how
//     many mixtures / events / streams were created.
// The scheduler itself is compiled UNCHANGED: this file #includes generator.hip.  The driver at the bottom runs the patterns
// named on the command line and prints one JSON object per scenario.
#include <algorithm>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../pnp_vcve_amd/csrc/conv_mfma.h"
#include "../../pnp_vcve_amd/csrc/dcn.h"
#include "../../pnp_vcve_amd/csrc/prep.h"
#include "../../pnp_vcve_amd/csrc/warp.h"
#ifndef __has_feature
#define __has_feature(x) 0
#endif
#if defined(__SANITIZE_ADDRESS__) || __has_feature(address_sanitizer)
#include <sanitizer/asan_interface.h>
#define PNP_HAVE_ASAN 1
#else
#define PNP_HAVE_ASAN 0
#endif

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace stub {

std::vector<std::string> errors;
void fail(const std::string& m) {
    if (errors.size() < 50) errors.push_back(m);
}

// ---- written-range bookkeeping (merged, sorted intervals of addresses)
std::map<uintptr_t, uintptr_t> written;      // lo -> hi
void mark(const void* p, size_t n) {
    if (!n) return;
    uintptr_t lo = (uintptr_t)p, hi = lo + n;
    auto it = written.upper_bound(lo);
    if (it != written.begin()) {
        auto pr = std::prev(it);
        if (pr->second >= lo) {
            lo = pr->first;
            hi = std::max(hi, pr->second);
            it = written.erase(pr);
        }
    }
    while (it != written.end() && it->first <= hi) {
        hi = std::max(hi, it->second);
        it = written.erase(it);
    }
    written[lo] = hi;
}
bool covered(const void* p, size_t n) {
    if (!n) return true;
    const uintptr_t lo = (uintptr_t)p, hi = lo + n;
    auto it = written.upper_bound(lo);
    if (it == written.begin()) return false;
    --it;
    return it->second >= hi;
}
bool addressable(const void* p, size_t n) {
#if PNP_HAVE_ASAN
    return __asan_region_is_poisoned(const_cast<void*>(p), n) == nullptr;
#else
    (void)p;
    (void)n;
    return true;
#endif
}
const char* cur = "?";       // launcher being emulated (for messages)
void RD(const char* what, const void* p, size_t n) {
    if (!p) return fail(std::string(cur) + ": null read of " + what);
    if (!addressable(p, n)) return fail(std::string(cur) + ": read of " + what + " leaves its buffer (" + std::to_string(n) + " B)");
    if (!covered(p, n)) fail(std::string(cur) + ": reads " + what + " (" + std::to_string(n) + " B) before anything wrote it");
}
void WR(const char* what, void* p, size_t n) {
    if (!p) return fail(std::string(cur) + ": null write of " + what);
    if (!addressable(p, n)) return fail(std::string(cur) + ": write of " + what + " leaves its buffer (" + std::to_string(n) + " B)");
    mark(p, n);
}

// ---- HIP objects
int next_stream = 1, next_event = 1, live_streams = 0, live_events = 0, streams_created = 0, events_created = 0;
struct Wait { int stream, event, event_recorded_on; };
std::vector<Wait> waits;
std::vector<std::pair<int, int>> records;      // (event, stream)
std::vector<int> launch_streams;               // stream id of every recorded launch, in order

int sid(hipStream_t s) { return s ? s->id : 0; }
void note_launch(hipStream_t s) { launch_streams.push_back(sid(s)); }

// ---- what the schedule did (filled by the launchers, interpreted by the driver, which knows the workspace layout)
struct WarpRec { const void *feat, *fx, *out; bool f16; };
struct ConvRec {
    ConvArgs a;
    int cfg, gy, stream;
    int path;          // 0 fp32 kernels | 1 fp16-operand kernels | 2 split-fp16 kernel
};
struct MixRec { const void* dst; int E, gy; };
std::vector<WarpRec> warps;
std::vector<ConvRec> convs;
std::vector<MixRec> mixes;
int dcn_calls = 0;

}  // namespace stub

// =================================================================================================== HIP runtime stand-ins
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
    *e = new pnp_stub_event{stub::next_event++, -1};
    ++stub::live_events;
    ++stub::events_created;
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    --stub::live_events;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    e->recorded_on = stub::sid(s);
    stub::records.push_back({e->id, stub::sid(s)});
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    if (e->recorded_on < 0) stub::fail("hipEventSynchronize on an event that was never recorded");
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    if (a->recorded_on < 0 || b->recorded_on < 0) stub::fail("hipEventElapsedTime on an unrecorded event");
    *ms = 1.0f;
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    *s = new pnp_stub_stream{stub::next_stream++};
    ++stub::live_streams;
    ++stub::streams_created;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    delete s;
    --stub::live_streams;
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    if (e->recorded_on < 0) stub::fail("hipStreamWaitEvent on an event that was never recorded");
    stub::waits.push_back({stub::sid(s), e->id, e->recorded_on});
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int, size_t bytes, hipStream_t s) {
    stub::cur = "hipMemsetAsync";
    stub::note_launch(s);
    stub::WR("memset destination", dst, bytes);
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDevice(int* d) {
    *d = 0;
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

// the scheduler's own two little kernels (generator.hip): fill_kernel(dst, v, n), small_copy_kernel(src, dst, n_valid, n_total, mode)
void pnp_stub_kernel_launch_impl(const char* name, dim3, dim3, hipStream_t s, const PnpStubArg* a, int n) {
    stub::cur = name;
    stub::note_launch(s);
    const std::string k(name);
    if (k == "fill_kernel" && n == 3) {
        stub::WR("fill destination", const_cast<void*>(a[0].p), (size_t)a[2].i * 4);
    } else if (k == "small_copy_kernel" && n == 5) {
        const long long mode = a[4].i;
        stub::RD("copy source", a[0].p, (size_t)(mode == 0 ? a[2].i : mode == 1 ? 256 : 432) * 4);
        stub::WR("copy destination", const_cast<void*>(a[1].p), (size_t)a[3].i * 4);
    } else {
        stub::fail("unexpected kernel launch from the scheduler: " + k);
    }
}

// =================================================================================================== recording launchers

int conv_pick_cfg(int H, int W) {
    const long tiles_big = (long)((W + 15) / 16) * ((H + 7) / 8);
    return tiles_big >= 1024 ? CONV_CFG_BIG : CONV_CFG_SMALL;
}

static void conv_touch(const ConvArgs& a, int cfg, int gy, int path) {
    const bool f16 = path == 1, x3 = path == 2;
    using namespace stub;
    const size_t hw = (size_t)a.H * a.W;
    const bool rgb_head = a.out_mode == 2 || a.out_mode == 3;
    for (int s = 0; s < a.nsrc; ++s) {
        const bool s16 = f16 && ((a.src_f16 >> s) & 1);
        RD("a conv source", a.src[s], hw * a.src_c[s] * (s16 ? 2 : 4));
        const size_t img = a.src_c[s] == 64 ? (rgb_head ? 9 * 2048 : 9 * 4096) : 4096;
        for (int y = 0; y < gy; ++y) {
            if (f16) RD("an fp16 weight image", (const uint16_t*)a.wsrc_h[s] + (size_t)y * a.w_ystride, img * 2);
            else if (x3 && a.src_c[s] == 64) RD("a split fp16 weight image", (const char*)a.wsrc_h[s] + (size_t)y * a.w_ystride * 4, img * 4);
            else RD("a weight image", a.wsrc[s] + (size_t)y * a.w_ystride, img * 4);
        }
    }
    if (a.wpar || a.wpar_h) {
        if (f16) RD("the fp16 1x1 weight images", a.wpar_h, 3 * 4096 * 2);
        else if (x3) RD("the split fp16 1x1 weight images (+ their 1/255-scaled twins)", a.wpar_h, (a.wpar_h_scaled ? 6 : 3) * 4096 * 4);
        else RD("the 1x1 weight images", a.wpar, 3 * 4096 * 4);
        RD("the partition planes", a.par, (size_t)(2 * a.par_plane + hw) * 4);
        if (a.par_flags) RD("the partition tile flags", a.par_flags, (size_t)((a.W + 15) / 16) * ((a.H + 7) / 8) * 4);
        if (a.par_any) RD("the frame's partition-record word", a.par_any, 4);
    }
    if (path == 0 && conv_wino_ms_eligible(a, cfg, gy) && !a.wwino) {
        RD("the frame's Winograd weight image", a.wwino_rgb, (size_t)PNP_WINO_RGB_FLOATS * 4);
        for (int s = 1; s < a.nsrc; ++s) RD("a source's Winograd weight image", a.wwino_src[s], (size_t)PNP_WINO_IMG_FLOATS * 4);
    }
    if (path == 0 && a.wwino && conv_wino_eligible(a, cfg, gy)) {      // the Winograd kernel reads these INSTEAD of wsrc / wpar (read above as well: harmless over-check)
        RD("the Winograd weight image", a.wwino, (size_t)PNP_WINO_IMG_FLOATS * 4);
        if (a.wpar) RD("the Winograd 1x1 weight image", a.wwino_par, (size_t)PNP_WINO_PAR_FLOATS * 4);
    }
    if (a.bias) RD("the bias", a.bias, (size_t)((gy - 1) * a.bias_ystride + (rgb_head ? 3 : 64)) * 4);
    if (a.gamma) RD("the channel gain", a.gamma, 64 * 4);
    if (a.residual) RD("the residual map", a.residual, hw * 256);
    if (a.wvalu && !f16) RD("the vector-ALU conv_last weights", a.wvalu, 9 * 64 * 4 * 4);
    if (rgb_head) {
        RD("the low-quality frame", a.lr, (size_t)(2 * a.lr_plane + (a.out_mode == 2 ? hw : hw / 16)) * 4);
        WR("the output frame", a.out, hw * 3 * 4);
    } else if (a.out_mode == 1) {
        WR("the pixel-shuffled map", a.out, hw * 4 * ((f16 && a.out_f16) ? 128 : 256));
    } else if (a.out_mode == 4) {
        WR("the offset/mask map", a.out, hw * (size_t)a.out_cstride * 4);
    } else {
        WR("the output map", a.out, hw * 64 * ((f16 && a.out_f16) ? 2 : 4));
        if (f16 && a.out16) WR("the fp16 mirror of the output", a.out16, hw * 128);
    }
    (void)cfg;
}

int launch_conv3x3(const ConvArgs& a, int cfg, int gy, hipStream_t s) {
    stub::cur = "launch_conv3x3";
    stub::note_launch(s);
    if (a.nsrc < 1 || a.nsrc > 4) return PNP_ERR_BAD_ARG;
    const bool f16 = a.prec == 1 && conv_f16_eligible(a, cfg, gy);
    const bool x3 = a.prec == 2 && conv_f16x3_eligible(a, cfg, gy);
    if (!f16 && (a.src_f16 || a.out_f16 || a.out16)) {
        stub::fail("a conv with fp16 maps is not eligible for the fp16 kernels (the fp32 kernel would read halfs as floats)");
        return PNP_ERR_UNSUPPORTED;
    }
    conv_touch(a, cfg, gy, f16 ? 1 : (x3 ? 2 : 0));
    stub::convs.push_back({a, cfg, gy, stub::sid(s), f16 ? 1 : (x3 ? 2 : 0)});
    return 0;
}
int launch_conv3x3_f16(const ConvArgs& a, int gy, hipStream_t s) {
    stub::cur = "launch_conv3x3_f16";
    stub::note_launch(s);
    conv_touch(a, CONV_CFG_BIG, gy, 1);
    stub::convs.push_back({a, CONV_CFG_BIG, gy, stub::sid(s), 1});
    return 0;
}
int launch_conv3x3_f16x3(const ConvArgs& a, int cfg, hipStream_t s) {
    stub::cur = "launch_conv3x3_f16x3";
    stub::note_launch(s);
    conv_touch(a, cfg, 1, 2);
    stub::convs.push_back({a, cfg, 1, stub::sid(s), 2});
    return 0;
}
int launch_f16x3_image(const float* src, void* dst, int nchunks, hipStream_t s) {
    stub::cur = "launch_f16x3_image";
    stub::note_launch(s);
    const size_t n = (size_t)nchunks * pnp_chunk_floats(2);
    stub::RD("fp32 weight images", src, n * 4);
    stub::WR("split fp16 weight images", dst, n * 4);
    return 0;
}
int launch_f16_image(const float* src, void* dst, int nchunks, int ntb, hipStream_t s) {
    stub::cur = "launch_f16_image";
    stub::note_launch(s);
    const size_t n = (size_t)nchunks * pnp_chunk_floats(ntb);
    stub::RD("fp32 weight images", src, n * 4);
    stub::WR("fp16 weight images", dst, n * 2);
    return 0;
}
bool conv_wino_eligible(const ConvArgs& a, int cfg, int grid_y) {       // conv_wino.hip's rule, restated
    if (!a.wwino || a.prec != 0 || cfg == CONV_CFG_RGB || grid_y != 1 || a.out_mode != 0) return false;
    if (a.nsrc != 1 || a.src_c[0] != 64 || a.src_f16 || a.out_f16 || a.out16) return false;
    if (a.wpar && (!a.wwino_par || !a.par)) return false;
    return (long)a.H * a.W * 256 < ((long)1 << 32) - 65536;
}
bool conv_wino_ms_eligible(const ConvArgs& a, int cfg, int grid_y) {    // conv_wino.hip's rule, restated
    if (!a.wwino_rgb || a.prec != 0 || cfg == CONV_CFG_RGB || grid_y != 1 || a.out_mode != 0) return false;
    if (a.nsrc < 2 || a.nsrc > 4 || a.src_c[0] != 4 || a.src_f16 || a.out_f16 || a.out16) return false;
    if (a.wpar || a.residual || a.gamma) return false;
    for (int s = 1; s < a.nsrc; ++s)
        if (a.src_c[s] != 64 || !a.wwino_src[s]) return false;
    return (long)a.H * a.W * 256 < ((long)1 << 32) - 65536;
}
int launch_wino_rgb_image(const float* src, float* dst, hipStream_t s) {
    stub::cur = "launch_wino_rgb_image";
    stub::note_launch(s);
    stub::RD("the frame's packed weight chunk", src, 4096 * 4);
    stub::WR("the frame's Winograd weight image", dst, (size_t)PNP_WINO_RGB_FLOATS * 4);
    return 0;
}
int launch_wino_images(const float* const* src, float* const* dst, int n, const float* gamma, hipStream_t s) {
    stub::cur = "launch_wino_images";
    stub::note_launch(s);
    if (n < 1 || n > 16) return PNP_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i) {
        stub::RD("a packed 3x3 weight image", src[i], 9 * 4096 * 4);
        stub::WR("a Winograd weight image", dst[i], (size_t)PNP_WINO_IMG_FLOATS * 4);
    }
    if (gamma) stub::RD("the channel gain", gamma, 64 * 4);
    return 0;
}
int launch_wino_par_image(const float* src, float* dst, hipStream_t s) {
    stub::cur = "launch_wino_par_image";
    stub::note_launch(s);
    stub::RD("the packed 1x1 weight images", src, 3 * 4096 * 4);
    stub::WR("the Winograd 1x1 weight image", dst, (size_t)PNP_WINO_PAR_FLOATS * 4);
    return 0;
}
int launch_conv3x3_wino(const ConvArgs&, hipStream_t) { return PNP_ERR_UNSUPPORTED; }      // only reached through launch_conv3x3
int launch_par_tile_flags(const float* par, long plane, int* flags, int frames, int H, int W, hipStream_t s) {
    stub::cur = "launch_par_tile_flags";
    stub::note_launch(s);
    stub::RD("partition maps", par, (size_t)frames * 3 * plane * 4);
    stub::WR("partition tile flags", flags, (size_t)frames * ((W + 15) / 16) * ((H + 7) / 8) * 4);
    return 0;
}
int launch_par_frame_any(const int* flags, int* any, int frames, int H, int W, hipStream_t s) {
    stub::cur = "launch_par_frame_any";
    stub::note_launch(s);
    stub::RD("partition tile flags", flags, (size_t)frames * ((W + 15) / 16) * ((H + 7) / 8) * 4);
    stub::WR("per-frame partition-record words", any, (size_t)frames * 4);
    return 0;
}
int launch_pack_last_valu(const float* w, float* dst, hipStream_t s) {
    stub::cur = "launch_pack_last_valu";
    stub::note_launch(s);
    stub::RD("conv_last.weight", w, 3 * 64 * 9 * 4);
    stub::WR("vector-ALU conv_last weights", dst, 9 * 64 * 4 * 4);
    return 0;
}
int launch_mv_warp_nhwc(const float* feat, const float* fx, const float* fy, void* out, int H, int W, int C, hipStream_t s, bool f16, bool) {
    stub::cur = "launch_mv_warp_nhwc";
    stub::note_launch(s);
    const size_t hw = (size_t)H * W;
    stub::RD("the key-frame feature", feat, hw * C * 4);
    stub::RD("flow plane x", fx, hw * 4);
    stub::RD("flow plane y", fy, hw * 4);
    stub::WR("the aligned map", out, hw * C * (f16 ? 2 : 4));
    stub::warps.push_back({feat, fx, out, f16});
    return 0;
}
int launch_flow_warp_nchw(const float* x, const float* flow, float* out, int N, int C, int H, int W, hipStream_t s, bool) {
    stub::cur = "launch_flow_warp_nchw";
    stub::note_launch(s);
    stub::RD("x", x, (size_t)N * C * H * W * 4);
    stub::RD("flow", flow, (size_t)N * H * W * 8);
    stub::WR("out", out, (size_t)N * C * H * W * 4);
    return 0;
}
int launch_pack_lr(const float* lrs, float* lr4, int T, int H, int W, hipStream_t s) {
    stub::cur = "launch_pack_lr";
    stub::note_launch(s);
    stub::RD("the low-quality frames", lrs, (size_t)T * 3 * H * W * 4);
    stub::WR("the packed RGB0 frames", lr4, (size_t)T * H * W * 16);
    return 0;
}
int launch_par_sparse(const float* par, float* out, int T, int H, int W, hipStream_t s) {
    stub::cur = "launch_par_sparse";
    stub::note_launch(s);
    stub::RD("partition maps", par, (size_t)T * 3 * H * W * 4);
    stub::WR("sparse-equivalent partition maps", out, (size_t)T * 3 * H * W * 4);
    return 0;
}
int launch_pack_flow4(const float* fx, const float* fy, float* out4, int H, int W, hipStream_t s) {
    stub::cur = "launch_pack_flow4";
    stub::note_launch(s);
    stub::RD("flow plane x", fx, (size_t)H * W * 4);
    stub::RD("flow plane y", fy, (size_t)H * W * 4);
    stub::WR("flow4", out4, (size_t)H * W * 16);
    return 0;
}
int launch_nchw_to_nhwc(const float* in, float* out, int N, int C, int H, int W, hipStream_t s) {
    stub::cur = "launch_nchw_to_nhwc";
    stub::note_launch(s);
    stub::RD("in", in, (size_t)N * C * H * W * 4);
    stub::WR("out", out, (size_t)N * C * H * W * 4);
    return 0;
}
int launch_nhwc_to_nchw(const float* in, float* out, int N, int C, int H, int W, hipStream_t s) { return launch_nchw_to_nhwc(in, out, N, C, H, W, s); }
int launch_pack_weights(const PackArgs& a, int gy, hipStream_t s) {
    stub::cur = "launch_pack_weights";
    stub::note_launch(s);
    const size_t co_max = a.co_mode == 1 ? 431 : (size_t)(a.n_valid - 1) * a.co_mul + a.co_add;
    const size_t rows = (co_max + 1) * (size_t)a.cin_total * a.ktaps;
    const size_t img = (a.kind == PACK_WIDE ? 9 : 1) * (size_t)a.ntb * 2048;
    for (int y = 0; y < gy; ++y) {
        stub::RD("reference-layout weights", a.w + (size_t)y * a.w_ystride, ((size_t)(a.E - 1) * a.e_stride + rows) * 4);
        stub::WR("a packed weight image", a.dst + (size_t)y * a.dst_ystride, img * 4);
    }
    if (a.ew) stub::RD("expert attention", a.ew, (size_t)a.E * 4);
    if (a.E > 1 && gy > 1) stub::mixes.push_back({a.dst, a.E, gy});
    return 0;
}
int launch_mix_bias(const float* b, const float* ew, float* out, int E, int C, int nconv, hipStream_t s) {
    stub::cur = "launch_mix_bias";
    stub::note_launch(s);
    stub::RD("expert biases", b, (size_t)nconv * E * C * 4);
    stub::RD("expert attention", ew, (size_t)E * 4);
    stub::WR("mixed biases", out, (size_t)nconv * C * 4);
    return 0;
}
int launch_caa_predict(const CaaArgs& a, hipStream_t s) {
    stub::cur = "launch_caa_predict";
    stub::note_launch(s);
    if (a.count < 1 || a.count > 32) stub::fail("caa: count out of range");
    stub::RD("BasePredictor.0.weight", a.w1, 64 * 4);
    stub::RD("BasePredictor.0.bias", a.b1, 64 * 4);
    stub::RD("BasePredictor.2.weight", a.w2, (size_t)a.E * 64 * 4);
    stub::RD("BasePredictor.2.bias", a.b2, (size_t)a.E * 4);
    if (a.with_se) {
        stub::RD("BiasePredictor.fc.0.weight", a.v1, 4 * 4);
        stub::RD("BiasePredictor.fc.2.weight", a.v2, 256 * 4);
    }
    stub::WR("expert attention", a.ew + (size_t)a.t0 * a.E, (size_t)a.count * a.E * 4);
    stub::WR("channel gains", a.gamma + (size_t)a.t0 * 64, (size_t)a.count * 64 * 4);
    return 0;
}
int launch_dcn(const DcnArgs& a, hipStream_t s) {
    stub::cur = "launch_dcn";
    stub::note_launch(s);
    const size_t hw = (size_t)a.H * a.W;
    stub::RD("the feature to sample", a.x, hw * 256);
    stub::RD("the offset/mask map", a.om, hw * 448 * 4);
    if (a.fx) stub::RD("flow plane x", a.fx, hw * 4);
    if (a.fy) stub::RD("flow plane y", a.fy, hw * 4);
    if (a.w16) stub::RD("the fp16 DCN weight image", a.w16, 9 * 4096 * 2);
    else stub::RD("the DCN weight image", a.w, 9 * 4096 * 4);
    stub::RD("deform_align.bias", a.bias, 64 * 4);
    stub::WR("the aligned map", a.out, hw * 256);
    ++stub::dcn_calls;
    return 0;
}
int dcn_trace_u64s() { return 256 * 64; }
int launch_dcn_f16_image(const float* w, void* dst, hipStream_t s) {
    stub::cur = "launch_dcn_f16_image";
    stub::note_launch(s);
    stub::RD("the DCN weight image", w, 9 * 4096 * 4);
    stub::WR("the fp16 DCN weight image", dst, 9 * 4096 * 2);
    return 0;
}

// =================================================================================================== the scheduler, unchanged
#include "../../pnp_vcve_amd/csrc/generator.hip"

// =================================================================================================== driver
namespace {

struct Scenario {
    std::string name;
    pnp_generator_cfg cfg;
    int prec, n, t, h, w, contexts, forwards, profile;
    std::vector<float> slices, qps, bqs;        // n * t each
    int mirrors;                                // 0 none, 1 PNP_OPT_F16_MIRRORS (default), 2 + PNP_OPT_F16_CHAIN_MIRRORS
    int wino = 0;                               // PNP_OPT_WINOGRAD
};

pnp_generator_cfg default_cfg() {
    pnp_generator_cfg c;
    memset(&c, 0, sizeof(c));
    c.mid_channels = 64;
    c.num_blocks = 8;
    c.num_experts = 6;
    c.with_cat = c.use_base_qp = c.expert_softmax = c.with_bias = c.with_se = c.one_layer = c.channel_first = c.align_key = 1;
    return c;
}

void json_ints(const char* key, const std::vector<int>& v, bool last = false) {
    printf("\"%s\": [", key);
    for (size_t i = 0; i < v.size(); ++i) printf("%s%d", i ? ", " : "", v[i]);
    printf("]%s", last ? "" : ", ");
}

int run(const Scenario& sc) {
    using namespace stub;
    errors.clear();
    written.clear();
    waits.clear();
    records.clear();
    launch_streams.clear();
    warps.clear();
    convs.clear();
    mixes.clear();
    dcn_calls = 0;
    streams_created = events_created = 0;
    pnp_generator* g = nullptr;
    int rc = pnp_generator_create(&sc.cfg, &g);
    if (rc) {
        printf("{\"name\": \"%s\", \"create_rc\": %d}\n", sc.name.c_str(), rc);
        return 0;
    }
    pnp_generator_set_precision(g, sc.prec);
    pnp_generator_set_option(g, PNP_OPT_F16_MIRRORS, sc.mirrors >= 1);
    pnp_generator_set_option(g, PNP_OPT_F16_CHAIN_MIRRORS, sc.mirrors >= 2);
    pnp_generator_set_option(g, PNP_OPT_WINOGRAD, sc.wino);
    const int64_t flat_n = pnp_generator_flat_floats(g), packed_n = pnp_generator_packed_floats(g);
    const int64_t ctx_bytes = pnp_generator_workspace_bytes(g, sc.t, sc.h, sc.w);
    const int64_t ws_bytes = ctx_bytes * sc.contexts;
    const size_t hw = (size_t)sc.h * sc.w, os = sc.cfg.vsr ? 4 : 1;
    // exact-size heap blocks: ASan's red zones start at the first byte past what the ABI asked for
    float* flat = (float*)malloc((size_t)flat_n * 4);
    float* packed = (float*)malloc((size_t)packed_n * 4);
    char* ws = nullptr;
    // self-test of the harness: PNP_STUB_SHRINK_WS=<bytes> hands the scheduler a workspace that is smaller than it was told
    const char* shrink_s = getenv("PNP_STUB_SHRINK_WS");
    const int64_t shrink = shrink_s ? atoll(shrink_s) : 0;
    if (posix_memalign((void**)&ws, 256, (size_t)(ws_bytes - shrink))) return 2;
    float* lrs = (float*)malloc((size_t)sc.n * sc.t * 3 * hw * 4);
    float* mvs = (float*)malloc((size_t)sc.n * sc.t * 4 * hw * 4);
    float* par = (float*)malloc((size_t)sc.n * sc.t * 3 * hw * 4);
    float* out = (float*)malloc((size_t)sc.n * sc.t * 3 * hw * os * os * 4);
    mark(flat, (size_t)flat_n * 4);
    mark(lrs, (size_t)sc.n * sc.t * 3 * hw * 4);
    mark(mvs, (size_t)sc.n * sc.t * 4 * hw * 4);
    mark(par, (size_t)sc.n * sc.t * 3 * hw * 4);
    pnp_stub_stream caller{0};
    rc = pnp_generator_pack(g, flat, packed, &caller);
    const size_t launches_pack = launch_streams.size();
    if (sc.profile) pnp_generator_profile(g, 1);
    std::vector<int> pool_sizes, streams_after, events_after;
    int frc = 0;
    size_t launches_first = 0;
    for (int f = 0; f < sc.forwards && frc == 0; ++f) {
        if (f > 0) {            // a fresh forward must not depend on what the previous one left in the workspace
            for (auto it = written.begin(); it != written.end();) {
                if (it->first >= (uintptr_t)ws && it->second <= (uintptr_t)ws + (size_t)ws_bytes) it = written.erase(it);
                else ++it;
            }
            warps.clear();
            convs.clear();
            mixes.clear();
            waits.clear();
            launch_streams.clear();
            if (sc.profile) pnp_generator_profile(g, 1);
        }
        frc = pnp_generator_forward(g, flat, packed, lrs, mvs, par, sc.slices.data(), sc.qps.data(), sc.bqs.data(), out, ws, ws_bytes,
                                    sc.n, sc.t, sc.h, sc.w, &caller);
        if (f == 0) launches_first = launch_streams.size() - (sc.forwards > 1 ? 0 : launches_pack);
        if (sc.profile) {
            double ms, wk;
            int64_t nl;
            for (int k = 0; k < 5; ++k) pnp_generator_profile_read(g, k, &ms, &nl, &wk);
        }
        pool_sizes.push_back((int)g->prof_pool.size());
        streams_after.push_back(streams_created);
        events_after.push_back(events_created);
    }
    // ---- interpret the records with the scheduler's own carve of context 0.. (anonymous-namespace functions: same TU)
    printf("{\"name\": \"%s\", \"pack_rc\": %d, \"forward_rc\": %d, ", sc.name.c_str(), rc, frc);
    std::vector<int> warp_frame, warp_dir, warp_key, warp_ctx, warp_f16;
    const size_t fm = hw * 64;
    for (const WarpRec& wr : warps) {
        int ctx = -1, key = -1;
        for (int k = 0; k < sc.contexts; ++k) {
            const Workspace W = carve(g, ws + (int64_t)k * ctx_bytes, sc.t, sc.h, sc.w);
            if ((const float*)wr.feat >= W.slots && (const float*)wr.feat < W.slots + fm * sc.t) {
                ctx = k;
                key = (int)(((const float*)wr.feat - W.slots) / fm);
                if (((const float*)wr.feat - W.slots) % fm) fail("a warp reads from the middle of a slot");
                if (wr.out != (sc.cfg.deform == 0 ? (void*)W.kw : (void*)W.tmp0)) fail("a warp writes somewhere unexpected");
            }
        }
        const size_t plane = ((const float*)wr.fx - mvs) / hw;        // (sample * t + frame) * 4 + {0 fwd, 2 bwd}
        warp_frame.push_back((int)((plane / 4) % sc.t));
        warp_dir.push_back((int)(plane % 4));
        warp_key.push_back(key);
        warp_ctx.push_back(ctx);
        warp_f16.push_back(wr.f16 ? 1 : 0);
    }
    json_ints("warp_frame", warp_frame);
    json_ints("warp_flow_plane", warp_dir);
    json_ints("warp_key_slot", warp_key);
    json_ints("warp_context", warp_ctx);
    json_ints("warp_f16", warp_f16);
    // expert mixtures: one per distinct routing value and context-sample; which one each partition-branch conv used
    std::vector<int> mix_slot, block_frame, block_mix, conv_f16, conv_nsrc, conv_mask, conv_wino, conv_wino_ms, conv_wino_units;
    const Workspace W0 = carve(g, ws, sc.t, sc.h, sc.w);
    for (const MixRec& m : mixes) mix_slot.push_back((int)(((const float*)m.dst - W0.mixw) % ((int64_t)ctx_bytes / 4) / ((int64_t)g->ndyn * IMG_WIDE)));
    for (const ConvRec& c : convs) {
        conv_f16.push_back(c.path);
        conv_nsrc.push_back(c.a.nsrc);
        conv_wino.push_back((c.path == 0 && conv_wino_eligible(c.a, c.cfg, c.gy)) ? 1 : 0);
        conv_wino_ms.push_back((c.path == 0 && conv_wino_ms_eligible(c.a, c.cfg, c.gy)) ? 1 : 0);
        conv_wino_units.push_back((c.path == 0 && (conv_wino_eligible(c.a, c.cfg, c.gy) || conv_wino_ms_eligible(c.a, c.cfg, c.gy)) && c.a.wino_units) ? 1 : 0);
        conv_mask.push_back(c.a.src_f16 | (c.a.out_f16 ? 16 : 0) | (c.a.out16 ? 32 : 0));
        if (!(c.a.wpar || c.a.wpar_h)) continue;
        const size_t pl = (c.a.par - par) / (3 * hw);
        block_frame.push_back((int)(pl % sc.t));
        const float* wimg = c.path ? nullptr : c.a.wsrc[0];
        int u = -1;
        for (int k = 0; k < sc.contexts && u < 0; ++k) {
            const Workspace W = carve(g, ws + (int64_t)k * ctx_bytes, sc.t, sc.h, sc.w);
            if (c.path) {
                const uint16_t* hh = (const uint16_t*)c.a.wsrc_h[0];
                const int64_t halfs = c.path == 2 ? 2 : 1;
                if (hh >= (const uint16_t*)W.mixh && hh < (const uint16_t*)W.mixh + halfs * sc.t * g->ndyn * IMG_WIDE)
                    u = (int)((hh - (const uint16_t*)W.mixh) / (halfs * g->ndyn * IMG_WIDE));
            } else if (wimg >= W.mixw && wimg < W.mixw + (int64_t)sc.t * g->ndyn * IMG_WIDE) {
                u = (int)((wimg - W.mixw) / ((int64_t)g->ndyn * IMG_WIDE));
            }
        }
        block_mix.push_back(u);
    }
    json_ints("mix_slot", mix_slot);
    json_ints("par_conv_frame", block_frame);
    json_ints("par_conv_mixture", block_mix);
    json_ints("conv_f16_path", conv_f16);
    json_ints("conv_nsrc", conv_nsrc);
    json_ints("conv_wino", conv_wino);
    json_ints("conv_wino_ms", conv_wino_ms);
    json_ints("conv_wino_units", conv_wino_units);
    json_ints("conv_map_mask", conv_mask);
    json_ints("launch_stream", launch_streams);
    std::vector<int> wait_stream, wait_on;
    for (const Wait& wv : waits) {
        wait_stream.push_back(wv.stream);
        wait_on.push_back(wv.event_recorded_on);
    }
    json_ints("wait_stream", wait_stream);
    json_ints("wait_event_recorded_on", wait_on);
    json_ints("event_pool_after_forward", pool_sizes);
    json_ints("streams_created_after_forward", streams_after);
    json_ints("events_created_after_forward", events_after);
    printf("\"dcn_calls\": %d, \"launches_first_forward\": %zu, \"workspace_bytes\": %lld, \"context_bytes\": %lld, ", dcn_calls, launches_first,
           (long long)ws_bytes, (long long)ctx_bytes);
    // every output byte of the call must have been written
    if (frc == 0 && !covered(out, (size_t)sc.n * sc.t * 3 * hw * os * os * 4)) fail("the output clip is not completely written");
    pnp_generator_destroy(g);
    printf("\"live_events_after_destroy\": %d, \"live_streams_after_destroy\": %d, \"errors\": [", live_events, live_streams);
    for (size_t i = 0; i < errors.size(); ++i) printf("%s\"%s\"", i ? ", " : "", errors[i].c_str());
    printf("]}\n");
    free(flat);
    free(packed);
    free(ws);
    free(lrs);
    free(mvs);
    free(par);
    free(out);
    return errors.empty() ? 0 : 1;
}

std::vector<float> pattern(const std::string& p, int t) {
    std::vector<float> s(t, 66.f);
    if (p == "IBBBP") {
        for (int i = 0; i < t; ++i) s[i] = i == 0 ? 73.f : (i % 4 == 0 ? 80.f : 66.f);
    } else if (p == "allP") {
        for (int i = 0; i < t; ++i) s[i] = i == 0 ? 73.f : 80.f;
    } else if (p == "allB") {
        s[0] = 73.f;
    } else {                       // explicit letters
        for (int i = 0; i < t && i < (int)p.size(); ++i) s[i] = (float)p[i];
    }
    return s;
}

}  // namespace

int main(int argc, char** argv) {
    std::vector<Scenario> all;
    auto add = [&](const std::string& name, pnp_generator_cfg c, int prec, int n, int t, int h, int w, int contexts,
                   const std::vector<std::string>& pats, const std::vector<float>& crf, int forwards = 1, int profile = 0, int mirrors = 1) {
        Scenario s;
        s.name = name;
        s.cfg = c;
        s.prec = prec;
        s.n = n;
        s.t = t;
        s.h = h;
        s.w = w;
        s.contexts = contexts;
        s.forwards = forwards;
        s.profile = profile;
        s.mirrors = mirrors;
        for (int b = 0; b < n; ++b) {
            const std::vector<float> sl = pattern(pats[b % pats.size()], t);
            for (int i = 0; i < t; ++i) {
                s.slices.push_back(sl[i]);
                s.qps.push_back((20.f + (float)((i * 7 + b) % 20)) / 255.f);
                s.bqs.push_back(crf[b % crf.size()] / 255.f);
            }
        }
        all.push_back(s);
    };
    const pnp_generator_cfg d = default_cfg();
    pnp_generator_cfg vsr = d, basic = d, nocat = d, chlast = d, sparse = d, qprouted = d;
    vsr.vsr = 1;
    basic.deform = 1;
    nocat.with_cat = 0;
    nocat.align_key = 0;
    chlast.channel_first = 0;
    chlast.one_layer = 0;
    sparse.sparse_val = 1;
    qprouted.use_base_qp = 0;
    qprouted.with_bias = 0;
    qprouted.with_se = 0;
    for (int prec = 0; prec < 3; ++prec) {
        const std::string p = prec == 0 ? "f32_" : (prec == 1 ? "f16_" : "x3_");
        add(p + "ibbbp_t7", d, prec, 1, 7, 64, 96, 1, {"IBBBP"}, {25});
        add(p + "allB_t7", d, prec, 1, 7, 64, 64, 1, {"allB"}, {35});
        add(p + "allP_t7", d, prec, 1, 7, 64, 64, 1, {"allP"}, {15});
        add(p + "n2_mixed_t5", d, prec, 2, 5, 64, 64, 1, {"IBBPB", "IPBBB"}, {15, 35});
        add(p + "t1", d, prec, 1, 1, 64, 64, 1, {"I"}, {25});
        add(p + "t100", d, prec, 1, 100, 64, 64, 1, {"IBBBP"}, {25});
        add(p + "n8_ctx8_twice", d, prec, 8, 3, 64, 64, 8, {"IBBBP", "allP"}, {15, 25, 35}, 2);
        add(p + "n5_ctx3", d, prec, 5, 3, 64, 64, 3, {"IBBBP"}, {15, 25});
        add(p + "profiled_twice", d, prec, 1, 4, 64, 64, 1, {"IBBBP"}, {25}, 2, 1);
        add(p + "vsr_t2", vsr, prec, 1, 2, 64, 80, 1, {"IBBBP"}, {25});
        add(p + "basic_t3", basic, prec, 1, 3, 64, 64, 1, {"IBBBP"}, {25});
        add(p + "nocat_noalign_t4", nocat, prec, 1, 4, 72, 88, 1, {"IBPB"}, {25});
        add(p + "channel_last_two_layer_t3", chlast, prec, 1, 3, 64, 64, 1, {"IBBBP"}, {25});
        add(p + "sparse_val_t3", sparse, prec, 1, 3, 64, 64, 1, {"IBBBP"}, {25});
        add(p + "qp_routed_t6", qprouted, prec, 1, 6, 64, 64, 1, {"IBBBP"}, {25});
        add(p + "p720_t2", d, prec, 1, 2, 720, 1280, 1, {"IBBBP"}, {25});
    }
    // PNP_OPT_WINOGRAD: 2 = every frame size, 1 = frames with >= 512 16x16 tiles only
    add("f32_wino2_ibbbp_t7", d, 0, 1, 7, 64, 96, 1, {"IBBBP"}, {25});
    all.back().wino = 2;
    add("f32_wino2_channel_last_two_layer_t3", chlast, 0, 1, 3, 64, 64, 1, {"IBBBP"}, {25});
    all.back().wino = 2;
    add("f32_wino2_vsr_t2", vsr, 0, 1, 2, 64, 80, 1, {"IBBBP"}, {25});
    all.back().wino = 2;
    add("f32_wino1_p720_t2", d, 0, 1, 2, 720, 1280, 1, {"IBBBP"}, {25});
    all.back().wino = 1;
    add("f32_wino1_ibbbp_t7", d, 0, 1, 7, 64, 96, 1, {"IBBBP"}, {25});
    all.back().wino = 1;
    add("f16_wino2_ibbbp_t7", d, 1, 1, 7, 64, 96, 1, {"IBBBP"}, {25});
    all.back().wino = 2;
    add("f16_nomirrors_ibbbp_t7", d, 1, 1, 7, 64, 96, 1, {"IBBBP"}, {25}, 1, 0, 0);
    add("f16_chainmirrors_ibbbp_t7", d, 1, 1, 7, 64, 96, 1, {"IBBBP"}, {25}, 1, 0, 2);
    add("f16_chainmirrors_channel_last_two_layer_t3", chlast, 1, 1, 3, 64, 64, 1, {"IBBBP"}, {25}, 1, 0, 2);
    add("f16_chainmirrors_p720_t2", d, 1, 1, 2, 720, 1280, 1, {"IBBBP"}, {25}, 1, 0, 2);
    int bad = 0;
    for (const Scenario& s : all) {
        bool want = argc < 2;
        for (int i = 1; i < argc; ++i) want = want || s.name == argv[i];
        if (want) bad += run(s);
    }
    return bad ? 1 : 0;
}
