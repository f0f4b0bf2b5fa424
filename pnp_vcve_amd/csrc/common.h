// Shared device/host helpers for the PnP-VCVE hot-path kernels (gfx950 only).
#pragma once
#ifdef PNP_HOST_STUB
#include "host_stub/hip_stub.h"   // host-only sanitizer build of the scheduler (tests/test_host_scheduler.py)
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PNP_CHECK(expr)                                   \
    do {                                                  \
        hipError_t _e = (expr);                           \
        if (_e != hipSuccess) return (int)_e;             \
    } while (0)

// Error codes of the C ABI live in include/pnpvcve.h (>= 1000, never a hipError_t value).
#include "../../include/pnpvcve.h"
#define PNP_OK 0

// ---------------------------------------------------------------------------
// Packed weight image ("B image") geometry, shared by the packers and the conv
// kernel.  One *chunk* = 8 q-steps; one q-step feeds 4 MFMA 32x32x2 k-steps.
//   float index inside a chunk = ((q * NTB + nt) * 64 + lane) * 4 + j
//   lane = h*32 + n  -> output channel co = nt*32 + n, k-half h
// A 64-channel source uses 9 chunks (one per 3x3 tap): element (q,h,j) is input
// channel 8q + 4h + j of that tap.  A 4-channel source (the RGB frame, 4th
// channel zero) uses 1 chunk whose q-step q (0..4) pairs taps 2q (h=0) and 2q+1
// (h=1), j = channel.  A 1x1 branch uses 1 chunk (channel 8q + 4h + j).
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// Per-device one-time setup.  hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count
// are properties of a DEVICE, not of the process: a process that drives several GPUs must
// apply / query them once per device.  One PerDevice object per kernel (a function-local
// static); run(f) calls f() the first time it is reached with a given current device and
// returns f's cached hipError_t afterwards.  `value` is a per-device int f may fill
// (e.g. the persistent grid size).
// ---------------------------------------------------------------------------
#include <mutex>
struct PnpPerDevice {
    static constexpr int MAXDEV = 64;
    std::mutex mu;
    bool done[MAXDEV] = {};
    hipError_t err[MAXDEV] = {};
    int value[MAXDEV] = {};
    template <class F>
    hipError_t run(F f, int* value_out = nullptr) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= MAXDEV) return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> lock(mu);
        if (!done[dev]) {
            err[dev] = f(dev, value[dev]);
            done[dev] = true;
        }
        if (value_out) *value_out = value[dev];
        return err[dev];
    }
};

#define PNP_CHUNK_Q 8
static inline __host__ __device__ int pnp_chunk_floats(int ntb) { return PNP_CHUNK_Q * ntb * 64 * 4; }
