// Bitstream side-info rasteriser (the step right BEFORE the hot path).
//
// Reference: LoadImageFromFileList_ipb.__call__ (mmedit/datasets/pipelines/loading_ipb.py:328-369):
// a python loop over the per-frame MV records of the decoder
//     (direction, w, h, x_w, y_w, x, y, motion_x, motion_y, scale)
// painting block rectangles into a dense (H,W,4) motion map [fwd x,y | bwd x,y] and a (H,W,3)
// one-hot partition map, later records overwriting earlier ones; a P frame's records are painted,
// negated, into the backward channels of the previous anchor frame (mvs[-p_offset]).  Followed by
// RescaleToZeroOne on the partitions (/255) and FramesToTensor (HWC -> CHW).
//
// Here: records stay on the device and two kernels produce the (T,4,H,W) / (T,3,H,W) tensors the
// generator consumes, so the 28 B/pixel/frame of dense maps never cross PCIe.
//   1. mark   : one block per record; "later wins" is an atomicMax of (record index + 1) into a
//               per-pixel winner map (forward / backward); partitions are an OR (plain stores of 1/255).
//   2. resolve: one thread per pixel reads its winners and writes the four motion planes.
// Python slice semantics of the reference are kept exactly (a negative start index wraps around,
// the far side is clipped) -- see py_slice().  HBM-bound, 8 (winners) + 28 (maps) bytes per pixel.
#include "prep.h"

namespace {

struct FrameInfo {
    signed char is_b[256];
    short target[256];       // frame receiving a non-B frame's direction>0 records, -1 if undefined
};

__device__ __forceinline__ void py_slice(int start, int stop, int n, int& s, int& e) {
    if (start < 0) { start += n; if (start < 0) start = 0; } else if (start > n) start = n;
    if (stop < 0) { stop += n; if (stop < 0) stop = 0; } else if (stop > n) stop = n;
    s = start;
    e = stop;
}

__global__ __launch_bounds__(256) void raster_mark_kernel(const float* __restrict__ rec, const int* __restrict__ rec_frame,
                                                          long nrec, FrameInfo fi, int H, int W, int* __restrict__ win,
                                                          float* __restrict__ par) {
    const long r = blockIdx.x;
    if (r >= nrec) return;
    const float* p = rec + r * 10;
    const float direction = p[0];
    const int bw = (int)p[1], bh = (int)p[2], xw = (int)p[3], yw = (int)p[4], x = (int)p[5], y = (int)p[6];
    const int f = rec_frame[r];
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const long hw = (long)H * W;
    int ys, ye, xs, xe;
    py_slice(y - bh / 2, y + bh / 2, H, ys, ye);
    py_slice(x - bw / 2, x + bw / 2, W, xs, xe);
    const bool in_own = (ys + ty < ye) && (xs + tx < xe);
    const long own = (long)(ys + ty) * W + xs + tx;
    // partition one-hot by block area (loading_ipb.py:331-336,359-363), RescaleToZeroOne -> 1/255
    const int area = bw * bh;
    const int ch = area == 256 ? 0 : area == 128 ? 1 : area == 64 ? 2 : -1;
    if (in_own && ch >= 0) par[((long)f * 3 + ch) * hw + own] = 1.0f / 255.0f;
    const int tag = (int)r + 1;
    if (direction < 0.f) {
        if (in_own) atomicMax(&win[((long)f * 2 + 0) * hw + own], tag);
    } else if (direction > 0.f) {
        if (fi.is_b[f]) {
            if (in_own) atomicMax(&win[((long)f * 2 + 1) * hw + own], tag);
        } else {
            const int tf = fi.target[f];
            int ys2, ye2, xs2, xe2;
            py_slice(yw - bh / 2, yw + bh / 2, H, ys2, ye2);
            py_slice(xw - bw / 2, xw + bw / 2, W, xs2, xe2);
            if (tf >= 0 && (ys2 + ty < ye2) && (xs2 + tx < xe2))
                atomicMax(&win[((long)tf * 2 + 1) * hw + (long)(ys2 + ty) * W + xs2 + tx], tag);
        }
    }
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(const float* __restrict__ rec, const int* __restrict__ rec_frame,
                                                             const int* __restrict__ win, float* __restrict__ mvs,
                                                             long hw, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // over T*hw
    if (i >= total) return;
    const long f = i / hw, p = i - f * hw;
    const int wf = win[(f * 2 + 0) * hw + p], wb = win[(f * 2 + 1) * hw + p];
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    if (wf > 0) {
        const float* q = rec + (long)(wf - 1) * 10;
        m0 = q[7] / q[9];
        m1 = q[8] / q[9];
    }
    if (wb > 0) {
        const float* q = rec + (long)(wb - 1) * 10;
        const float sgn = (rec_frame[wb - 1] == (int)f) ? 1.f : -1.f;   // painted by a later P frame: negated
        m2 = sgn * (q[7] / q[9]);
        m3 = sgn * (q[8] / q[9]);
    }
    float* o = mvs + f * 4 * hw + p;
    o[0] = m0;
    o[hw] = m1;
    o[2 * hw] = m2;
    o[3 * hw] = m3;
}

}  // namespace

extern "C" int pnp_rasterise_side_info_f32(const float* records, const int* rec_frame, long num_records,
                                           const float* slices_host, int t, int h, int w, float* mvs, float* par,
                                           int* scratch, void* stream) {
    if (t < 1 || t > 256 || h < 1 || w < 1 || num_records < 0 || num_records > 0x7ffffff0L) return PNP_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    FrameInfo fi;
    int p_offset = -1;                       // unassigned in the reference until the first frame has been processed
    for (int f = 0; f < t; ++f) {
        const bool is_b = slices_host[f] == 66.0f;
        fi.is_b[f] = is_b;
        fi.target[f] = (short)((!is_b && p_offset > 0 && f - p_offset >= 0) ? f - p_offset : -1);
        p_offset = is_b ? (p_offset < 0 ? -1 : p_offset + 1) : 1;
    }
    const long hw = (long)h * w;
    int ze = launch_zero_words(scratch, 2 * hw * t, st);          // kernels, not memset nodes (prep.h)
    if (ze != PNP_OK) return ze;
    ze = launch_zero_words(par, 3 * hw * t, st);
    if (ze != PNP_OK) return ze;
    if (num_records > 0)
        hipLaunchKernelGGL(raster_mark_kernel, dim3((unsigned)num_records), dim3(256), 0, st, records, rec_frame,
                           num_records, fi, h, w, scratch, par);
    const long total = hw * t;
    hipLaunchKernelGGL(raster_resolve_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, records, rec_frame,
                       scratch, mvs, hw, total);
    return (int)hipGetLastError();
}
