#pragma once
#include "common.h"
// out_f16: write the aligned map as fp16 NHWC (saturating round-to-nearest-even of the fp32 result) instead of fp32
int launch_mv_warp_nhwc(const float* feat, const float* fx, const float* fy, void* out, int H, int W, int C,
                        hipStream_t stream, bool out_f16 = false, bool nearest = false);
// nearest: flow_warp(interpolation='nearest') -- the pixel at nearbyint of the sampling position (ties to even), 0 outside the image
int launch_flow_warp_nchw(const float* x, const float* flow, float* out, int N, int C, int H, int W,
                          hipStream_t stream, bool nearest = false);
