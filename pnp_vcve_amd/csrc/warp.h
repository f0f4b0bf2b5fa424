#pragma once
#include "common.h"
int launch_mv_warp_nhwc(const float* feat, const float* fx, const float* fy, float* out, int H, int W, int C,
                        hipStream_t stream);
int launch_flow_warp_nchw(const float* x, const float* flow, float* out, int N, int C, int H, int W,
                          hipStream_t stream);
