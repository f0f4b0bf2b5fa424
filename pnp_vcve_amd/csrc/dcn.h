#pragma once
#include "common.h"

struct DcnArgs {
    const float* x;        // (H,W,64) pixel-major feature to sample (ref_unwarped)
    const float* om;       // (H,W,448) conv_offset[2] output in this build's channel order (prep.h)
    const float* fx;       // 'basic': flow planes added to every offset (dx += fx, dy += fy); nullptr for 'fvc'
    const float* fy;
    const float* w;        // packed B image of deform_align.weight (9 chunks)
    const float* bias;     // deform_align.bias [64]
    float* out;            // (H,W,64)
    int H, W;
};
int launch_dcn(const DcnArgs& a, hipStream_t stream);
