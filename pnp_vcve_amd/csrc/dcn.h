#pragma once
#include "common.h"

struct DcnArgs {
    const float* x;        // (H,W,64) pixel-major feature to sample (ref_unwarped)
    const float* om;       // (H,W,448) conv_offset[2] output in this build's channel order (prep.h)
    const float* fx;       // 'basic': flow planes added to every offset (dx += fx, dy += fy); nullptr for 'fvc'
    const float* fy;
    const float* w;        // packed B image of deform_align.weight (9 chunks)
    const void* w16;       // launch_dcn_f16_image of it: fp16 MFMA operands (PNP_PREC_F16), or nullptr = exact fp32 MFMA
    const float* bias;     // deform_align.bias [64]
    float* out;            // (H,W,64)
    int H, W;
    unsigned long long* dbg;   // diagnostic timeline (pnp_dcn_nhwc_f32_ex): 8 u64 per wave, or nullptr
};
int launch_dcn(const DcnArgs& a, hipStream_t stream);
// u64 elements DcnArgs::dbg must hold on the current device (8 per wave, 8 waves per block, one block per CU), -1 on error
int dcn_trace_u64s();
// fp32 B image (9 chunks) -> the fp16 image of DcnArgs::w16 (9 * 4096 halfs)
int launch_dcn_f16_image(const float* packed_w, void* dst, hipStream_t stream);
