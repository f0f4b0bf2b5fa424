// The multi-source instantiation of conv3x3_wino_kernel (the input convs over [frame, 1-3 wide sources]) and its quadrant-unit twin as
// their own translation unit: same source, other register-allocation flags (build_native.py; conv_wino.hip explains why).
#define WINO_MS_TU 1
#include "conv_wino.hip"
