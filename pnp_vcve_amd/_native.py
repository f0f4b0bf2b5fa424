"""ctypes binding of libpnpvcve_hip.so (the C ABI declared in include/pnpvcve.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libpnpvcve_hip.so')

PNP_ERR = {1001: 'bad argument', 1002: 'unsupported configuration', 1003: 'workspace too small or misaligned',
           1004: 'size assert', 1005: 'size value'}


class GeneratorCfg(ctypes.Structure):
    _fields_ = [(k, c_int) for k in (
        'mid_channels', 'num_blocks', 'num_experts', 'with_cat', 'use_base_qp', 'expert_softmax', 'with_bias',
        'with_se', 'one_layer', 'channel_first', 'align_key', 'vsr', 'deform', 'sparse_val', 'num_group', 'flow_inter',
        'blocktype')]


# name -> (restype, argtypes); every symbol include/pnpvcve.h declares
SIGNATURES = {
    'pnp_abi_version': (c_int, []),
    'pnp_generator_create': (c_int, [POINTER(GeneratorCfg), POINTER(c_void_p)]),
    'pnp_generator_destroy': (None, [c_void_p]),
    'pnp_generator_num_params': (c_int, [c_void_p]),
    'pnp_generator_param_name': (c_char_p, [c_void_p, c_int]),
    'pnp_generator_param_ndim': (c_int, [c_void_p, c_int]),
    'pnp_generator_param_dim': (c_int64, [c_void_p, c_int, c_int]),
    'pnp_generator_param_offset': (c_int64, [c_void_p, c_int]),
    'pnp_generator_flat_floats': (c_int64, [c_void_p]),
    'pnp_generator_packed_floats': (c_int64, [c_void_p]),
    'pnp_generator_pack': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'pnp_generator_workspace_bytes': (c_int64, [c_void_p, c_int, c_int, c_int]),
    'pnp_generator_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      POINTER(c_float), POINTER(c_float), POINTER(c_float),
                                      c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    'pnp_generator_set_precision': (c_int, [c_void_p, c_int]),
    'pnp_generator_get_precision': (c_int, [c_void_p]),
    'pnp_generator_set_option': (c_int, [c_void_p, c_int, c_int]),
    'pnp_generator_get_option': (c_int, [c_void_p, c_int]),
    'pnp_generator_profile': (c_int, [c_void_p, c_int]),
    'pnp_generator_profile_read': (c_int, [c_void_p, c_int, POINTER(ctypes.c_double), POINTER(c_int64),
                                           POINTER(ctypes.c_double)]),
    'pnp_flow_warp_nchw_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'pnp_mv_warp_nhwc_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'pnp_flow_warp_nchw_mode_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'pnp_mv_warp_nhwc_mode_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'pnp_nchw_to_nhwc_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'pnp_nhwc_to_nchw_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'pnp_caa_predict_f32': (c_int, [POINTER(c_float), POINTER(c_float), c_int, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'pnp_packed_conv_floats': (c_int64, [c_int]),
    'pnp_pack_conv3x3_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'pnp_pack_conv1x1_f32': (c_int, [c_void_p, c_void_p, c_void_p]),
    'pnp_frames_to_rgb8': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'pnp_psnr_sse_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'pnp_rasterise_side_info_f32': (c_int, [c_void_p, c_void_p, ctypes.c_long, POINTER(c_float), c_int, c_int, c_int,
                                            c_void_p, c_void_p, c_void_p, c_void_p]),
    'pnp_dcn_nhwc_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 c_void_p]),
    'pnp_dcn_ref_channel': (c_int, [c_int]),
    'pnp_dcn_f16_image_from_f32': (c_int, [c_void_p, c_void_p, c_void_p]),
    'pnp_dcn_nhwc_f16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 c_void_p]),
    'pnp_ssim_blocks': (c_int, [c_int, c_int, c_int]),
    'pnp_ssim_partials_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'pnp_conv3x3_f32': (c_int, [c_int, POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'pnp_bae_block_f32': (c_int, [c_void_p] * 10 + [c_int, c_int, c_void_p]),
    'pnp_packed_pixel_shuffle_floats': (c_int64, []),
    'pnp_pack_pixel_shuffle_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'pnp_pixel_shuffle_conv_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'pnp_par_tile_flags_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'pnp_f16_image_from_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'pnp_conv3x3_f16': (c_int, [c_int, POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'pnp_f16x3_image_from_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'pnp_wino_image_floats': (c_int64, []),
    'pnp_wino_par_image_floats': (c_int64, []),
    'pnp_wino_rgb_image_floats': (c_int64, []),
    'pnp_wino_rgb_image_from_packed_f32': (c_int, [c_void_p, c_void_p, c_void_p]),
    'pnp_conv3x3_wino_ms_f32': (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'pnp_conv3x3_wino_ms_units_f32': (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'pnp_wino_image_from_packed_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'pnp_wino_par_image_from_packed_f32': (c_int, [c_void_p, c_void_p, c_void_p]),
    'pnp_conv3x3_wino_f32': (c_int, [c_void_p] * 8 + [c_int, c_void_p, c_int, c_int, c_void_p]),
    'pnp_conv3x3_wino_units_f32': (c_int, [c_void_p] * 8 + [c_int, c_void_p, c_int, c_int, c_void_p]),
    'pnp_conv3x3_f16x3': (c_int, [c_int, POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), POINTER(c_void_p),
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_void_p, c_int, c_int, c_void_p]),
}

# include/pnpvcve_debug.h: kernel-variant selection / timelines for tests and tools
DEBUG_SIGNATURES = {
    'pnp_conv3x3_f16x3_ex': (c_int, [c_int, POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), POINTER(c_void_p),
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'pnp_conv3x3_f32_ex': (c_int, [c_int, POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                   c_void_p, c_void_p]),
    'pnp_conv3x3_f16_ex': (c_int, [c_int, POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
}

DEBUG_SIGNATURES['pnp_dcn_nhwc_f32_ex'] = (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                   c_int, c_void_p, c_void_p])

DEBUG_SIGNATURES['pnp_dcn_trace_u64s'] = (c_int, [])
DEBUG_SIGNATURES['pnp_debug_wino_gate_word'] = (c_int, [c_void_p])
DEBUG_SIGNATURES['pnp_conv3x3_wino_f32_ex'] = (c_int, [c_void_p] * 8 + [c_int, c_void_p, c_int, c_int, c_void_p, c_void_p])
DEBUG_SIGNATURES['pnp_conv3x3_f16_maps'] = (c_int, [c_int, POINTER(c_void_p), POINTER(c_int), c_int, POINTER(c_void_p), c_void_p,
                                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                                    c_int, c_int, c_int, c_void_p, c_void_p])
DEBUG_SIGNATURES['pnp_mv_warp_nhwc_f16out'] = (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p])

# pnp_generator_set_option ids (include/pnpvcve.h)
(OPT_F16_MAPS, OPT_PAR_SKIP, OPT_CONV_LAST_VALU, OPT_PERSIST, OPT_SMALL_F16, OPT_SPARSE_EVAL, OPT_F16_MIRRORS, OPT_F16_CHAIN_MIRRORS,
 OPT_TILE_QUEUE, OPT_WINOGRAD) = range(10)
CONV_AUTO, CONV_TILE, CONV_TILE_BIG = range(3)
WINO_UNITS_MAX_TILES = 128      # PNP_WINO_UNITS_MAX_TILES (checked against the header by tests/test_native_abi.py)


def wino_kernel_form(h, w, wopt):
    """which Winograd kernels PNP_OPT_WINOGRAD = wopt takes on h x w frames: None (direct kernels), 'units' or 'tiles' (pnpvcve.h)"""
    if wopt <= 0:
        return None
    tiles = ((h + 15) // 16) * ((w + 15) // 16)
    return 'units' if wopt == 1 and tiles <= WINO_UNITS_MAX_TILES else 'tiles'

_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `python -m pnp_vcve_amd.build_native` '
                '(hipcc --offload-arch=gfx950).  There is no CPU fallback for this path.')
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
            fn = getattr(L, name)       # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc == 0:
        return
    if rc == 1004:
        raise AssertionError(f'{what}: the height and width of inputs should be at least 64')
    if rc == 1005:
        raise ValueError(f'{what}: the spatial sizes of input and flow are not the same '
                         '(frame size must be a multiple of 4)')
    if rc in PNP_ERR:
        raise RuntimeError(f'{what}: {PNP_ERR[rc]} (pnp error {rc})')
    raise RuntimeError(f'{what}: HIP error {rc}')
