"""ctypes binding of libt2h_hip.so (include/t2h_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol
cannot be resolved, importing the ops fails loudly (the oracle under oracle/ is
test infrastructure and is never used here).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libt2h_hip.so')

c_f32p = ctypes.c_void_p
c_i64p = ctypes.c_void_p
c_u8p = ctypes.c_void_p
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float
c_vp = ctypes.c_void_p


class GemmArgs(ctypes.Structure):
    """struct t2h_gemm_args (include/t2h_hip.h)."""
    _fields_ = [
        ('A', c_vp), ('B', c_vp), ('C', c_vp), ('bias', c_vp), ('residual', c_vp),
        ('pro_scale', c_vp), ('pro_shift', c_vp),
        ('M', c_i32), ('N', c_i32), ('K', c_i32),
        ('lda', c_i32), ('ldb', c_i32), ('ldc', c_i32), ('ldr', c_i32),
        ('a_mode', c_i32), ('b_trans', c_i32), ('pro_act', c_i32),
        ('pro_rows', c_i32), ('pro_ld', c_i32), ('epi_act', c_i32), ('res_pre', c_i32),
        ('alpha', c_f32),
        ('Hin', c_i32), ('Win', c_i32), ('Cin', c_i32), ('Hout', c_i32), ('Wout', c_i32),
        ('stride', c_i32), ('pad', c_i32), ('ups', c_i32), ('batch', c_i32),
        ('strideA', c_i64), ('strideB', c_i64), ('strideC', c_i64),
        ('gn_part_out', c_vp),
        ('splitk_ws', c_vp), ('splitk_ws_floats', c_i64), ('ksplit', c_i32),
    ]


class GemmSplitArgs(ctypes.Structure):
    """struct t2h_gemm_split_args (include/t2h_hip.h)."""
    _fields_ = [
        ('A', c_vp), ('B', c_vp), ('C', c_vp), ('C_split', c_vp), ('bias', c_vp), ('residual', c_vp),
        ('M', c_i32), ('N', c_i32), ('K', c_i32), ('ldc', c_i32), ('ldr', c_i32), ('epi_act', c_i32),
        ('Vt', c_vp), ('vt_col0', c_i32), ('vt_T', c_i32), ('vt_hd', c_i32), ('overflow_flag', c_vp),
        ('fmt', c_i32), ('out_fmt', c_i32), ('lo_mul', c_f32), ('out_scale', c_f32), ('ksplit', c_i32),
    ]


MAX_HEADS = 32


class SampleHeadsArgs(ctypes.Structure):
    """struct t2h_sample_heads_args (include/t2h_hip.h)."""
    _fields_ = [
        ('hidden', c_vp), ('lnf_gamma', c_vp), ('lnf_beta', c_vp), ('w_heads', c_vp),
        ('expo', c_vp * MAX_HEADS), ('rows', c_vp), ('tex', c_vp), ('x_t', c_vp), ('out_idx', c_vp),
        ('temp', c_f32), ('n_rows', c_i32), ('n', c_i32), ('C', c_i32), ('n_class', c_i32), ('n_heads', c_i32),
        ('logits_ws', c_vp), ('hidden_compact', c_i32),
        ('philox_seed', ctypes.c_uint64), ('philox_offset', ctypes.c_uint64 * MAX_HEADS),
        ('philox_grid_threads', ctypes.c_uint32),
        ('row_philox_offset', c_vp), ('expo_rows', c_vp), ('expo_slot', c_vp), ('philox_seed_dev', c_vp),
        ('rng_rows', c_vp),
    ]


# name -> (restype, argtypes); must list EVERY symbol declared in include/t2h_hip.h
SIGNATURES = {
    't2h_gemm_split_f32': (ctypes.c_int, [ctypes.POINTER(GemmSplitArgs), c_vp]),
    't2h_gemm_split_time_next_launch': (ctypes.c_int, [c_vp, c_vp]),
    't2h_gemm_split_force_config': (ctypes.c_int, [ctypes.c_int]),
    't2h_gemm_split_probe_next_launch': (ctypes.c_int, [c_vp]),
    't2h_gemm_split_tile_config': (ctypes.c_int, [ctypes.POINTER(GemmSplitArgs)]),
    't2h_conv3x3_small_f32': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32,
                                             c_i32, c_i32, c_vp]),
    't2h_conv_split_force_tile': (ctypes.c_int, [ctypes.c_int]),
    't2h_mha_split_force_form': (ctypes.c_int, [ctypes.c_int]),
    't2h_split_rows_x8_f32': (ctypes.c_int, [c_vp, c_i32, c_vp, ctypes.c_int64, c_i32, c_f32, c_vp, c_vp]),
    't2h_layernorm_x8_f32': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp]),
    't2h_mha_split_x8_f32': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_f32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    't2h_split_overflow_async': (ctypes.c_int, [c_vp, c_vp, c_i32, c_vp]),
    't2h_split_rows_f32': (ctypes.c_int, [c_vp, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp]),
    't2h_layernorm_split_f32': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp, c_vp]),
    't2h_mha_noncausal_split_f32': (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    't2h_mha_split_f32': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    't2h_version': (ctypes.c_int, []),
    't2h_last_error': (ctypes.c_char_p, []),
    't2h_gemm_f32': (ctypes.c_int, [ctypes.POINTER(GemmArgs), c_vp]),
    't2h_gemm_tile_config': (ctypes.c_int, [ctypes.POINTER(GemmArgs)]),
    't2h_gemm_ksplit': (ctypes.c_int, [ctypes.POINTER(GemmArgs)]),
    't2h_conv_split_f32': (ctypes.c_int, [ctypes.POINTER(GemmArgs), c_vp]),
    't2h_conv_halo_f32': (ctypes.c_int, [ctypes.POINTER(GemmArgs), c_vp, c_vp]),
    't2h_conv_halo_force_variant': (ctypes.c_int, [ctypes.c_int]),
    't2h_gn_apply_split_f32': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    't2h_gemm_force_config': (ctypes.c_int, [ctypes.c_int]),
    't2h_layernorm_f32': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp]),
    't2h_groupnorm_workspace_bytes': (c_i64, [c_i32, c_i32, c_i32]),
    't2h_groupnorm_tables_f32': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32,
                                                c_i32, c_i32, c_f32, c_vp, c_vp]),
    't2h_groupnorm_finalize_f32': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]),
    't2h_spatial_attention_f32': (ctypes.c_int, [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]),
    't2h_softmax_rows_f32': (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp]),
    't2h_embed_sum4_f32': (ctypes.c_int, [c_vp] * 8 + [c_i32, c_i32, c_i32, c_vp]),
    't2h_mha_noncausal_f32': (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    't2h_unmask_step': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp]),
    't2h_sample_heads': (ctypes.c_int, [ctypes.POINTER(SampleHeadsArgs), c_vp]),
    't2h_absmax_f32': (ctypes.c_int, [c_vp, c_i32, c_i64, c_i32, c_vp, c_vp]),
    't2h_split_rows_absmax': (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp]),
    't2h_gather_rows': (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    't2h_philox_exponential_f32': (ctypes.c_int, [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, c_vp, c_i64, c_vp]),
    't2h_philox_uniform_f32': (ctypes.c_int, [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, c_vp, c_i64, c_vp]),
    't2h_unmask_schedule': (ctypes.c_int, [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                           ctypes.c_uint32, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    't2h_schedule_advance': (ctypes.c_int, [c_vp] * 7 + [c_i32, c_vp]),
    't2h_q_sample': (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp, c_i32, c_i32, c_vp]),
    't2h_masked_ce_heads': (ctypes.c_int, [c_vp] * 9 + [c_i32] * 5 + [c_vp]),
    't2h_sample_head': (ctypes.c_int, [c_vp] * 7 + [c_i32, c_f32, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    't2h_vq_argmin_tex_f32': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    't2h_vq_l2_argmin_f32': (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    't2h_codebook_gather_tex_f32': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    't2h_codebook_gather_fold_f32': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                                    c_i32, c_i32, c_vp]),
    't2h_routed_head_argmax': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32,
                                              c_i32, c_vp]),
    't2h_onehot_nhwc_f32': (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    't2h_nchw_to_nhwc_f32': (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    't2h_nhwc_to_nchw_f32': (ctypes.c_int, [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp]),
    't2h_maxpool2_nhwc_f32': (ctypes.c_int, [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    't2h_bilinear_up2_nhwc_f32': (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    't2h_argmax_rows_f32': (ctypes.c_int, [c_vp, c_i32, c_vp, c_i64, c_i32, c_vp]),
    't2h_image_epilogue': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp]),
    't2h_shape_attr_embed_f32': (ctypes.c_int, [c_vp] * 11 + [c_i32] * 4 + [c_vp]),
    't2h_tap_bias_map_f32': (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    't2h_texture_map': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
}

_lib = None


class T2HError(RuntimeError):
    pass


def load():
    """Loads libt2h_hip.so and resolves every declared symbol (fails loudly)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise T2HError(
            f'{LIB_PATH} not found: build it with `python -m text2human_amd.build` '
            '(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback.')
    # PyTorch-ROCm bundles its own libamdhip64.so (same SONAME).  Load torch FIRST so
    # that this library binds to the HIP runtime torch already initialised -- the
    # streams and device pointers handed over by torch belong to that runtime.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


n_calls = 0  # C-ABI calls checked so far (bench.py: host launches per sampling round)


def check(status, what=''):
    global n_calls
    n_calls += 1
    if status != 0:
        msg = load().t2h_last_error()
        raise T2HError(f'{what} failed ({status}): {msg.decode() if msg else "?"}')
