"""ctypes binding of libtante_hip.so (include/tante_hip.h).  There is NO fallback: if the library is
missing or a call fails, the product path raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# TANTE_LIB: an A/B build of the library (tools/build_variant.sh -> tools/_ab/lib_<name>.so) instead of the product one -- measurement
# scripts only, so that they never overwrite tante_amd/lib/libtante_hip.so; the file must exist (there is no fallback either way)
LIB_PATH = os.environ.get("TANTE_LIB") or os.path.join(HERE, "lib", "libtante_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_RELU = 0, 1, 2, 3
A_LINEAR, A_PATCH_NHWC, A_PATCH_NCHW = 0, 1, 2
E_LINEAR, E_FILM, E_DECONV_NHWC, E_DECONV_NCHW = 0, 1, 2, 3
W_LINEAR, W_CONV_NHWC, W_DECONV_NHWC, W_DECONV_NCHW = 0, 1, 2, 3
W_LINEAR_T, W_CONV_NHWC_T, W_DECONV_NHWC_T, W_DECONV_NCHW_T = 4, 5, 6, 7

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class PackGeom(C.Structure):
    _fields_ = [("n_pad", c_i32), ("k_pad", c_i32), ("nt", c_i32), ("cb", c_i32), ("bytes", c_i64)]


class Gemm(C.Structure):
    _fields_ = [
        ("a", c_vp), ("a_dtype", c_i32), ("a_mode", c_i32), ("M", c_i32), ("K", c_i32),
        ("a_s1", c_i64), ("a_s0", c_i64), ("a_off", c_i64), ("a_n0", c_i32),
        ("Hin", c_i32), ("Win", c_i32), ("Cin", c_i32), ("P", c_i32), ("ln", c_i32), ("ln_eps", c_f32),
        ("w", c_vp), ("bias", c_vp), ("N", c_i32), ("compute", c_i32),
        ("act", c_i32), ("e_mode", c_i32), ("out", c_vp), ("out_dtype", c_i32), ("out_ld", c_i64),
        ("residual", c_vp), ("res_ld", c_i64),
        ("film_a", c_vp), ("film_b", c_vp), ("s_emb", c_vp), ("T", c_i32), ("HW", c_i32),
        ("Hi", c_i32), ("Wi", c_i32), ("Po", c_i32), ("Cout", c_i32),
        ("drop_p", c_f32), ("drop_seed", C.c_uint64), ("dact", c_vp), ("dact_dtype", c_i32), ("dact_kind", c_i32),
        ("a_pad", c_i32),
    ]


class RowMat(C.Structure):
    _fields_ = [("p", c_vp), ("dtype", c_i32), ("mode", c_i32), ("s1", c_i64), ("s0", c_i64), ("off", c_i64), ("es", c_i64),
                ("n0", c_i32), ("Hin", c_i32), ("Win", c_i32), ("Cin", c_i32), ("P", c_i32)]


class WgradJob(C.Structure):      # TanteWgradJob
    _fields_ = [("U", C.POINTER(RowMat)), ("V", C.POINTER(RowMat)), ("n_seg", c_i32), ("R", c_i64), ("I", c_i32), ("J", c_i32), ("dW", c_vp), ("dbias", c_vp),
                ("layout", c_i32), ("P", c_i32), ("C_other", c_i32), ("swap", c_i32)]


class Seq(C.Structure):
    _fields_ = [("nseq", c_i32), ("L", c_i32), ("n_s0", c_i32), ("S1", c_i64), ("S0", c_i64),
                ("n_l0", c_i32), ("P1", c_i64), ("P0", c_i64)]


class Fold(C.Structure):      # TanteFold (include/tante_hip.h)
    _fields_ = [("GW", c_vp), ("Gb", c_vp), ("W", c_vp), ("gamma", c_vp), ("beta", c_vp), ("dW", c_vp), ("db", c_vp), ("dgamma", c_vp),
                ("dbeta", c_vp), ("N", c_i32), ("K", c_i32)]


class FoldFwd(C.Structure):      # TanteFoldFwd
    _fields_ = [("W", c_vp), ("b", c_vp), ("gamma", c_vp), ("beta", c_vp), ("We", c_vp), ("be", c_vp), ("N", c_i32), ("K", c_i32)]


class BlockWeights(C.Structure):  # TanteBlockWeights
    _fields_ = [("in_w", c_vp), ("in_b", c_vp), ("out_w", c_vp), ("out_b", c_vp), ("fc1_w", c_vp), ("fc1_b", c_vp), ("fc2_w", c_vp), ("fc2_b", c_vp),
                ("block_stream", c_vp)]


class Frames(C.Structure):       # TanteFrames
    _fields_ = [("f", c_vp * 8), ("bstride", c_i64 * 8)]


class Mat3(C.Structure):         # TanteMat3
    _fields_ = [("a", c_vp), ("b", c_vp), ("c", c_vp), ("dst", c_vp)]


class BlockTrain(C.Structure):
    _fields_ = [("out", c_vp), ("xh1", c_vp), ("qkv", c_vp), ("o", c_vp), ("xh2", c_vp), ("hpre", c_vp), ("act", c_vp),
                ("st1", c_vp), ("x1", c_vp), ("st2", c_vp), ("p_drop", c_f32), ("seed_attn", C.c_uint64), ("seed_out", C.c_uint64),
                ("seed_mlp", C.c_uint64)]


class TailOrdF(C.Structure):      # TanteTailOrdF
    _fields_ = [("x", c_vp), ("w", c_vp), ("coef", c_f32), ("xl16", c_vp), ("pre1", c_vp), ("act1", c_vp), ("pre2", c_vp), ("act2", c_vp)]


class TailFwd(C.Structure):       # TanteTailFwd
    _fields_ = [("o", TailOrdF * 3), ("n_ord", c_i32), ("a_s1", c_i64), ("a_s0", c_i64), ("a_off", c_i64), ("a_n0", c_i32),
                ("n_img", c_i32), ("Hp", c_i32), ("Wp", c_i32), ("D", c_i32), ("base", c_vp), ("base_bstride", c_i64),
                ("out", c_vp), ("out_bstride", c_i64), ("we", c_vp), ("f16", c_vp), ("pre1e", c_vp), ("act1e", c_vp),
                ("pre2e", c_vp), ("act2e", c_vp), ("z", c_vp)]


class TailOrdB(C.Structure):      # TanteTailOrdB
    _fields_ = [("w", c_vp), ("coef", c_f32), ("pre1", c_vp), ("pre2", c_vp), ("dpre1", c_vp), ("dpre2", c_vp), ("dder", c_vp),
                ("dx", c_vp), ("db1", c_vp), ("db2", c_vp), ("db3", c_vp), ("act2", c_vp), ("dw3", c_vp)]


class TailBwd(C.Structure):       # TanteTailBwd
    _fields_ = [("o", TailOrdB * 3), ("n_ord", c_i32), ("a_s1", c_i64), ("a_s0", c_i64), ("a_off", c_i64), ("a_n0", c_i32),
                ("n_img", c_i32), ("Hp", c_i32), ("Wp", c_i32), ("D", c_i32), ("dext", c_vp), ("dext_bstride", c_i64),
                ("dbase", c_vp), ("dbase_bstride", c_i64), ("dz", c_vp), ("we", c_vp), ("pre1e", c_vp), ("pre2e", c_vp),
                ("dz16", c_vp), ("dpre2e", c_vp), ("dpre1e", c_vp), ("bias_ws", c_vp), ("f16", c_vp), ("dwe1", c_vp), ("dbe1", c_vp)]


SIGNATURES = {
    "tante_tail_supported": ([c_i32, c_i32, c_i32, c_i32], c_i32),
    "tante_tail_stream_bytes": ([c_i32], c_i64),
    "tante_tail_pack_dec": ([c_vp] * 6 + [c_i32, c_vp, c_vp, c_vp], c_i32),
    "tante_tail_pack_enc": ([c_vp] * 6 + [c_i32, c_vp, c_vp, c_vp], c_i32),
    "tante_tail_fwd": ([C.POINTER(TailFwd), c_vp], c_i32),
    "tante_tail_bwd": ([C.POINTER(TailBwd), c_vp], c_i32),
    "tante_pack_geom": ([c_i32, c_i32, c_i32, C.POINTER(PackGeom)], c_i32),
    "tante_pack_weight": ([c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp], c_i32),
    "tante_gemm": ([C.POINTER(Gemm), c_vp], c_i32),
    "tante_attention": ([c_vp, c_vp, c_i32, c_i32, c_i32, C.POINTER(Seq), c_i32, c_vp], c_i32),
    "tante_axis_mlp": ([c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp], c_i32),
    "tante_axis_mlp_c": ([c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp], c_i32),
    "tante_axis_mlp_oop": ([c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp], c_i32),
    "tante_axis_hw": ([c_vp, c_i64, c_i32, c_i32, c_i32] + [c_vp] * 8 + [c_i32, c_vp], c_i32),
    "tante_axis_hw_film": ([c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32] + [c_vp] * 8 + [c_i32, c_vp], c_i32),
    "tante_axis_hw_oop": ([c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp], c_i32),
    "tante_axis_hw_train": ([c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp], c_i32),
    "tante_film_table": ([c_vp, c_i32, c_i32] + [c_vp] * 8 + [c_vp, c_vp, c_vp, c_vp], c_i32),
    "tante_film_apply": ([c_vp, c_i64, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp], c_i32),
    "tante_format_input": ([c_vp, c_i64, c_i32, c_i64, c_i32, c_vp, c_i64, c_vp], c_i32),
    "tante_taylor": ([c_vp, c_i64, C.POINTER(c_vp), c_i32, C.c_double, c_i32, c_vp, c_i64, c_i64, c_i64, c_vp], c_i32),
    "tante_rt_reduce": ([c_vp, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp], c_i32),
    "tante_gather_last": ([c_vp, c_i64, c_i32, c_vp, c_vp], c_i32),
    "tante_block_fused_supported": ([c_i32, c_i32, c_i32, c_i32], c_i32),
    "tante_block_stream_bytes": ([c_i32, c_i32], c_i64),
    "tante_pack_block": ([c_vp] * 12 + [c_i32, c_i32, c_vp, c_vp], c_i32),
    "tante_block_fused": ([c_vp, c_vp, c_i32, c_i32, c_i32, C.POINTER(Seq), c_i32, c_f32, c_vp], c_i32),
    "tante_block_fused_tprop": ([c_vp, c_vp, c_i32, c_i32, c_i32, C.POINTER(Seq), c_i32, c_f32, c_vp, c_vp], c_i32),
    "tante_block_tail_bwd_stream_bytes": ([c_i32, c_i32], c_i64),
    "tante_pack_block_tail_bwd": ([c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp], c_i32),
    "tante_block_tail_bwd": ([c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_f32, C.c_uint64, C.c_uint64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp], c_i32),
    "tante_block_head_bwd": ([c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp], c_i32),
    "tante_block_bwd_fused_supported": ([c_i32, c_i32, c_i32, c_i32, c_i32], c_i32),
    "tante_block_bwd_fused": ([c_vp] * 9 + [c_i32, c_i32, c_i32, C.POINTER(Seq), c_i32, c_f32, C.c_uint64, C.c_uint64, C.c_uint64] + [c_vp] * 6, c_i32),
    "tante_pack_block_train": ([c_vp] * 8 + [c_i32, c_i32, c_vp, c_vp], c_i32),
    "tante_block_fused_train": ([c_vp, c_vp, c_i32, c_i32, c_i32, C.POINTER(Seq), c_i32, c_f32, C.POINTER(BlockTrain), c_vp], c_i32),
    "tante_head_fused_supported": ([c_i32, c_i32], c_i32),
    "tante_head_stream_bytes": ([c_i32], c_i64),
    "tante_pack_head": ([c_vp] * 6 + [c_i32, c_i32, c_vp, c_vp], c_i32),
    "tante_head_fused": ([c_vp, c_i32, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp], c_i32),
    "tante_enc23_supported": ([c_i32], c_i32),
    "tante_enc23_stream_bytes": ([c_i32], c_i64),
    "tante_pack_enc23": ([c_vp] * 4 + [c_i32, c_vp, c_vp], c_i32),
    "tante_enc23_fused": ([c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp], c_i32),
    "tante_enc23_frames": ([c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp], c_i32),
    "tante_im2col": ([c_vp, c_i32, c_i32, c_i64] + [c_i32] * 10 + [c_vp, c_i32, c_vp], c_i32),
    "tante_avgpool_nhwc": ([c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp], c_i32),
    "tante_avgpool_nhwc_bwd": ([c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp], c_i32),
    "tante_col2im_nhwc": ([c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp], c_i32),
    "tante_resize_bilinear": ([c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_i64, c_i32, c_i32, c_i64, c_i64,
                               c_i64, c_i64, c_i32, c_vp, c_i32, c_vp], c_i32),
    "tante_layernorm_affine": ([c_vp, c_i32, c_i64, c_i32, c_f32, c_vp, c_vp, c_vp, c_i32, c_vp], c_i32),
    "tante_spectral_workspace_bytes": ([c_i64, c_i32, c_i32, c_i32, c_i32], c_i64),
    "tante_spectral_layer": ([c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp,
                              c_i64, c_vp], c_i32),
    "tante_wgrad_jobs_ws": ([C.POINTER(WgradJob), c_i32, c_i32, c_vp, c_i64, c_vp], c_i32),
    "tante_spectral_layer_c": ([c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp,
                                c_i64, c_i32, c_vp], c_i32),
    "tante_spectral_layer_x_supported": ([c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32], c_i32),
    "tante_spectral_layer_x": ([c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp,
                               c_i32, c_vp, c_i64, c_vp], c_i32),
    "tante_spectral_layer_bwd": ([c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp,
                                  c_vp, c_i64, c_vp], c_i32),
    "tante_col2im_nhwc_sized": ([c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp], c_i32),
    "tante_resize_bilinear_bwd": ([c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_i64, c_i32, c_i32, c_i64, c_i64,
                                   c_i64, c_i64, c_vp, c_vp], c_i32),
    "tante_cross_attention": ([c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_vp], c_i32),
    "tante_cvit_chain512": ([c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp], c_i32),
    "tante_cvit_chain512_qkv": ([c_vp, c_vp, c_i64, c_vp, c_vp, c_f32, c_f32, c_i64, c_vp, c_vp, c_vp], c_i32),
    "tante_cross_attention_q": ([c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_i64, c_vp], c_i32),
    "tante_cross_attention_bwd": ([c_vp] * 9 + [c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_i64, c_vp], c_i32),
    "tante_layernorm_affine_bwd": ([c_vp, c_i32, c_vp, c_i32, c_vp, c_i64, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp], c_i32),
    "tante_grid_embed_bwd": ([c_vp] * 5 + [c_i64, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp], c_i32),
    "tante_grid_embed": ([c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_f32, c_vp, c_vp], c_i32),
    "tante_fourier_embed": ([c_vp, c_vp, c_i64, c_i32, c_vp, c_vp], c_i32),
    "tante_metric_sums": ([c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i32, c_i32, c_i64, c_i32, c_vp, c_vp], c_i32),
    "tante_mse_grad": ([c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i32, c_i32, c_i64, c_i32, c_f32, c_vp, c_vp], c_i32),
    "tante_sumsq": ([c_vp, c_i64, c_vp, c_vp], c_i32),
    "tante_adamw_step": ([c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_f32, c_vp], c_i32),
    "tante_layernorm_fwd": ([c_vp, c_i64, c_i32, c_f32, c_vp, c_i32, c_vp, c_vp], c_i32),
    "tante_layernorm_bwd": ([c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp], c_i32),
    "tante_act_fwd": ([c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp], c_i32),
    "tante_act_bwd": ([c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp], c_i32),
    "tante_colsum": ([c_vp, c_i32, c_i64, c_i32, c_i64, c_vp, c_i32, c_vp], c_i32),
    "tante_film_pos_fwd": ([c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i64, c_vp, c_vp], c_i32),
    "tante_pos_embed_tmajor": ([c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_i32, c_vp, c_vp], c_i32),
    "tante_film_pos_bwd": ([c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp], c_i32),
    "tante_taylor_bwd": ([c_vp, c_i64, C.POINTER(c_vp), c_i32, C.c_double, c_i32, c_vp, c_i64, c_i32, c_i64, c_i64, c_vp], c_i32),
    "tante_attention_bwd": ([c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, C.POINTER(Seq), c_i32, c_f32, C.c_uint64, c_vp], c_i32),
    "tante_attention_dropout": ([c_vp, c_vp, c_i32, c_i32, c_i32, C.POINTER(Seq), c_i32, c_f32, C.c_uint64, c_vp], c_i32),
    "tante_dropout_add": ([c_vp, c_i32, c_vp, c_f32, C.c_uint64, c_i64, c_vp, c_vp], c_i32),
    "tante_dropout_bwd": ([c_vp, c_f32, C.c_uint64, c_i64, c_vp, c_i32, c_vp], c_i32),
    "tante_axis_mlp_bwd": ([c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp], c_i32),
    "tante_fold_fwd": ([c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp], c_i32),
    "tante_fold_bwd": ([c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp], c_i32),
    "tante_fold_bwd_clear": ([c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp], c_i32),
    "tante_fold_bwd_multi": ([c_vp, c_i32, c_i32, c_vp], c_i32),
    "tante_fold_fwd_multi": ([c_vp, c_i32, c_vp], c_i32),
    "tante_axis_mlp_film_supported": ([c_i64, c_i32, c_i32, c_i64, c_i32], c_i32),
    "tante_axis_mlp_film": ([c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp], c_i32),
    "tante_film_pos_fwd_frames": ([c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_i32, c_vp, c_vp], c_i32),
    "tante_film_pos_bwd_frames": ([c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp], c_i32),
    "tante_film_pos_bwd_frames_acc": ([c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp], c_i32),
    "tante_film_table_bwd": ([c_vp, c_i32, c_i32] + [c_vp] * 6 + [c_vp, c_vp] + [c_vp] * 8 + [c_i32, c_vp], c_i32),
    "tante_pack_block_train_multi": ([c_vp, c_i32, c_i32, c_i32, c_vp], c_i32),
    "tante_pack_block_tail_bwd_multi": ([c_vp, c_i32, c_i32, c_i32, c_vp], c_i32),
    "tante_axis_mlp_bwd_fused_supported": ([c_i32, c_i64], c_i32),
    "tante_axis_mlp_bwd_fused": ([c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp], c_i32),
    "tante_axis_mlp_bwd_fused_ws": ([c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp], c_i32),
    "tante_axis_wgrad": ([c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_i32, c_vp], c_i32),
    "tante_axis_wgrad_workspace_bytes": ([], c_i64),
    "tante_set_seed_mix": ([c_vp], c_i32),
    "tante_axis_wgrad_ws": ([c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp], c_i32),
    "tante_wgrad": ([C.POINTER(RowMat), C.POINTER(RowMat), c_i64, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp], c_i32),
    "tante_wgrad_ws": ([C.POINTER(RowMat), C.POINTER(RowMat), c_i64, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp], c_i32),
    "tante_wgrad_multi_ws": ([c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp], c_i32),
    "tante_wgrad_multi": ([c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp], c_i32),
    "tante_clip_value": ([c_vp, c_i64, c_f32, c_vp], c_i32),
    "tante_rt_reduce_bwd": ([c_vp, c_i32, c_i32, c_vp, c_vp], c_i32),
    "tante_last_error": ([], C.c_char_p),
    "tante_abi_version": ([], c_i32),
    "tante_set_option": ([C.c_char_p, c_i32], c_i32),
    "tante_nan_to_num": ([c_vp, c_vp, c_i64, c_vp], c_i32),
    "tante_head_fused_multi": ([c_i32, c_vp, c_vp, c_vp, c_i32, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64,
                               c_vp], c_i32),
    "tante_head_fused_multi_streams": ([c_i32, c_vp, c_vp, c_vp, c_i32, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64,
                               c_vp], c_i32),
    "tante_spectral_bf16out_supported": ([c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32], c_i32),
    "tante_spectral_layer_bf16out": ([c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp,
                                     c_i64, c_vp], c_i32),
    "tante_get_option": ([C.c_char_p, c_i32], c_i32),
    "tante_attention_masked": ([c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp], c_i32),
    "tante_attention_masked_bwd": ([c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp], c_i32),
    "tante_head_enc_supported": ([c_i32, c_i32], c_i32),
    "tante_head_enc_stream_bytes": ([c_i32], c_i64),
    "tante_head_enc_ws_bytes": ([c_i64], c_i64),
    "tante_pack_head_enc": ([c_vp] * 6 + [c_i32, c_i32, c_vp, c_vp], c_i32),
    "tante_head_enc_fused": ([c_i32, c_vp, c_vp, c_vp, c_i32, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64,
                             c_vp, c_vp, c_vp, c_i64, c_vp], c_i32),
}

# every option name the library looks up (tante_opt in csrc/): tests/test_host_cpu.py checks this list against the sources
LIB_OPTIONS = ("TANTE_ATTN_BWD_HG", "TANTE_ATTN_BWD_NO_SPLIT", "TANTE_ATTN_BWD_VALU", "TANTE_ATTN_FWD_VALU", "TANTE_AXIS_BWD_SMALL_WGS",
               "TANTE_AXIS_BWD_WGS", "TANTE_AXIS_CT", "TANTE_AXIS_FILM", "TANTE_AXIS_GENERIC", "TANTE_AXIS_MFMA", "TANTE_AXIS_NT", "TANTE_AXIS_WGRAD_WGS",
               "TANTE_BLOCK_KERNEL", "TANTE_COLSUM_ROWS", "TANTE_CVIT_CHAIN_TOKENS", "TANTE_ENC23_SPLIT", "TANTE_FILM_BWD_ROWS",
               "TANTE_FS_GROUPS", "TANTE_FS_HALF", "TANTE_FS_SKEW", "TANTE_FS_WAVES", "TANTE_GEMM_NO_LITE", "TANTE_GEMM_SMALLM", "TANTE_GEMM_WGS", "TANTE_HEAD_WAVES",
               "TANTE_IM2COL_TILED", "TANTE_RESIZE_TILED", "TANTE_SPECTRAL_BF16OUT", "TANTE_SPECTRAL_DFT", "TANTE_SPECTRAL_X3", "TANTE_WGRAD_DEEP",
               "TANTE_WGRAD_JOBS", "TANTE_WGRAD_JOBS_WGS", "TANTE_WGRAD_NO_SLAB", "TANTE_WGRAD_NO_TR", "TANTE_WGRAD_REDUCE_NY",
               "TANTE_WGRAD_RG", "TANTE_WGRAD_SLAB_WGS", "TANTE_WGRAD_TR_WGS", "TANTE_WGRAD_WGS", "TANTE_XATTN_GPW", "TANTE_XATTN_VALU",)


_lib = None
ABI_VERSION = 12     # include/tante_hip.h: bumped whenever an entry point is added or changes (round 4: 6 tante_head_enc_*, 7 tante_pos_embed_tmajor + tante_spectral_*bf16out*; round 5: 8 tante_block_bwd_fused, 9 TanteGemm.a_pad; round 6: 10 tante_tail_*, 11 tante_attention_masked_bwd, 12 tante_spectral_layer_x, tante_axis_mlp_film)


def lib():
    """The loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m tante_amd.build` "
                "(or __graft_entry__.build()).  tante_amd has no CPU / eager fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (argtypes, restype) in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError here = header / library mismatch
            fn.argtypes = argtypes
            fn.restype = restype
        got = int(L.tante_abi_version())
        if got != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH}: library ABI {got}, this binding expects {ABI_VERSION} -- a stale build (or a stale variant under "
                               "TANTE_LIB); rebuild with `python -m tante_amd.build`")
        _lib = L
        # The library never reads the environment; the measurement scripts under tools/ drive its A/B switches through TANTE_*
        # environment variables, which are forwarded ONCE here -- only the names the LIBRARY looks up (LIB_OPTIONS: the Python-side
        # switches would only fill its 64-entry table), and a refused name is reported instead of silently measuring the default form.
        import warnings
        for k in LIB_OPTIONS:
            v = os.environ.get(k)
            if v is None:
                continue
            try:
                rc = L.tante_set_option(k.encode(), int(v))
            except ValueError:
                warnings.warn(f"{k}={v!r}: library options are integers; ignored")
                continue
            if rc != 0:
                warnings.warn(f"{k}={v}: tante_set_option refused it ({L.tante_last_error().decode(errors='replace')})")
    return _lib


def set_option(name: str, value: int) -> None:
    """Override one launch heuristic of the library (include/tante_hip.h: tante_set_option), e.g. ("TANTE_FS_GROUPS", 1)."""
    check(lib().tante_set_option(name.encode(), int(value)), "tante_set_option")


def get_option(name: str, default: int = 0) -> int:
    return int(lib().tante_get_option(name.encode(), int(default)))


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().tante_last_error().decode(errors="replace")
        raise RuntimeError(f"libtante_hip {what} failed (rc={rc}): {msg}")
