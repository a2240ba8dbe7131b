"""ctypes binding of libmuse_hip.so (C ABI: include/muse_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, this raises.
torch is used only for device memory (tensor.data_ptr()) and the current HIP stream.
"""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MM_LIB') or os.path.join(HERE, 'libmuse_hip.so')      # MM_LIB: tools-only A/B of two builds
HEADER_PATH = os.path.join(HERE, '..', 'include', 'muse_hip.h')

MM_OK = 0
MM_NOISE_NONE, MM_NOISE_GUMBEL, MM_NOISE_UNIFORM, MM_NOISE_PHILOX = 0, 1, 2, 3
MM_GEN_NO_FUSED_SAMPLING = 1
MM_GEN_CAN_REMASK = 2

c_i64, c_int, c_f32, c_vp, c_u64, c_u32, c_sz = C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t


class MuseHipError(RuntimeError):
    pass


class AttnWeights(C.Structure):
    _fields_ = [(n, c_vp) for n in ('ln_gamma', 'ln_beta', 'w_q', 'w_kv', 'w_out', 'null_k', 'null_v', 'q_scale', 'k_scale', 'w_q_scale', 'w_kv_scale', 'w_out_scale', 'w_q_ln', 'ln_c1', 'ln_c2')]


class FFWeights(C.Structure):
    _fields_ = [(n, c_vp) for n in ('ln1_gamma', 'ln1_beta', 'w1', 'ln2_gamma', 'ln2_beta', 'w2', 'w2_folded', 'ln2_c1', 'ln2_c2', 'w1_scale', 'w2_scale', 'w1_ln', 'ln1_c1', 'ln1_c2', 'w1_terms_geglu')]


class LayerWeights(C.Structure):
    _fields_ = [('self_attn', AttnWeights), ('cross_attn', AttnWeights), ('ff', FFWeights)]


MM_LN_FOLD_MAX_RATIO = 4.0      # include/muse_hip.h


class TransformerDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('dim', 'depth', 'heads', 'dim_head', 'ff_inner', 'ff_inner_padded', 'seq_len',
                                         'num_tokens', 'vocab_rows', 'dim_out', 'text_dim', 'self_cond')] + \
               [('token_emb', c_vp), ('pos_emb', c_vp), ('text_proj', c_vp), ('layers', C.POINTER(LayerWeights)),
                ('final_gamma', c_vp), ('final_beta', c_vp), ('to_logits', c_vp), ('self_cond_ff', FFWeights), ('logits_wmean', c_vp), ('logits_wcov', c_vp),
                ('split_products', C.c_int32), ('fp8', C.c_int32), ('split_alpha', c_f32), ('ln_fold_off', C.c_int32), ('ln_probe', c_vp), ('logits_wsub', c_vp), ('logits_wsub_rows', C.c_int32)]


class TrainAttn(C.Structure):
    _fields_ = [(n, c_vp) for n in ('gamma', 'beta', 'to_q', 'to_kv', 'q_scale', 'k_scale', 'null_kv', 'to_out',
                                    'd_gamma', 'd_to_q', 'd_to_kv', 'd_q_scale', 'd_k_scale', 'd_null_kv', 'd_to_out')]


class TrainFF(C.Structure):
    _fields_ = [(n, c_vp) for n in ('g1', 'b1', 'w1', 'g2', 'b2', 'w2', 'd_g1', 'd_w1', 'd_g2', 'd_w2')]


class TrainLayer(C.Structure):
    _fields_ = [('sa', TrainAttn), ('ca', TrainAttn), ('ff', TrainFF)]


class TrainDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('dim', 'depth', 'heads', 'ff_inner', 'seq_len', 'vocab_rows', 'dim_out', 'text_dim')] + \
               [(n, c_vp) for n in ('token_emb', 'pos_emb', 'text_proj', 'final_gamma', 'final_beta', 'to_logits',
                                    'd_token_emb', 'd_pos_emb', 'd_text_proj', 'd_final_gamma', 'd_to_logits')] + [('layers', C.POINTER(TrainLayer))]


class VaeLayer(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('kind', 'cout', 'k', 'groups')] + [('w', c_vp * 4), ('b', c_vp * 3), ('gn_g', c_vp * 2), ('gn_b', c_vp * 2)]


class VaeDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('channels', 'encoded_dim', 'bits', 'n_enc', 'n_dec', 'half')] + \
               [('enc', C.POINTER(VaeLayer)), ('dec', C.POINTER(VaeLayer)), ('lfq_wi', c_vp), ('lfq_bi', c_vp), ('lfq_wo', c_vp), ('lfq_bo', c_vp),
                ('alpha', c_f32), ('reserved', C.c_int32)]


class GenerateParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('batch', 'n', 'timesteps', 'k_keep', 'noise_kind', 'nc', 'L', 'flags')] + \
               [('cond_scale', c_f32), ('pad0', c_f32), ('seed', c_u64), ('row_offset', c_u64),
                ('mask_counts', C.POINTER(C.c_int32)), ('temperatures', C.POINTER(c_f32)),
                ('text_embeds', c_vp), ('cond_ids', c_vp), ('noise', c_vp), ('ids', c_vp), ('scores', c_vp),
                ('trace_masked_ids', c_vp), ('trace_ids', c_vp), ('trace_scores', c_vp), ('status', c_vp),
                ('critic', c_vp), ('critic_head_w', c_vp), ('critic_head_b', c_vp), ('critic_noise', c_vp), ('critic_noise_scale', c_f32),
                ('pad1', c_f32), ('critic_workspace', c_vp), ('critic_workspace_bytes', c_sz), ('seed_dev', c_vp)]


# name -> (restype, argtypes); every symbol declared in include/muse_hip.h must appear here (tests check both ways)
SIGNATURES = {
    'mm_abi_version': (c_int, []),
    'mm_last_error': (C.c_char_p, []),
    'mm_device_check': (c_int, []),
    'mm_gemm_bf16': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_i64, c_int, c_vp]),
    'mm_gemm_split': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_int, c_f32, c_vp, c_i64, c_vp]),
    'mm_gemm_split_geglu': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_int, c_f32, c_vp, c_i64, c_vp]),
    'mm_gemm_cfg_logits': (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_i64, c_f32]),
    'mm_embed': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp]),
    'mm_layernorm': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_i64]),
    'mm_geglu_ln': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_i64]),
    'mm_gemm_geglu': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_i64]),
    'mm_layernorm_inner': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_i64]),
    'mm_attend': (c_int, [c_vp] + [c_vp, c_i64, c_i64, c_i64] * 4 + [c_int, c_int, c_int, c_int, c_vp, c_i64, c_int,
                                                                  c_vp, c_vp, c_vp, c_vp, c_f32, c_int]),
    'mm_mask_step': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_i64, c_vp]),
    'mm_sample_rows': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_f32, c_int, c_vp, c_i64, c_u64, c_u64,
                               c_u32, c_vp, c_vp, c_vp, c_vp]),
    'mm_fused_z': (c_f32, [c_int, c_int, c_f32]),
    'mm_fused_threshold_workspace_bytes': (c_sz, [c_int, c_int]),
    'mm_fused_threshold': (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_f32, c_vp, c_vp, c_f32, c_vp, c_vp]),
    'mm_gemm_cfg_logits_fused': (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_f32, c_vp, c_vp, c_vp]),
    'mm_fused_emit': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    'mm_fused_sample': (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_f32, c_int, c_vp, c_i64, c_u64, c_u64, c_u32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int]),
    'mm_ce_loss': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_i64, c_vp, c_vp]),
    'mm_bce_loss': (c_int, [c_vp, c_vp, c_vp, c_int, c_vp]),
    'mm_quantize_e4m3_rows': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp]),
    'mm_quantize_act_e4m3': (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_int, c_int, c_vp, c_vp]),
    'mm_gemm_fp8': (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_int, c_vp]),
    'mm_vq_nearest': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_int, c_int, c_vp, c_vp]),
    'mm_vq_gather': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp]),
    'mm_gemm_wgrad_splits': (c_int, [c_int, c_int, c_int]),
    'mm_gemm_wgrad': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    'mm_gemm_wgrad_tn_splits': (c_int, [c_int, c_int, c_int]),
    'mm_gemm_wgrad_tn_prefer': (c_int, [c_int, c_int, c_int, c_i64, c_i64]),
    'mm_gemm_wgrad_tn': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp]),
    'mm_transpose_bf16': (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64]),
    'mm_f32_to_bf16': (c_int, [c_vp, c_vp, c_vp, c_i64]),
    'mm_colsum_f32': (c_int, [c_vp, c_vp, c_int, c_int, c_vp]),
    'mm_ln_bwd_workspace_floats': (c_i64, [c_int, c_int]),
    'mm_layernorm_bwd': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_vp, c_i64, c_int, c_vp, c_vp]),
    'mm_geglu_ln_bwd': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_vp]),
    'mm_ce_bwd': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_f32, c_vp, c_i64]),
    'mm_bce_head_bwd': (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_i64, c_vp, c_vp]),
    'mm_embed_bwd_workspace_bytes': (c_sz, [c_int, c_int, c_int]),
    'mm_embed_bwd': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_sz]),
    'mm_scatter_rows_bf16': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    'mm_sum_parts_bf16': (c_int, [c_vp, c_vp, c_int, c_i64, c_vp]),
    'mm_attention_bwd': (c_int, [c_vp] + [c_vp, c_i64, c_i64, c_i64] * 8 + [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_f32]),
    'mm_qk_norm_bwd_blocks': (c_i64, [c_i64]),
    'mm_qk_norm_bwd': (c_int, [c_vp, c_vp, c_i64, c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_vp]),
    'mm_train_step_workspace_bytes': (c_sz, [C.POINTER(TrainDesc), c_int, c_int, c_int, c_int]),
    'mm_train_step': (c_int, [C.POINTER(TrainDesc), c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_sz]),
    'mm_philox_uniform': (c_int, [c_vp, c_u64, c_u64, c_u32, c_int, c_int, c_vp]),
    'mm_conv2d_nhwc': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int] + [c_int] * 12 + [c_vp, c_int, c_vp, c_vp, c_int]),
    'mm_conv2d_nhwc_f16': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int] + [c_int] * 12 + [c_vp, c_int, c_vp, c_vp, c_int, c_f32]),
    'mm_conv2d_nhwc_half': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int] + [c_int] * 12 + [c_vp, c_int, c_vp, c_vp, c_int, c_f32]),
    'mm_conv2d_nhwc_terms': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int] + [c_int] * 12 + [c_vp, c_int, c_vp, c_vp, c_int, c_f32, c_int]),
    'mm_glu_nhwc': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    'mm_groupnorm_nhwc': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp]),
    'mm_lfq_decode': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    'mm_lfq_encode': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'mm_nchw_f32_to_nhwc8_bf16': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    'mm_nhwc_bf16_to_nchw_f32': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    'mm_vae_create': (c_int, [C.POINTER(VaeDesc), C.POINTER(c_vp)]),
    'mm_vae_destroy': (None, [c_vp]),
    'mm_vae_decode_workspace_bytes': (c_sz, [c_vp, c_int, c_int, c_int]),
    'mm_vae_encode_workspace_bytes': (c_sz, [c_vp, c_int, c_int, c_int]),
    'mm_vae_decode_from_ids': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_sz]),
    'mm_vae_encode': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_sz]),
    'mm_f32_gemm': (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_int, c_vp]),
    'mm_f32_conv2d_nhwc': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int] + [c_int] * 12 + [c_vp, c_int, c_vp, c_vp, c_int]),
    'mm_f32_layernorm': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_i64]),
    'mm_f32_geglu': (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_i64]),
    'mm_f32_cfg_combine': (c_int, [c_vp, c_vp, c_vp, c_f32, c_i64, c_vp]),
    'mm_f32_embed': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_i64]),
    'mm_f32_text_mask': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    'mm_attend_terms': (c_int, [c_vp] + [c_vp, c_i64, c_i64, c_i64] * 4 + [c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_f32]),
    'mm_f32_attend': (c_int, [c_vp] + [c_vp, c_i64, c_i64, c_i64] * 4 + [c_int, c_int, c_int, c_int, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_f32, c_int]),
    'mm_f32_glu_nhwc': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    'mm_f32_groupnorm_nhwc': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp]),
    'mm_f32_lfq_decode': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    'mm_f32_lfq_bits': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    'mm_f32_nchw_to_nhwc': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    'mm_f32_nhwc_to_nchw': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    'mm_split_rows': (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp]),
    'mm_cfg_mix': (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_f32, c_vp]),
    'mm_transformer_create': (c_int, [C.POINTER(TransformerDesc), C.POINTER(c_vp)]),
    'mm_transformer_destroy': (None, [c_vp]),
    'mm_context_workspace_bytes': (c_sz, [c_vp, c_int, c_int]),
    'mm_transformer_context': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_sz]),
    'mm_transformer_workspace_bytes': (c_sz, [c_vp, c_int, c_int, c_int]),
    'mm_transformer_forward': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_sz]),
    'mm_fused_quantile_rank': (c_int, [c_int, c_int, c_int]),
    'mm_fused_quantile': (c_int, [c_vp, c_vp, C.c_int64, c_int, c_int, c_int, c_vp]),
    'mm_cross_attention_block_workspace_bytes': (c_sz, [c_vp, c_int, c_int, c_int]),
    'mm_cross_attention_block': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_sz]),
    'mm_generate_critic_workspace_bytes': (c_sz, [c_vp, c_int, c_int, c_int, c_int]),
    'mm_generate_workspace_bytes': (c_sz, [c_vp, c_int, c_int, c_int, c_int]),
    'mm_generate': (c_int, [c_vp, c_vp, C.POINTER(GenerateParams), c_vp, c_sz]),
    'mm_comm_unique_id': (c_int, [c_vp]),
    'mm_comm_create': (c_int, [c_vp, c_int, c_int, C.POINTER(c_vp)]),
    'mm_comm_destroy': (None, [c_vp]),
    'mm_comm_world': (c_int, [c_vp]),
    'mm_comm_rank': (c_int, [c_vp]),
    'mm_allgather_ids_workspace_bytes': (c_sz, [c_vp, c_i64]),
    'mm_allgather_ids': (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_sz]),
    'mm_debug_set': (c_int, [c_int]),
    'mm_debug_set2': (c_int, [c_int]),
    'mm_debug_trace': (c_int, [c_vp, c_int]),
    'mm_debug_trace_count': (c_int, []),
    'mm_debug_capture': (c_int, [c_vp, c_sz, c_int, c_int]),
    'mm_profile_enable': (c_int, [c_int]),
    'mm_profile_read': (c_int, [c_int, C.POINTER(c_i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


def header_symbols():
    """Function names declared in include/muse_hip.h."""
    with open(HEADER_PATH) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    return sorted(set(re.findall(r'\b(mm_[a-z0-9_]+)\s*\(', text)))


def lib():
    """Loads libmuse_hip.so (once).  No fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MuseHipError(f'{LIB_PATH} not found: build it with `python -m muse_maskgit_pytorch_amd.build` '
                               '(hipcc, gfx950).  There is no CPU / eager fallback in this package.')
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.mm_abi_version() != 9:
            raise MuseHipError('libmuse_hip ABI version mismatch')
        if os.environ.get('MM_DEBUG'):      # tools / A-B runs only: kernel-selection bits (see muse_hip_internal.h)
            l.mm_debug_set(int(os.environ['MM_DEBUG'], 0))
        if os.environ.get('MM_DEBUG2'):
            l.mm_debug_set2(int(os.environ['MM_DEBUG2'], 0))
        _lib = l
    return _lib


def check(rc, what=''):
    if rc != MM_OK:
        msg = lib().mm_last_error()
        raise MuseHipError(f'{what} failed ({rc}): {msg.decode() if msg else ""}')


def stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


_device_ok = False


def require_device():
    """Fail loudly unless a gfx950 device is current."""
    global _device_ok
    if not _device_ok:
        import torch
        if not torch.cuda.is_available():
            raise MuseHipError('no HIP device visible: the muse_maskgit_pytorch_amd hot path only runs on MI355X (gfx950)')
        check(lib().mm_device_check(), 'mm_device_check')
        _device_ok = True
