"""ctypes binding of librgqa_hip.so (include/rgqa.h). There is no CPU fallback: a missing library is an error."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RGQA_LIB") or os.path.join(_HERE, "lib", "librgqa_hip.so")      # RGQA_LIB: development A/B of two builds

PREC_F32, PREC_BF16, PREC_BF16X3, PREC_BF16X3_FWD = 0, 1, 2, 3


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab_size", "hidden", "heads", "inter", "max_pos", "type_vocab", "l_layers",
                                         "x_layers", "r_layers", "feat_dim", "pos_dim", "num_answers", "precision")] + \
               [("ln_eps", C.c_float), ("hidden_dropout", C.c_float), ("attn_dropout", C.c_float), ("arch", C.c_int32), ("emb_dim", C.c_int32)]


_vp, _sz, _i, _f, _u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_uint64

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/rgqa.h
SIGNATURES = {
    "rgqa_version": [],
    "rgqa_debug_set": [_i, _i],
    "rgqa_tokenizer_create": [C.c_char_p, _i, C.POINTER(_vp)],
    "rgqa_tokenizer_destroy": [_vp],
    "rgqa_tokenizer_vocab_size": [_vp, C.POINTER(C.c_int64)],
    "rgqa_tokenizer_encode": [_vp, C.POINTER(C.c_char_p), _i, _i, _vp, _vp, _vp, _vp],
    "rgqa_engine_create": [C.POINTER(Config), C.POINTER(_vp)],
    "rgqa_engine_destroy": [_vp],
    "rgqa_engine_arena_elems": [_vp, C.POINTER(_sz)],
    "rgqa_engine_num_params": [_vp, C.POINTER(_i)],
    "rgqa_engine_param_info": [_vp, _i, C.c_char_p, _sz, C.POINTER(_sz), C.POINTER(C.c_int64), C.POINTER(_i), C.POINTER(_i)],
    "rgqa_engine_dead_range": [_vp, C.POINTER(_sz), C.POINTER(_sz)],
    "rgqa_engine_workspace_bytes": [_vp, _i, _i, _i, C.POINTER(_sz)],
    "rgqa_engine_bind": [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i],
    "rgqa_engine_sync_weights": [_vp, _vp],
    "rgqa_engine_sync_transposed": [_vp, _vp],
    "rgqa_engine_forward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _u64, _vp],
    "rgqa_engine_loss_backward": [_vp, _vp, _i, _vp, _f, _i, _vp],
    "rgqa_engine_backward": [_vp, _vp, _i, _i, _vp],
    "rgqa_engine_backward_pooled": [_vp, _vp, _i, _i, _vp],
    "rgqa_engine_get_activation": [_vp, C.c_char_p, _vp, _sz, _vp],
    "rgqa_engine_get_cross_attention": [_vp, _i, _i, _vp, _sz, _vp],
    "rgqa_engine_set_lengths": [_vp, _vp, _i],
    "rgqa_engine_set_input_grads": [_vp, _vp, _vp],
    "rgqa_engine_set_grad_sumsq_slots": [_vp, _vp, _i],
    "rgqa_engine_num_grad_segments": [_vp, C.POINTER(_i)],
    "rgqa_engine_grad_segment": [_vp, _i, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_i)],
    "rgqa_engine_wait_grad_event": [_vp, _i, _vp],
    "rgqa_engine_set_weight_event": [_vp, _i, _vp],
    "rgqa_engine_set_backward_event": [_vp, _vp],
    "rgqa_engine_num_weight_segments": [_vp, C.POINTER(_i)],
    "rgqa_set_side_stream": [_i, _vp],
    "rgqa_engine_profile": [_vp, _i],
    "rgqa_engine_profile_read": [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), _i],
    "rgqa_engine_profile_blocks": [_vp, _vp, _vp, _i],
    "rgqa_engine_profile_operand_bytes": [_vp, _vp, _i],
    "rgqa_probe_gemm": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "rgqa_grad_sumsq": [_vp, _sz, _vp, _vp, _i, _vp],
    "rgqa_clip_scale": [_vp, _sz, _vp, _f, _vp],
    "rgqa_bertadam_step": [_vp, _vp, _vp, _vp, _vp, _i, _sz, _f, _f, _f, _f, _f, _vp, _f, _f, _vp],
    "rgqa_cast_bf16": [_vp, _vp, _sz, _vp],
    "rgqa_split_f32": [_vp, _vp, _sz, _vp],
    "rgqa_unsplit_f32": [_vp, _vp, _sz, _vp],
    "rgqa_sum_bf16_parts": [_vp, _sz, _i, _vp, _sz, _vp],
    "rgqa_sum_parts": [_vp, _i, _sz, _i, _vp, _sz, _vp, _vp, _vp],
    "rgqa_peer_comm_create": [_i, _i, _sz, C.POINTER(_vp)],
    "rgqa_peer_comm_stage": [_vp, C.POINTER(_vp), C.POINTER(_sz)],
    "rgqa_peer_comm_export": [_vp, _vp],
    "rgqa_peer_comm_connect": [_vp, _vp],
    "rgqa_peer_comm_is_fine_grained": [_vp],
    "rgqa_peer_comm_destroy": [_vp],
    "rgqa_peer_pull": [_vp, _sz, _sz, _vp, _sz, _vp],
    "rgqa_mixup_gather": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "rgqa_mixup_perturb": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "rgqa_mixup_weighted_sum": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "rgqa_scale_rows": [_vp, _vp, _i, _i, _i, _i, _vp],
    "rgqa_op_linear": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "rgqa_op_linear_ex": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "rgqa_op_linear_kn": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, C.c_size_t, _vp],
    "rgqa_op_linear_splitk": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, C.c_size_t, _vp],
    "rgqa_op_matmul_tn": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "rgqa_op_matmul_tn_group": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "rgqa_op_layernorm": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp],
    "rgqa_op_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "rgqa_op_attention": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "rgqa_op_attention_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "rgqa_op_bce": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "rgqa_batch_prepare": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "rgqa_host_gather_rows": [_vp, _sz, _sz, _vp, _i, _vp, _i],
    "rgqa_score_rows": [_vp, _i, _i, _i, C.c_float, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rgqa_last_error_string": [],
}
_RESTYPES = {"rgqa_last_error_string": C.c_char_p, "rgqa_engine_destroy": None, "rgqa_tokenizer_destroy": None, "rgqa_peer_comm_destroy": None}

_lib = None


def load():
    """Loads the library (once). Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; it must be the HIP runtime already resident when our library is
    # mapped, otherwise the process ends up with two runtimes and torch's device pointers / streams mean nothing to ours
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("rgqa_amd: %s not found. Build it with `python -m rgqa_amd.build` "
                           "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        if os.environ.get("RGQA_LIB") and not hasattr(lib, name):
            continue              # an older build named by RGQA_LIB (tools/ab_lib.sh): entry points added since are simply absent
        fn = getattr(lib, name)   # AttributeError here = header / library mismatch
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().rgqa_last_error_string()
        raise RuntimeError("rgqa: %s (code %d)" % (msg.decode("utf-8", "replace") if msg else "unknown error", rc))


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
