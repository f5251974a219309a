"""ctypes binding of libpcompanion_hip.so (include/pcompanion_hip.h).

There is NO fallback: if the shared object is missing or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpcompanion_hip.so")
# Developer A/B runs (scripts/dev/ab_*.sh) point PC_DEV_LIB at another BUILD of the same library instead of copying it over the
# product file (a copied-over .so looks up to date to build.py and would silently become what tests and bench.py run).  Said
# out loud on stderr; never set by the tests, bench.py or the driver.
if os.environ.get("PC_DEV_LIB"):
    import sys as _sys
    LIB_PATH = os.path.abspath(os.environ["PC_DEV_LIB"])
    print(f"[p_companion_amd] PC_DEV_LIB: loading {LIB_PATH} instead of the in-tree library (developer A/B run)", file=_sys.stderr, flush=True)

c_f32p = ctypes.c_void_p
c_i32p = ctypes.c_void_p
PC_MAX_SEG = 4
D, H, HEADS, L = 128, 256, 4, 64

_ERR = {-1: "PC_EINVAL (null pointer / bad size)", -2: "PC_ESHAPE (unsupported dimension/alignment)",
        -3: "PC_EWORKSPACE (workspace too small)",
        -4: "PC_EBATCHNORM (a BatchNorm call group of one row in training mode)",
        -5: "PC_ECOMM (RCCL could not be loaded, or a communicator call failed)"}


class HipKernelError(RuntimeError):
    pass


class Dropout(ctypes.Structure):
    """pc_dropout: p == 0 (the zero-initialised default) = off."""
    _fields_ = [("p", ctypes.c_float), ("seed", ctypes.c_uint64), ("offset", ctypes.c_uint64)]


class P2VTensors(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "w0", "b0", "gamma", "beta", "w3", "b3", "w5", "b5", "in_proj_w", "in_proj_b", "out_proj_w",
        "out_proj_b", "running_mean", "running_var", "num_batches_tracked")] + [("dropout", Dropout), ("dim", ctypes.c_int)]


class Segments(ctypes.Structure):
    _fields_ = [("nseg", ctypes.c_int), ("start", ctypes.c_int * (PC_MAX_SEG + 1)),
                ("count", ctypes.c_int * PC_MAX_SEG), ("weighted_row", ctypes.c_int), ("weight", ctypes.c_float),
                ("row_weight", ctypes.c_void_p), ("row_weight_start", ctypes.c_int), ("row_weight_rows", ctypes.c_int)]


class FfnSaved(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("h0", "a2", "bn_mean", "bn_invstd", "bn_scale", "bn_shift", "a1")]


class AttnSaved(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("q", "qt", "probs", "c", "sp", "ctx")]


class JointTensors(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "product_table", "enc_w", "enc_b", "dec_w", "dec_b", "typ_w", "typ_b", "itm_w", "itm_b",
        "query_types", "comp_types")] + [("dropout", Dropout)]


class JointSaved(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("h", "c", "pi", "tp")]


_vp, _i, _sz, _f, _d, _u64, _i64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_float,
                                    ctypes.c_double, ctypes.c_uint64, ctypes.c_int64)
_P = ctypes.POINTER

class AdamFused(ctypes.Structure):
    """pc_adam_fused (ABI 8): torch.optim.Adam riding in the Product2Vec step's last gradient launch."""
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("n", ctypes.c_size_t), ("step_count", ctypes.c_void_p), ("t", ctypes.c_int64),
                ("lr", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double)]


PC_OPT_SIDE_QUEUE = 1      # pc_set_option: the fused Product2Vec step's side queue (include/pcompanion_hip.h)
PC_OPT_FUSED_OUT_CHAIN = 5           # ... and the out-projection's forward chain in front of it in the same launch (default 1)
PC_OPT_FUSED_LOSS = 4                # ... the triplet hinge inside the attention backward's first launch (default 1)
PC_OPT_BN_FINALIZE_SIDE = 3          # ... the BatchNorm-backward finalize of the fused Product2Vec step on the side queue (default 0: on the step's own)
PC_OPT_SORTED_TABLE_GRADIENTS = 2    # ... the [T,64] table gradients of the fused joint step through the sorted form wherever it fits (default 1)

# name -> (restype, argtypes).  Must list every symbol include/pcompanion_hip.h declares
# (tests/test_abi.py parses the header and checks this table and the .so against it).
SIGNATURES = {
    "pc_abi_version": (_i, []),
    "pc_build_flags": (ctypes.c_uint, []),
    "pc_set_option": (_i, [_i, _i]),
    "pc_get_option": (_i, [_i, ctypes.POINTER(ctypes.c_int)]),
    "pc_release_device_state": (_i, []),
    "pc_p2v_ffn_workspace_bytes": (_sz, [_i]),
    "pc_p2v_ffn_forward_train": (_i, [_P(P2VTensors), _vp, _vp, _i, _P(Segments), _i, _vp, _P(FfnSaved), _vp, _sz, _vp]),
    "pc_p2v_ffn_forward_eval": (_i, [_P(P2VTensors), _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "pc_p2v_ffn_backward": (_i, [_P(P2VTensors), _P(P2VTensors), _vp, _vp, _i, _P(Segments), _vp, _P(FfnSaved),
                                 _vp, _i, _vp, _sz, _vp]),
    "pc_p2v_attention_workspace_bytes": (_sz, [_i, _i]),
    "pc_p2v_attention_workspace_bytes_dim": (_sz, [_i, _i, _i]),
    "pc_p2v_attention_forward": (_i, [_P(P2VTensors), _vp, _vp, _i, _i, _vp, _P(AttnSaved), _vp, _sz, _vp]),
    "pc_p2v_attention_backward": (_i, [_P(P2VTensors), _P(P2VTensors), _vp, _vp, _i, _i, _vp, _P(AttnSaved), _vp,
                                       _vp, _i, _vp, _sz, _vp]),
    "pc_p2v_triplet_loss": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pc_p2v_triplet_loss_dim": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pc_adam_step": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _d, _d, _d, _d, _vp]),
    "pc_adam_step_at": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _i64, _d, _d, _d, _d, _vp]),
    "pc_p2v_train_step_workspace_bytes": (_sz, [_i, _i, _i]),
    "pc_p2v_train_step_workspace_bytes_dim": (_sz, [_i, _i, _i, _i]),
    "pc_p2v_train_step": (_i, [_P(P2VTensors), _P(P2VTensors), _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp,
                               _vp, _vp, _vp, _vp, _sz, _vp]),
    "pc_p2v_train_step_compact": (_i, [_P(P2VTensors), _P(P2VTensors), _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _f,
                                       _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pc_p2v_train_step_unique_adam": (_i, [_P(P2VTensors), _P(P2VTensors), _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i,
                                           _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _P(AdamFused), _vp]),
    "pc_p2v_concat_step_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "pc_p2v_train_step_unique_rows": (_i, [_P(P2VTensors), _P(P2VTensors), _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i,
                                           _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _P(AdamFused), _vp]),
    "pc_p2v_train_step_unique": (_i, [_P(P2VTensors), _P(P2VTensors), _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i,
                                      _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pc_build_similarity_batch_unique_scratch_bytes": (_sz, [_i, _i]),
    "pc_build_similarity_batch_unique": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _u64, _u64, _i, _vp, _vp, _vp,
                                              _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pc_p2v_train_step_compact_sync": (_i, [_P(P2VTensors), _P(P2VTensors), _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i,
                                            _f, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pc_build_similarity_batch_compact": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _u64, _u64, _vp, _vp, _vp,
                                               _vp, _vp, _vp, _vp]),
    "pc_profile_create": (_i, [_i, _P(ctypes.c_void_p)]),
    "pc_profile_destroy": (_i, [_vp]),
    "pc_profile_reset": (_i, [_vp]),
    "pc_profile_set_kinds": (_i, [_vp, ctypes.c_uint]),
    "pc_profile_summary": (_i, [_vp, _i, _P(ctypes.c_int), _P(ctypes.c_double), _P(ctypes.c_double)]),
    "pc_build_similarity_batch": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _u64, _u64, _vp, _vp, _vp,
                                       _vp, _vp]),
    "pc_build_complementary_batch": (_i, [_vp, _i, _vp, _vp, _i, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pc_mt_state_bytes": (_sz, []),
    "pc_mt_seed": (_i, [_vp, _u64]),
    "pc_mt_getrandbits": (ctypes.c_uint32, [_vp, _i]),
    "pc_mt_randbelow": (_u64, [_vp, _u64]),
    "pc_mt_shuffle": (_i, [_vp, _vp, _i64]),
    "pc_mt_negative_samples": (_i, [_vp, ctypes.c_int32, _vp, _vp, _vp, _i64, _i, _vp]),
    "pc_joint_workspace_bytes": (_sz, [_i, _i, _i]),
    "pc_joint_forward": (_i, [_P(JointTensors), _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _P(JointSaved), _vp, _sz, _vp]),
    "pc_joint_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "pc_joint_loss_dim": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "pc_expand_type_grad": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "pc_joint_backward": (_i, [_P(JointTensors), _P(JointTensors), _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp,
                               _P(JointSaved), _vp, _sz, _vp]),
    "pc_joint_train_step": (_i, [_P(JointTensors), _P(JointTensors), _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f,
                                 _f, _vp, _vp, _vp, _sz, _vp]),
    "pc_joint_fused_workspace_bytes": (_sz, [_i, _i, _i]),
    "pc_joint_fused_supported": (_i, [_i, _i, _f]),
    "pc_joint_fused_touched": (_i, [_vp, _sz, _i, _i, _i, _P(ctypes.c_void_p), _P(ctypes.c_void_p), _P(ctypes.c_void_p)]),
    "pc_joint_fused_step": (_i, [_P(JointTensors), _P(JointTensors), _P(JointTensors), _P(JointTensors), _vp, _d, _d, _d, _d,
                                 _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pc_joint_fused_step_pairs": (_i, [_P(JointTensors), _P(JointTensors), _P(JointTensors), _P(JointTensors), _vp, _d, _d, _d, _d,
                                       _vp, _vp, _vp, _i, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f,
                                       _vp, _vp, _vp, _vp, _sz, _vp]),
    "pc_joint_train_epoch": (_i, [_P(JointTensors), _P(JointTensors), _P(JointTensors), _P(JointTensors), _vp, _d, _d, _d, _d,
                                  _vp, _i64, _vp, _vp, _i, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f,
                                  _vp, _vp, _vp, _vp, _sz, _vp]),
    # ABI 6: the data-parallel exchange slot (a pc_exchange_fn travels as a plain address) and the library's RCCL communicator
    "pc_exchange_adam": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _i64, _vp, _d, _d, _d, _d, _vp]),
    "pc_joint_train_epoch_dp": (_i, [_P(JointTensors), _P(JointTensors), _vp, _vp, _vp, _vp, _sz, _vp, _i64, _vp, _d, _d, _d, _d,
                                     _vp, _vp, _vp, _i64, _vp, _vp, _i, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i,
                                     _i, _f, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    # ABI 8: the optimizer sharded over the replicas (a pc_exchange_plan travels by reference)
    "pc_exchange_adam_plan": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _i64, _vp, _d, _d, _d, _d, _vp]),
    "pc_joint_train_epoch_plan": (_i, [_P(JointTensors), _P(JointTensors), _vp, _vp, _vp, _vp, _sz, _vp, _i64, _vp, _d, _d, _d, _d,
                                       _vp, _vp, _i64, _vp, _vp, _i, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i,
                                       _i, _f, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pc_rccl_reduce_scatter_mean": (_i, [_vp, _vp, _sz, _vp]),
    "pc_rccl_all_gather": (_i, [_vp, _vp, _sz, _vp]),
    "pc_rccl_available": (_i, []),
    "pc_rccl_unique_id": (_i, [_vp]),
    "pc_rccl_comm_create": (_i, [_vp, _i, _i, _P(ctypes.c_void_p)]),
    "pc_rccl_comm_destroy": (_i, [_vp]),
    "pc_rccl_allreduce_mean": (_i, [_vp, _vp, _sz, _vp]),
    "pc_rccl_allreduce_sum_f64": (_i, [_vp, _vp, _sz, _vp]),
    "pc_rccl_alltoall": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "pc_rccl_comm_stats": (_i, [_vp, _P(ctypes.c_int64), _P(ctypes.c_int64)]),
    "pc_rccl_last_error": (ctypes.c_char_p, []),
    "pc_linear_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "pc_linear_backward_input": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "pc_linear_backward_weight_workspace_bytes": (_sz, [_i, _i, _i]),
    "pc_linear_backward_weight": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "pc_topk_rows": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "pc_hit_rank": (_i, [_vp, _i, _i, _vp, _vp]),
    "pc_cosine_rows": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "pc_cosine_rows_dim": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pc_retrieve_topk": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "pc_retrieve_topk_dim": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "pc_hadamard_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "pc_hadamard_forward_dim": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pc_hadamard_backward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "pc_hadamard_backward_dim": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "pc_gather_rows": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "pc_scatter_add_rows": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "pc_scatter_add_rows_small": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    "pc_scatter_rows": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "pc_act_backward": (_i, [_vp, _vp, _sz, _i, _vp, _vp]),
    "pc_sample_negatives_zipf": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _u64, _u64, _vp, _i, _vp, _vp, _vp, _vp]),
    "pc_epoch_permutation": (_i, [_i, _u64, _u64, _vp, _vp]),
    "pc_gen_types": (_i, [_i64, _i, _u64, _vp, _vp]),
    "pc_gen_features": (_i, [_i64, _i64, _i64, _i, _i, _u64, _vp, _vp]),
    "pc_gen_degrees": (_i, [_i64, ctypes.c_double, _i, ctypes.c_double, _u64, _vp, _vp, _vp]),
    "pc_scan_scratch_bytes": (_sz, [_i64]),
    "pc_exclusive_scan_i32": (_i, [_vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "pc_gen_coview": (_i, [_i64, _i, _u64, _vp, _vp, _vp, _vp]),
    "pc_gen_similarity": (_i, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pc_gen_complementary": (_i, [_i64, _i, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pc_shuffle_rows_i32": (_i, [_vp, _i, _i, _u64, _u64, _vp, _vp]),
    "pc_epoch_plan": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pc_shard_bucket": (_i, [_P(ctypes.c_void_p), _P(ctypes.c_int), _P(ctypes.c_void_p), _P(ctypes.c_int), _P(ctypes.c_void_p),
                             _i, _i, _i, _vp, _vp, _vp, _vp]),
    "pc_shard_bucket_hot": (_i, [_P(ctypes.c_void_p), _P(ctypes.c_int), _P(ctypes.c_void_p), _P(ctypes.c_int), _P(ctypes.c_void_p),
                                 _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "pc_dropout_hidden": (_i, [_vp, _sz, _P(Dropout), _vp, _vp]),
    "pc_check_indices": (_i, [_P(ctypes.c_void_p), _P(ctypes.c_int), _P(ctypes.c_int), _P(ctypes.c_int), _i, _vp, _vp]),
}

_lib = None


def lib():
    """The loaded shared object.  Raises if it has not been built (python -m p_companion_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipKernelError(
                f"{LIB_PATH} not found: build it with `python -m p_companion_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # torch ships its own libamdhip64: it must be resident BEFORE this library is mapped so
        # that both resolve to ONE HIP runtime (two runtimes => hipErrorNoDevice on torch's streams)
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError = symbol missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc == 0:
        return
    if rc == -4:
        # what nn.BatchNorm1d raises in the reference (torch/nn/functional.py _verify_batch_size)
        raise ValueError(f"Expected more than 1 value per channel when training, got input size torch.Size([1, {H}])")
    if rc == -5:
        raise HipKernelError(f"{what}: {_ERR[rc]}: {lib().pc_rccl_last_error().decode(errors='replace')}")
    if rc < 0:
        raise HipKernelError(f"{what}: {_ERR.get(rc, rc)}")
    raise HipKernelError(f"{what}: hipError_t {rc}")
