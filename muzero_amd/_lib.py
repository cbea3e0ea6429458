"""ctypes binding of libmuzero_amd.so (the C ABI declared in include/muzero_amd.h).

There is no CPU fallback: if the shared library is missing or a call fails, an
exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MUZERO_AMD_LIB points at another build of the same library (kernel A/B comparisons)
LIB_PATH = os.environ.get("MUZERO_AMD_LIB") or os.path.join(_HERE, "csrc", "libmuzero_amd.so")


class MuZeroAmdError(RuntimeError):
    pass


class TreeLayout(C.Structure):
    _fields_ = [("B", C.c_int32), ("A", C.c_int32), ("max_nodes", C.c_int32), ("reserved", C.c_int32),
                ("total_bytes", C.c_int64)] + [(n, C.c_int64) for n in (
                    "child", "visit", "value_sum", "prior", "root_prior64", "reward", "num_nodes",
                    "root_visit", "root_value_sum", "minmax", "root_is_f64", "path_node", "path_action",
                    "path_len", "leaf_parent", "leaf_action", "status")]


_P = C.c_void_p


class ChainLayer(C.Structure):
    _fields_ = [("wpk", C.c_void_p), ("bias", C.c_void_p), ("relu", C.c_int32), ("add_residual", C.c_int32)]

class ChainLayerX3(C.Structure):
    _fields_ = [("wpk_hi", C.c_void_p), ("wpk_lo", C.c_void_p), ("bias", C.c_void_p), ("relu", C.c_int32),
                ("add_residual", C.c_int32)]


class WgradLayer(C.Structure):
    _fields_ = [("grad_out", C.POINTER(C.c_void_p)), ("x", C.POINTER(C.c_void_p)), ("n_segments", C.c_int32),
                ("B", C.c_int32), ("dw", C.c_void_p), ("dbias", C.c_void_p)]


_SIGNATURES = {
    "mz_version": (C.c_int, []),
    "mz_last_error": (C.c_char_p, []),
    "mz_tree_layout_query": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(TreeLayout)]),
    "mz_tree_root_init": (C.c_int, [_P, C.POINTER(TreeLayout), _P, _P, C.c_float, C.c_double, _P]),
    "mz_tree_select": (C.c_int, [_P, C.POINTER(TreeLayout), _P, _P, C.c_int32, _P]),
    "mz_tree_expand_backup": (C.c_int, [_P, C.POINTER(TreeLayout), _P, _P, _P, C.c_int32, C.c_double, _P]),
    "mz_tree_expand_backup_select": (C.c_int, [_P, C.POINTER(TreeLayout), _P, _P, _P, C.c_int32, C.c_double, _P, _P,
                                               C.c_int32, _P]),
    "mz_ucb_score": (C.c_int, [_P] * 9 + [C.c_int32, C.c_int32, _P, _P]),
    "mz_tree_readout": (C.c_int, [_P, C.POINTER(TreeLayout), _P, _P, _P, _P]),
    "mz_select_action": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_double, _P, _P, _P]),
    "mz_conv3x3_f32": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P, _P, _P] + [C.c_int32] * 7 + [_P]),
    "mz_conv3x3_bf16_info": (C.c_int, [C.c_int32] * 5 + [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mz_conv3x3_bf16": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P, _P, _P, _P] + [C.c_int32] * 7 + [_P]),
    "mz_conv_chain_bf16": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, _P, _P, _P, C.c_int32, _P]),
    "mz_conv_chain_mid_bf16": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P]),
    "mz_conv3x3_f16x2": (C.c_int, [_P] * 5 + [C.c_int32] * 7 + [_P]),
    "mz_conv_chain_f16x2": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, _P, _P, _P, C.c_int32, _P]),
    "mz_split_f32_half": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "mz_join_f32_half": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "mz_conv3x3_f16": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P, _P, _P, _P] + [C.c_int32] * 7 + [_P]),
    "mz_conv_chain_f16": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, _P, _P, _P, C.c_int32, _P]),
    "mz_conv_chain_mid_f16": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P]),
    "mz_conv3x3_f16_rot": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    "mz_conv3x3_f16_rot_sel": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    "mz_split_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "mz_join_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "mz_conv3x3_bf16x3": (C.c_int, [_P] * 6 + [C.c_int32] * 7 + [_P]),
    "mz_conv_chain_bf16x3": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, _P, _P, _P, C.c_int32, _P]),
    "mz_spatial_mean_f32": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mz_heads": (C.c_int, [_P] + [C.c_int32] * 5 + [_P] * 10),
    "mz_support_to_scalar": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    "mz_policy_softmax": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    "mz_nchw_to_nhwc_f32": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mz_nhwc_to_nchw_f32": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mz_support_ce": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "mz_nstep_targets": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_double, _P, _P, _P, _P]),
    "mz_ring_push": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    "mz_ring_to_obs": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mz_conv3x3_bf16_rot": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    "mz_conv3x3_bf16_rot_sel": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    "mz_obs_slot_write": (C.c_int, [_P, _P] + [C.c_int32] * 7 + [_P, _P]),
    "mz_gather_stacked_obs": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "mz_prepare_observation": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mz_packed_conv_elems": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "mz_weights_fold_pack": (C.c_int, [_P] * 6 + [C.c_double] + [C.c_int32] * 6 + [_P, _P, _P]),
    "mz_pack_linear": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "mz_weights_pack_many": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mz_action_vectors": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    "mz_comm_unique_id": (C.c_int, [_P]),
    "mz_comm_init": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "mz_comm_bcast_weights": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    "mz_comm_destroy": (C.c_int, [_P]),
    "mz_conv_wgrad_workspace_bytes": (C.c_int64, [C.c_int32]),
    "mz_conv_wgrad_workspace_bytes_for": (C.c_int64, [C.c_int32] * 5),
    "mz_conv_wgrad_group_workspace_bytes": (C.c_int64, [C.POINTER(C.c_int32), C.c_int32]),
    "mz_conv_wgrad_bf16_group": (C.c_int, [C.POINTER(WgradLayer), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "mz_step_input_bf16": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 3 + [C.c_void_p]),
    "mz_step_input_grad_bf16": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_void_p] * 4),
    "mz_packed_dgrad_s2_elems": (C.c_int64, [C.c_int32, C.c_int32]),
    "mz_weights_pack_dgrad_s2": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mz_conv_dgrad_s2_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p]),
    "mz_conv_wgrad_bf16": (C.c_int, [_P, _P] + [C.c_int32] * 5 + [_P, _P, _P, C.c_int64, _P]),
    "mz_conv_wgrad_bf16_multi": (C.c_int, [_P, _P] + [C.c_int32] * 6 + [_P, _P, _P, C.c_int64, _P]),
    "mz_bn_workspace_bytes": (C.c_int64, [C.c_int32]),
    "mz_bn_relu_train_fwd": (C.c_int, [_P] * 6 + [C.c_double, C.c_double, C.c_int32, C.c_int64, C.c_int32, _P, _P, _P, _P,
                                               C.c_int64, _P]),
    "mz_bn_relu_train_fwd_stats": (C.c_int, [_P] * 6 + [C.c_double, C.c_double, C.c_int32, C.c_int64, C.c_int32, _P, _P, _P,
                                                     _P, C.c_int32, _P, C.c_int64, _P]),
    "mz_conv3x3_bf16_stats": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, _P]),
    "mz_bn_relu_train_bwd": (C.c_int, [_P] * 6 + [C.c_int32, C.c_int64, C.c_int32, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "mz_bn_relu_train_bwd_acc": (C.c_int, [_P] * 6 + [C.c_int32, C.c_int64, C.c_int32, _P, _P, _P, _P, C.c_int32, _P,
                                           C.c_int64, _P]),
    "mz_replay_workspace_bytes": (C.c_int64, [C.c_int64]),
    "mz_replay_sample": (C.c_int, [_P, C.c_int64, _P, C.c_int32, C.c_double, _P, C.c_int32, _P, _P, _P, _P, _P,
                                   C.c_int64, _P]),
    "mz_replay_static_max_positions": (C.c_int64, []),
    "mz_replay_sample_static": (C.c_int, [_P, C.c_int64, _P, C.c_int32, C.c_double, _P, C.c_int32, _P, _P, _P, _P,
                                          _P]),
    "mz_replay_update": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int32, C.c_double, _P]),
}

_lib = None


def declared_symbols():
    return sorted(_SIGNATURES)


def lib():
    """Load the library once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        # torch ships its own HIP runtime: load it first so that this library binds to
        # the same runtime instance (two runtimes in one process do not share devices)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise MuZeroAmdError(
                f"{LIB_PATH} not found: build it with `make -C muzero_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def call(name, *args):
    """Call an int-returning entry point and turn a negative code into an exception."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        msg = lib().mz_last_error().decode(errors="replace")
        raise MuZeroAmdError(f"{name} failed ({rc}): {msg}")


def expect(t, name, dtype=None, shape=None, cuda=True, contiguous=True):
    """Validate a tensor before its address goes to a kernel; ValueError otherwise (kernels index
    by the shapes their arguments claim, so a mismatch would be an out-of-bounds access)."""
    if t is None:
        raise ValueError(f"{name}: tensor required")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: shape {tuple(t.shape)}, expected {tuple(shape)}")
    if cuda and not t.is_cuda:
        raise ValueError(f"{name}: must live on the GPU")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def ptr(t):
    """Device (or host) address of a torch tensor, or NULL for None."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
