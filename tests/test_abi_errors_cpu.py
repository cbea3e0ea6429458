"""Argument checks of the C ABI that fail before any device work: callable without a GPU.  Every entry point
returns a negative MZ_E_* code and leaves a message in mz_last_error(); nothing throws across the boundary."""
import ctypes as C

import pytest

from muzero_amd import _lib


def _fails(name, *args, match=None):
    with pytest.raises(_lib.MuZeroAmdError) as e:
        _lib.call(name, *args)
    assert name in str(e.value)
    if match:
        assert match in str(e.value), str(e.value)


def test_null_and_shape_errors_are_reported_not_crashed():
    one = C.c_void_p(16)     # a non-null address that is never dereferenced: the checks come first
    _fails("mz_conv3x3_bf16x3", None, one, one, one, None, one, 1, 96, 96, 96, 256, 2, 1, None)
    _fails("mz_conv3x3_bf16x3", one, one, one, one, None, one, 1, 7, 7, 3, 5, 1, 1, None, match="no kernel")
    _fails("mz_conv_chain_bf16x3", one, None, 0, one, 0, one, None, None, None, 1, None, match="n_layers")
    _fails("mz_conv3x3_f16x2", None, one, one, None, one, 1, 96, 96, 96, 256, 2, 1, None, match="mz_conv3x3_f16x2: null")
    _fails("mz_conv3x3_f16x2", one, one, one, None, one, 1, 7, 7, 3, 5, 1, 1, None, match="no kernel")
    _fails("mz_conv_chain_f16x2", one, None, 0, one, 21, one, None, None, None, 1, None, match="n_layers")
    _fails("mz_split_f32_half", None, one, 1, 1, None, match="bad arguments")
    _fails("mz_join_f32_half", one, None, 1, 1, None, match="bad arguments")
    _fails("mz_weights_fold_pack", one, None, None, None, None, None, 1e-5, 256, 256, 256, 7, 0, 0, one, None, None,
           match="mode")
    _fails("mz_weights_fold_pack", one, None, one, None, None, None, 1e-5, 256, 256, 256, 0, 0, 0, one, None, None,
           match="BatchNorm")
    _fails("mz_weights_fold_pack", one, None, None, None, None, None, 1e-5, 256, 128, 128, 0, 1, 0, one, None, None,
           match="transposed")
    _fails("mz_weights_fold_pack", one, None, None, None, None, None, 1e-5, 250, 96, 96, 0, 0, 0, one, None, None,
           match="fragments")
    _fails("mz_pack_linear", one, None, 601, 250, 1, one, None, None, match="multiple")
    _fails("mz_conv_wgrad_bf16", one, one, 4, 20, 20, 128, 128, one, None, one, 1 << 30, None, match="no kernel")
    _fails("mz_conv_wgrad_bf16", one, one, 4, 24, 24, 128, 128, one, None, one, 16, None, match="workspace")
    _fails("mz_conv_wgrad_bf16", one, one, 4000, 96, 96, 96, 256, one, None, one, 1 << 40, None, match="4 GiB")
    _fails("mz_conv_wgrad_bf16", one, one, 4, 12, 12, 256, 256, one, None, one, 16, None, match="workspace")
    two = (C.c_void_p * 2)(16, 16)
    _fails("mz_conv_wgrad_bf16_multi", two, two, 9, 4, 12, 12, 256, 256, one, None, one, 1 << 30, None, match="n_segments")
    _fails("mz_conv_wgrad_bf16_multi", two, two, 2, 4, 24, 24, 128, 128, one, None, one, 1 << 30, None, match="12x12")
    _fails("mz_conv_wgrad_bf16_multi", (C.c_void_p * 2)(16, None), two, 2, 4, 12, 12, 256, 256, one, None, one, 1 << 30,
           None, match="segment 1")
    _fails("mz_weights_pack_many", two, (C.c_void_p * 2)(16, None), None, 2, 256, 256, 256, None, match="item 1")
    _fails("mz_weights_pack_many", two, two, None, 2, 250, 96, 96, None, match="fragments")
    _fails("mz_conv3x3_bf16_stats", one, one, one, one, None, 4, None, match="null")
    _fails("mz_bn_relu_train_fwd_stats", one, None, one, one, None, None, 1e-5, 0.1, 1, 100, 256, one, one, one, None, 4,
           one, 1 << 30, None, match="stats")
    _fails("mz_conv_chain_mid_bf16", one, one, 0, one, 4, None, match="n_layers")
    _fails("mz_bn_relu_train_fwd", one, None, one, one, None, None, 1e-5, 0.1, 1, 100, 100, one, one, one, one, 1 << 30, None,
           match="C in")
    _fails("mz_bn_relu_train_bwd", one, one, None, one, one, one, 1, 100, 256, one, None, one, one, one, 1 << 30, None,
           match="ReLU mask")
    _fails("mz_replay_sample", one, 0, one, 4, 1.0, None, 0, one, None, None, one, one, 1 << 20, None, match="positions")
    _fails("mz_replay_sample", one, 100, one, 4, 1.0, None, 0, one, None, None, one, one, 8, None, match="workspace")
    _fails("mz_replay_sample_static", one, (16 << 20) + 1, one, 4, 1.0, None, 0, one, None, None, one, None, match="at most")
    _fails("mz_ucb_score", one, one, one, one, one, one, one, one, one, 10, 10, one, None, match="exactly one")
    _fails("mz_gather_stacked_obs", one, one, one, one, 0, 32, 9216, one, None)
    _fails("mz_comm_bcast_weights", None, one, 16, 0, None)
    _fails("mz_ring_to_obs", one, one, 5, 4, 32, 9216, 0, None, match="format")
    # the IEEE-half entry points check as their bf16 namesakes do
    _fails("mz_conv3x3_f16", None, None, 0, None, None, None, None, None, None, None, 1, 12, 12, 256, 256, 1, 1, None,
           match="mz_conv3x3_f16: null tensor")
    _fails("mz_conv3x3_f16", one, None, 0, one, one, None, one, None, None, None, 1, 10, 10, 64, 64, 1, 1, None,
           match="no MFMA kernel")
    _fails("mz_conv_chain_f16", one, None, 0, None, 5, one, None, None, None, 4, None, match="mz_conv_chain_f16: null")
    _fails("mz_conv_chain_mid_f16", one, one, 9, one, 4, None, match="n_layers")
    _fails("mz_conv3x3_f16_rot", None, one, one, one, 4, 1, None, match="null tensor")
    _fails("mz_conv3x3_f16_rot_sel", one, one, 24, None, one, one, 4, 1, None, match="null index")
    _fails("mz_conv3x3_f16_rot_sel", one, one, 12, one, one, one, 4, 1, None, match="16-byte aligned")
    _fails("mz_obs_slot_write", one, C.c_void_p(32), 2, 4, 32, 96, 96, 0, 0, None, None, match="format")
    _fails("mz_weights_fold_pack", one, None, None, None, None, None, 1e-5, 256, 256, 256, 4, 0, 0, one, None, None,
           match="mode")
    _fails("mz_pack_linear", one, None, 601, 256, 3, one, None, None, match="format")
    _fails("mz_step_input_bf16", one, None, None, one, one, 4, 144, 256, None, match="needs table and actions")
    _fails("mz_step_input_bf16", one, one, one, one, one, 4, 144, 24, None, match="power of two")
    _fails("mz_step_input_grad_bf16", None, None, None, 4, 144, 256, 18, one, None, None, None, match="null tensor")
    _fails("mz_step_input_grad_bf16", one, None, None, 4, 144, 256, 18, one, None, one, None, match="dtable needs")
    # the stride-2 data gradient and its packing
    _fails("mz_conv_dgrad_s2_bf16", None, one, one, 4, 48, 48, 256, 128, None, match="null tensor")
    _fails("mz_conv_dgrad_s2_bf16", one, one, one, 4, 96, 96, 96, 256, None, match="no kernel")
    _fails("mz_weights_pack_dgrad_s2", one, 100, 256, one, None, match="multiple of")


def test_size_queries():
    lib = _lib.lib()
    assert lib.mz_packed_conv_elems(256, 256, 0) == 256 * 256 * 9 + 2 * 16 * 512
    assert lib.mz_packed_conv_elems(256, 96, 2) == 256 * 96 * 9
    assert lib.mz_replay_workspace_bytes(1) == 8 * 4 and lib.mz_replay_workspace_bytes(4097) == 8 * (1 + 4 + 3)
    assert lib.mz_replay_static_max_positions() == 16 << 20
    assert lib.mz_bn_workspace_bytes(256) == 256 * 2 * 256 * 8 + 3 * 256 * 4
    assert lib.mz_conv_wgrad_workspace_bytes(0) == 0
    assert lib.mz_conv_wgrad_workspace_bytes_for(4, 20, 20, 128, 128) == 0          # a shape without a kernel
    assert lib.mz_packed_dgrad_s2_elems(128, 256) == 128 * 256 * 9 + 4 * 16 * 512
