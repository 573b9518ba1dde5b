"""CPU: libdcd_hip.so loads and exports every entry point include/dcd_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "dcd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dcd_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for s in ("dcd_mg_step", "dcd_mg_step_adversary", "dcd_gae_returns", "dcd_episode_scores", "dcd_env_return"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    import dcd_hip
    assert os.path.exists(dcd_hip.LIB_PATH), "build with make -C dcd-isaac_amd/csrc"
    lib = ctypes.CDLL(dcd_hip.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.dcd_abi_version.restype = ctypes.c_int
    assert lib.dcd_abi_version() == dcd_hip.ABI_VERSION == 2


def test_python_structs_match_header_layout():
    import dcd_hip
    assert ctypes.sizeof(dcd_hip.MgConfig) == 72          # 14 x 4 + 8 + 2 x 4
    assert ctypes.sizeof(dcd_hip.StudentObs) == 32 and ctypes.sizeof(dcd_hip.AdvObs) == 32
    assert ctypes.sizeof(dcd_hip.StepOut) == 32 + 5 * 8 + 32 + 3 * 8


def test_product_fails_loudly_without_gpu():
    import torch
    import dcd_hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dcd_hip.DcdHipError, match="no CPU path"):
        dcd_hip.MultiGridEngine(4)


def test_new_struct_layouts_and_argument_checks_need_no_gpu():
    """DcdPolicyAgent / DcdStudentNet mirror the header, and the entry points added in round 2 reject bad arguments BEFORE any
    HIP call (NULL pointers, unsupported geometry), returning non-zero with a message — callable without a device."""
    import dcd_hip
    from dcd_hip.policy import PolicyAgent, StudentNet
    assert ctypes.sizeof(StudentNet) == 12 * 8 + 8 * 4
    assert ctypes.sizeof(PolicyAgent) == 15 * 8 + 8 + 4 * 4
    L = ctypes.CDLL(dcd_hip.LIB_PATH)
    L.dcd_last_error.restype = ctypes.c_char_p
    net = StudentNet()
    for k, v in dict(V=15, F=128, E=10, D=28, H=256, A1=32, C1=32, NA=169).items():
        setattr(net, k, v)
    null = ctypes.c_void_p(None)
    calls = {
        "dcd_grid_delta_rows": lambda: L.dcd_grid_delta_rows(ctypes.byref(net), null, null, ctypes.c_int64(0), null, null, null, null, null,
                                                             ctypes.c_int32(1), null, null, ctypes.c_int32(4), null),
        "dcd_grid_tile_gemm": lambda: L.dcd_grid_tile_gemm(null, null, null, null, null, ctypes.c_int32(1), ctypes.c_int32(128), ctypes.c_int32(1024), null),
        "dcd_policy_grid_rest_features": lambda: L.dcd_policy_grid_rest_features(ctypes.byref(net), null, null, ctypes.c_int32(50), null, null, null,
                                                                                 ctypes.c_int32(320), ctypes.c_int32(4), null),
        "dcd_policy_wide_pack_heads": lambda: L.dcd_policy_wide_pack_heads(ctypes.byref(net), null, null),
        "dcd_policy_wide_cell_heads": lambda: L.dcd_policy_wide_cell_heads(ctypes.byref(net), null, null, null, null, null, null, null, null, null, null,
                                                                           ctypes.c_uint32(0), ctypes.c_uint32(0), null, ctypes.c_int64(0),
                                                                           ctypes.c_int32(0), ctypes.c_int32(4), null),
        "dcd_policy_features_u8_pair": lambda: L.dcd_policy_features_u8_pair(null, null, ctypes.c_int32(416), ctypes.c_int32(4), null),
        "dcd_policy_cell_heads_pair": lambda: L.dcd_policy_cell_heads_pair(null, null, ctypes.c_int32(4), null),
    }
    for name, call in calls.items():
        assert call() != 0, name
        assert L.dcd_last_error(), name
    assert L.dcd_grid_delta_tiles(ctypes.c_int32(15), ctypes.c_int32(4096)) == (9 * 4096 + 169 * 127 + 127) // 128
    assert L.dcd_grid_delta_tiles(ctypes.c_int32(40), ctypes.c_int32(4096)) == 0       # grid too large for the carried projection
    assert L.dcd_policy_wide_packed_heads_floats(ctypes.byref(net)) == 0              # head weights not set -> refused


def test_ppo_entry_points_refuse_unsupported_shapes_without_a_device():
    """dcd_ppo_* (round 4) are built for the registered student only and say so before any HIP call."""
    import dcd_hip
    from dcd_hip.policy import StudentNet
    L = ctypes.CDLL(dcd_hip.LIB_PATH)
    L.dcd_last_error.restype = ctypes.c_char_p
    assert L.dcd_ppo_features_partial_floats() == 512 and L.dcd_ppo_heads_partial_floats() == 384
    net = StudentNet()
    for k, v in dict(V=15, F=128, E=10, D=28, H=256, A1=32, C1=32, NA=169).items():      # the teacher's geometry
        setattr(net, k, v)
    one = ctypes.c_void_p(16)
    assert L.dcd_ppo_features_fwd(ctypes.byref(net), one, one, 0, one, 152, ctypes.c_int64(4), None) != 0
    assert b"registered student" in L.dcd_last_error()
    assert L.dcd_ppo_heads_fwd(ctypes.byref(net), one, one, one, one, one, one, ctypes.c_int64(4), None) != 0
    assert b"at most 8 actions" in L.dcd_last_error()
    for k, v in dict(V=5, F=16, E=5, D=4, NA=7).items():
        setattr(net, k, v)
    assert L.dcd_ppo_features_fwd(ctypes.byref(net), None, one, 0, one, 152, ctypes.c_int64(4), None) != 0          # NULL image
    assert L.dcd_ppo_features_bwd(ctypes.byref(net), one, one, 0, one, one, 152, ctypes.c_int64(4), 24, one, None) != 0  # 24 % 16
    assert b"multiple of 16" in L.dcd_last_error()
