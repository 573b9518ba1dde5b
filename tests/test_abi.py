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
