"""ctypes binding of libdcd_hip.so — the C ABI declared in include/dcd_hip.h.

This is the only bridge between the host Python (envs/, algos/, level_replay/) and the HIP
kernels.  There is NO fallback: if the shared library is missing or a tensor is not a
contiguous CUDA(=HIP) tensor of the right dtype the call raises.  PyTorch is used for device
memory and streams only — every pointer handed to the library is `tensor.data_ptr()` and every
launch goes to `torch.cuda.current_stream()`.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdcd_hip.so")
CSRC = os.path.join(_HERE, "..", "csrc")


class DcdHipError(RuntimeError):
    pass


class MgConfig(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("size", C.c_int32), ("agent_view_size", C.c_int32),
                ("max_steps", C.c_int32), ("max_episode_steps", C.c_int32), ("n_clutter", C.c_int32),
                ("resample_n_clutter", C.c_int32), ("choose_goal_last", C.c_int32),
                ("see_through_walls", C.c_int32), ("random_z_dim", C.c_int32),
                ("editor_actions", C.c_int32), ("fixed_environment", C.c_int32),
                ("goal_noise", C.c_float), ("engine_seed", C.c_uint32), ("env_index_base", C.c_int64),
                ("replicas", C.c_int32), ("reserved", C.c_int32)]


class StudentObs(C.Structure):
    _fields_ = [("image_u8", C.c_void_p), ("dir_u8", C.c_void_p),
                ("image_f32", C.c_void_p), ("dir_f32", C.c_void_p)]


class AdvObs(C.Structure):
    _fields_ = [("image_u8", C.c_void_p), ("image_f32", C.c_void_p),
                ("time_step", C.c_void_p), ("random_z", C.c_void_p)]


class StepOut(C.Structure):
    _fields_ = [("obs", StudentObs), ("reward", C.c_void_p), ("done", C.c_void_p), ("info", C.c_void_p),
                ("ep_ret", C.c_void_p), ("ep_len", C.c_void_p), ("term_obs", StudentObs),
                ("masks", C.c_void_p), ("bad_masks", C.c_void_p), ("cliffhanger_masks", C.c_void_p)]


# DcdEpisodeRec as a numpy-compatible layout (8 x 4 bytes)
EPISODE_REC_WORDS = 8

ERR_BITS = {
    1 << 0: "unknown student action (multigrid.py:939)",
    1 << 1: "Position passed to step_adversary is outside the grid.",
    1 << 2: "Trying to place agent at empty start position.",
    1 << 3: "malformed level encoding",
    1 << 4: "no free cell for random placement",
    1 << 5: "agent stepped outside the grid",
    1 << 6: "Wrong object in agent start position.",
}

STEP_RESET_RANDOM = 1
STEP_LAST = 2
INFO_EPISODE, INFO_TRUNC_KEY, INFO_TRUNCATED, INFO_CLIFFHANGER = 1, 2, 4, 8

SCORE_STRATEGY = {"value_l1": 0, "positive_value_loss": 1, "signed_value_loss": 2, "gae": 2,
                  "policy_entropy": 3, "least_confidence": 4, "min_margin": 5, "one_step_td_error": 6,
                  "uniform": 7, "grounded_signed_value_loss": 8, "grounded_positive_value_loss": 8}

ABI_VERSION = 2
_lib = None


def build(verbose=False):
    """Compile libdcd_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import subprocess
    cmd = ["make", "-C", CSRC]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DcdHipError(
                f"{LIB_PATH} is missing: build it with `make -C dcd-isaac_amd/csrc` "
                "(or __graft_entry__.build()).  There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.dcd_last_error.restype = C.c_char_p
        if _lib.dcd_abi_version() != ABI_VERSION:
            raise DcdHipError("libdcd_hip.so ABI version mismatch")
    return _lib


def check(rc):
    if rc != 0:
        raise DcdHipError(lib().dcd_last_error().decode())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None, numel=None, name="tensor"):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise DcdHipError(f"{name}: expected a CUDA/HIP tensor, got {type(t).__name__}"
                          f"{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)}")
    if not t.is_contiguous():
        raise DcdHipError(f"{name}: tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise DcdHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if numel is not None and t.numel() != numel:
        raise DcdHipError(f"{name}: expected {numel} elements, got {t.numel()} (shape {tuple(t.shape)})")
    return C.c_void_p(t.data_ptr())


from .engine import MultiGridEngine  # noqa: E402,F401
from . import scans  # noqa: E402,F401
