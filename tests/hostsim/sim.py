"""Builds and binds tests/hostsim/hostsim.cpp: the DEVICE env logic compiled for the CPU.

Test-only (see hostsim.cpp header).  Reuses the oracle's numpy call wrapper with the sim_ symbols.
"""
import ctypes as C
import os
import subprocess

from oracle import orc

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libhostsim.so")
_lib = None


def build(sanitize=False, pitch=16):
    """pitch: record pitch the env logic is compiled for (16: grids up to 15; 32 / 64 / 128: the large mazes)."""
    src = os.path.join(_HERE, "hostsim.cpp")
    hdr = os.path.join(_HERE, "..", "..", "dcd-isaac_amd", "csrc", "mg_env.h")
    out = _LIB if not sanitize else os.path.join(_HERE, "libhostsim_asan.so")
    if pitch != 16:
        out = out.replace(".so", f"_p{pitch}.so")
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return out
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if sanitize else ["-O2"]
    subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                           f"-DDCD_PITCH={pitch}", *flags, "-o", out, src])
    return out


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.sim_mg_create.restype = C.c_void_p
        _lib.sim_mg_poll_errors.restype = C.c_uint32
    return _lib


class HostSimMg(orc.OracleMg):
    _prefix = "sim_mg_"

    @staticmethod
    def _library():
        return lib()


_big = {}


def big_lib(pitch):
    if pitch not in _big:
        l = C.CDLL(build(pitch=pitch))
        l.sim_mg_create.restype = C.c_void_p
        l.sim_mg_poll_errors.restype = C.c_uint32
        _big[pitch] = l
    return _big[pitch]


def host_sim_class(pitch):
    """HostSimMg over the env logic compiled for a wider record pitch (grids above 15 x 15)."""
    return type(f"HostSimMgP{pitch}", (HostSimMg,), {"_library": staticmethod(lambda: big_lib(pitch))})
