"""ctypes binding of libcoskad_hip.so (the C-ABI HIP library, include/coskad_hip.h).

There is NO fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List

_HERE = os.path.dirname(os.path.abspath(__file__))
# COSKAD_LIB: another build of the same library (tools/ab_fused.sh: timing-only A/B variants); the shipped path otherwise
LIB_PATH = os.environ.get("COSKAD_LIB") or os.path.join(_HERE, "libcoskad_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "coskad_hip.h")

_lib = None


class CoskadHipError(RuntimeError):
    pass


def header_symbols(path: str = HEADER_PATH) -> List[str]:
    """Function names declared in include/coskad_hip.h."""
    with open(path) as f:
        src = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return re.findall(r"\b(coskad_\w+)\s*\(", src)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CoskadHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C coskad_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # torch ships its own libamdhip64; load it FIRST so this library binds to the same HIP runtime
        # (two runtimes in one process do not share devices/streams).
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.coskad_last_error.restype = ctypes.c_char_p
        _lib.coskad_abi_version.restype = ctypes.c_int
    return _lib


# Optional per-call timing probe (bench.py): PROBE = {"name": <entry point>, "filter": callable or None,
# "events": []}.  When set, matching calls are bracketed by HIP events on torch's current stream (the
# stream every kernel of this library is enqueued on).
PROBE = None


def call(name: str, *args, tag=None) -> None:
    """Call an `int coskad_*(...)` entry point; raise on a non-zero return."""
    fn = getattr(lib(), name)
    fn.restype = ctypes.c_int
    if PROBE is not None and PROBE["name"] == name and (PROBE.get("tag") is None or PROBE["tag"] == tag):
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        PROBE["events"].append((e0, e1))
    else:
        rc = fn(*args)
    if rc != 0:
        msg = lib().coskad_last_error().decode(errors="replace")
        raise CoskadHipError(f"{name} failed ({rc}): {msg}")


def ptr(t) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def i32(x: int) -> ctypes.c_int:
    return ctypes.c_int(int(x))


def f32(x: float) -> ctypes.c_float:
    return ctypes.c_float(float(x))
