"""Arithmetic mode for the bench tools of the layers around the path: GENCOMM_TOOL_ARITH=3 python tools/shell_bench.py runs the general
convolutions on the opt-in two-term split kernels, GENCOMM_TOOL_ARITH=1 on the exact-fp32 kernels (include/gencomm_hip.h
GENCOMM_MODE_ARITH); unset = the library default (three-term f16-pipe arithmetic on the hot path and, since round 5, in the general
convolutions wherever the shape allows)."""
import os


def apply_env_modes() -> None:
    v = os.environ.get("GENCOMM_TOOL_ARITH")
    if v:
        from gencomm_amd import _lib
        _lib.check(_lib.lib().gencomm_set_mode(_lib.MODE_ARITH, int(v)), "gencomm_set_mode")
        print(f"[mode] GENCOMM_MODE_ARITH = {int(v)}", flush=True)
