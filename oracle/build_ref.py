"""Build recipe for the pieces of the REFERENCE that compile here from their own sources with the tools in the image
(outputs only into oracle/_ref/, which is git-ignored but travels to the GPU box):

  box_overlaps   /root/reference/opencood/utils/box_overlaps.pyx  (Cython + gcc + numpy headers; the reference's own
                 recipe is opencood/utils/setup.py, a plain cythonize)

Not buildable here under the rules (would need stand-in headers): opencood/pcdet_utils/iou3d_nms/src/iou3d_cpu.cpp
includes <cuda.h>/<cuda_runtime_api.h>, which this ROCm image does not have.
The reference sources are read where they lie; nothing is copied into the repository.
"""
from __future__ import annotations

import importlib.machinery
import importlib.util
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
PYX = "/root/reference/opencood/utils/box_overlaps.pyx"


def _so_path() -> str:
    return os.path.join(REF_DIR, "box_overlaps" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build(verbose: bool = False) -> bool:
    """Returns True when oracle/_ref/box_overlaps*.so exists afterwards. Without /root/reference (GPU box) the
    prebuilt file is used as is."""
    so = _so_path()
    if not os.path.exists(PYX):
        return os.path.exists(so)
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(PYX):
        return True
    import numpy
    os.makedirs(REF_DIR, exist_ok=True)
    c_file = os.path.join(REF_DIR, "box_overlaps.c")
    cmds = [
        [sys.executable, "-m", "cython", "-3", PYX, "-o", c_file],
        ["gcc", "-O2", "-shared", "-fPIC", "-fopenmp", "-w", "-I", sysconfig.get_paths()["include"], "-I", numpy.get_include(),
         c_file, "-o", so],
    ]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    os.remove(c_file)
    return True


def load_box_overlaps():
    """The compiled reference module (None when it has not been built)."""
    so = _so_path()
    if not os.path.exists(so):
        return None
    if "box_overlaps" in sys.modules:
        return sys.modules["box_overlaps"]
    loader = importlib.machinery.ExtensionFileLoader("box_overlaps", so)
    spec = importlib.util.spec_from_loader("box_overlaps", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    sys.modules["box_overlaps"] = mod
    return mod


if __name__ == "__main__":
    print("built" if build(verbose=True) else "reference not available")
