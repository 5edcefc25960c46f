import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with -m gpu")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


@pytest.fixture
def modes():
    """Setter for the library modes of include/gencomm_hip.h (gencomm_set_mode), restored after the test:
    ``modes(arith="f32", sampler="direct", tile_want=1, xcd=0, enh_fuse=0, conv8h_mask=-1, dataflow=1)``."""
    from gencomm_amd import _lib
    l = _lib.lib()
    keys = {"arith": _lib.MODE_ARITH, "sampler": _lib.MODE_SAMPLER, "tile_want": _lib.MODE_TILE_WANT,
            "enh_fuse": _lib.MODE_ENH_FUSE, "conv8h_mask": _lib.MODE_CONV8H_MASK, "xcd": _lib.MODE_XCD_REMAP,
            "dataflow": _lib.MODE_DATAFLOW, "resfuse_emu": _lib.MODE_RESFUSE_EMU, "tile8": _lib.MODE_TILE8, "bwd_streams": _lib.MODE_BWD_STREAMS, "persist": _lib.MODE_PERSIST}
    names = {"split": 0, "f32": 1, "bf16": 2, "split2": 3, "latent": 2, "direct": 1, "auto": 0}
    prev = {k: l.gencomm_get_mode(k) for k in keys.values()}

    def set_(**kw):
        for name, value in kw.items():
            _lib.check(l.gencomm_set_mode(keys[name], names.get(value, value)), "gencomm_set_mode")

    yield set_
    for k, v in prev.items():
        _lib.check(l.gencomm_set_mode(k, v), "gencomm_set_mode")
