"""The C-ABI library loads and exports every symbol include/gencomm_hip.h declares; host-side
plan queries (no GPU compute) behave. CPU only."""
import ctypes
import os
import re

import pytest

from gencomm_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(REPO, "include", "gencomm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gencomm_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.lib()


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for sym in _declared_symbols():
        assert hasattr(raw, sym), sym
    assert lib.gencomm_abi_version() == _lib.ABI_VERSION


def test_unet_param_table_matches_module(lib):
    from gencomm_amd import GenComm, synth
    for C in (8, 16, 64, 128):
        gen = GenComm(synth.default_gencomm_cfg(C, 3))
        named = dict(gen.denoiser.named_parameters())
        table = _lib.unet_param_table(C, 2, 2, 0)
        assert sorted(n for n, _, _ in table) == sorted(named)  # every UNet parameter is consumed
        off = 0
        for name, numel, o in table:
            assert named[name].numel() == numel and o == off
            off += numel
        assert off == lib.gencomm_unet_raw_floats(C, 2, 2, 0)
        assert lib.gencomm_unet_prepared_floats(C, 2, 2, 0, 3) > off


def test_enhancer_param_table_matches_module(lib):
    from gencomm_amd import Enhancer
    enh = Enhancer(64, [8, 8], 4)
    named = dict(enh.named_parameters())
    table = _lib.enhancer_param_table(64)
    live = {n for n, _, _ in table}
    assert live == {k for k in named if k.startswith(("block_1.norm", "block_1.mlp", "split_attn"))}
    for name, numel, _ in table:
        assert named[name].numel() == numel


def test_argument_errors_are_status_codes_not_exits(lib):
    assert lib.gencomm_unet_raw_floats(7, 2, 2, 0) == -1
    assert b"multiple of 8" in lib.gencomm_last_error()
    assert lib.gencomm_denoise_workspace_bytes(4, 64, 201, 704, 2, 2, 0) == -1  # odd H with a downsample
    assert b"even" in lib.gencomm_last_error()
    assert lib.gencomm_denoise_workspace_bytes(4, 64, 200, 704, 2, 2, 0) > 0
    # null pointers are rejected before anything is launched
    assert lib.gencomm_unet_fwd(None, None, None, None, 0, 1, 64, 16, 16, 2, 2, 0, 3, None, 0, None) == 1
    assert lib.gencomm_warp_attfuse_fwd(None, None, None, None, 1, 1, 8, 4, 4, None) == 1


def test_cpu_tensors_fail_loudly():
    import torch
    from gencomm_amd import AttFusion, Enhancer, GenComm, synth
    gen = GenComm(synth.default_gencomm_cfg(16, 3)).eval()
    with pytest.raises(_lib.GenCommHipError):
        with torch.no_grad():
            gen(torch.zeros(2, 16, 8, 8), torch.zeros(2, 2, 8, 8), torch.tensor([2]))
    with pytest.raises(_lib.GenCommHipError):
        with torch.no_grad():
            Enhancer(16, [8, 8], 4)(torch.zeros(2, 16, 8, 8), None, torch.tensor([2]))
    with pytest.raises(_lib.GenCommHipError):
        with torch.no_grad():
            AttFusion(16)(torch.zeros(2, 16, 8, 8), torch.tensor([2]), torch.zeros(1, 5, 5, 2, 3))


def test_attnblock_params_are_enumerated_when_present(lib):
    from gencomm_amd import GenComm, synth
    cfg = synth.default_gencomm_cfg(8, 3)
    cfg["model"]["attn_resolutions"] = [64]  # nominal 128 -> 64 at level 1
    gen = GenComm(cfg)
    assert gen.denoiser.attn_mask == 0b10
    named = dict(gen.denoiser.named_parameters())
    table = _lib.unet_param_table(8, 2, 2, 0b10)
    assert sorted(n for n, _, _ in table) == sorted(named)
    assert lib.gencomm_unet_raw_floats(8, 2, 2, 0b100) == -1  # bit beyond the number of levels
