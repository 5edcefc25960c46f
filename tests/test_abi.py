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
    assert lib.gencomm_head_loss(None, None, None, None, None, None, None, None, None, None, 2, 2, 64, 128, 2, None, 0.7853,
                                 2.0, 2.0, 0.25, 2.0, 3.0, 2.0, 0.2, 2, None) == 1
    assert b"null pointer" in lib.gencomm_last_error()


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


def test_integration_md_example_matches_the_binding():
    """INTEGRATION.md section 2 shows a maintainer how to call the C ABI: every `lib.<fn>(...)` call and every `argtypes`
    list in it must have exactly as many entries as the binding the package itself uses (a stale example shifts arguments
    and corrupts the call)."""
    import re
    from gencomm_amd import _lib
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    block = text[text.index("```python\nimport ctypes as C, torch"):]
    block = block[:block.index("```", 10)]
    assert f"gencomm_abi_version() == {_lib.ABI_VERSION}" in block

    def top_level_args(s):
        depth, n, seen = 0, 0, False
        for ch in s:
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
            elif ch == "," and depth == 0:
                n += 1
            if not ch.isspace():
                seen = True
        return n + 1 if seen else 0

    checked = 0
    for m in re.finditer(r"lib\.(gencomm_\w+)\.argtypes = \[([^\]]*)\]", block):
        assert top_level_args(m.group(2)) == len(_lib._SIGNATURES[m.group(1)][1]), m.group(1)
        checked += 1
    for m in re.finditer(r"lib\.(gencomm_\w+)\(", block):
        name, start = m.group(1), m.end()
        if name == "gencomm_abi_version" or name == "gencomm_last_error":
            continue
        depth, i = 1, start
        while depth:
            depth += {"(": 1, ")": -1}.get(block[i], 0)
            i += 1
        nargs = top_level_args(block[start:i - 1])
        want = len(_lib._SIGNATURES[name][1])
        if name == "gencomm_unet_num_params" or nargs == want:
            assert nargs == want, (name, nargs, want)
            checked += 1
        else:
            raise AssertionError((name, nargs, want))
    assert checked >= 8


def test_code_object_has_no_packed_fp32_instructions(tmp_path):
    """Disassemble the gfx950 code object of the built library: no v_pk_{fma,mul,add}_f32 anywhere (see gencomm_build_info
    in include/gencomm_hip.h), and the matrix-core instructions the design rests on are there."""
    import shutil, subprocess
    from gencomm_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump) or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("llvm-objdump or the built library is not available")
    assert b"-packed-fp32-ops" in _lib.lib().gencomm_build_info()
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    objs = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert objs, os.listdir(tmp_path)
    asm = "\n".join(subprocess.run([objdump, "-d", str(tmp_path / o)], check=True, capture_output=True, text=True).stdout for o in sorted(objs))
    import re
    assert not re.search(r"v_pk_(fma|mul|add)_f32", asm)
    assert "v_mfma_f32_16x16x32_f16" in asm and "v_mfma_f32_4x4x1" in asm
    # the step noise is ONE field in every kernel: no kernel that runs Box-Muller (v_sin_f32) may round the product straight
    # to fp16 (v_fma_mix{lo,hi}_f16 = a single rounding; the other kernels round to fp32 first) -- csrc/common.h bm_pair
    parts = re.split(r"\n[0-9a-f]+ <([^>]+)>:\n", asm)
    noisy = [(n, b) for n, b in zip(parts[1::2], parts[2::2]) if "v_sin_f32" in b and ("latent_step" in n or "conv_out" in n or "step_noise" in n)]
    assert len(noisy) >= 6, [n for n, _ in noisy]
    for n, b in noisy:
        assert "v_fma_mixlo_f16" not in b and "v_fma_mixhi_f16" not in b, n


def _mfma_hazard_violations(asm):
    """gfx950 / ROCm 7.2: a matrix instruction that accumulates onto the result of a matrix instruction of ANOTHER input type
    (bf8 <-> f16) reads a stale half of the accumulator when it issues fewer than 6 wait states later; hipcc assumes SrcC
    forwarding and places such pairs back to back (tools/probes/mfma_mixed_dep_probe.hip). Returns the offending pairs: a
    dependent pair of different types with fewer than two other matrix instructions (>= 8 issue cycles) in between."""
    import re
    # Only the 16x16x32 forms (4 passes): the 32x32x16 forms (8 passes: enh_front_h_kernel, the general convolution) give exact results at
    # 0 wait states in both directions, alone and with 12 waves per SIMD (round 3: tools/probes/mfma32_mixed_dep_probe.hip,
    # profiles/r3_mfma32_probe.txt; re-confirmed in round 5: tools/probes/mfma_mixed_dep32_probe.hip, profiles/r5_mfma_mixed_dep32_probe.txt)
    pat = re.compile(r"^\s*(v_mfma_f32_16x16x32_(bf8_bf8|f16|bf16|fp8_fp8|bf8_fp8|fp8_bf8))\s+([av]\[\d+:\d+\]),\s*\S+,\s*\S+,\s*([av]\[\d+:\d+\]|0)")
    bad = []
    for name, body in zip(*(lambda p: (p[1::2], p[2::2]))(re.split(r"\n[0-9a-f]+ <([^>]+)>:\n", asm))):
        recent = []   # (type, dst) of the matrix instructions issued so far, newest last; cleared by anything that waits long
        for line in body.split("\n"):
            m = pat.match(line)
            if m is None:
                continue
            typ, dst, srcc = ("8" if "8" in m.group(2) else "16"), m.group(3), m.group(4)
            for age, (t0, d0) in enumerate(reversed(recent[-2:])):
                if d0 == srcc and t0 != typ:
                    bad.append((name, line.strip(), age))
            recent.append((typ, dst))
    return bad


def test_code_object_keeps_mixed_type_matrix_instructions_apart(tmp_path):
    import shutil, subprocess
    from gencomm_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump) or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("llvm-objdump or the built library is not available")
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    objs = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    asm = "\n".join(subprocess.run([objdump, "-d", str(tmp_path / o)], check=True, capture_output=True, text=True).stdout for o in sorted(objs))
    assert "v_mfma_f32_16x16x32_bf8_bf8" in asm          # the third-term instruction of the three-term products is there
    assert "v_mfma_f32_32x32x16_bf8_bf8" in asm          # ... and of the general convolution / wide weight gradient
    bad = _mfma_hazard_violations(asm)
    assert not bad, bad[:5]
    # the detector itself: the pair hipcc produced before the order was pinned
    sample = "\n0000 <k>:\n\tv_mfma_f32_16x16x32_bf8_bf8 v[18:21], v[22:23], v[18:19], v[2:5]\n\tv_mfma_f32_16x16x32_f16 v[18:21], v[36:39], v[58:61], v[18:21]\n"
    assert len(_mfma_hazard_violations(sample)) == 1
