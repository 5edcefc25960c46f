"""GENCOMM_MODE_DATAFLOW: the body of a UNet call as ONE persistent launch with per-(layer, agent) completion counters
(csrc/dataflow_kernels.h) against the per-layer launches (same device functions) and against the reference's golden vectors.
The two modes differ only in the order of the f64 statistics atomics, so they agree to ~1e-6; the dataflow error word must
stay 0 (no dependency wait timed out)."""
import pytest
import torch

from helpers import assert_close, build_inputs, build_modules, eval_noise, load_case, sub

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL, ATOL = 1e-4, 1e-5


def _df_error(gen, n, H, W):
    from gencomm_amd import _lib
    from gencomm_amd.runtime import ptr, stream_ptr
    den = gen.denoiser
    ws = den.denoise_workspace(n, H, W, torch.device(DEV))
    return _lib.lib().gencomm_dataflow_error(ptr(ws), n, den.feature_channels, H, W, den.num_resolutions, den.num_res_blocks, den.attn_mask,
                                             stream_ptr(torch.device(DEV)))


@pytest.mark.parametrize("sampler", ["latent", "direct"])
@pytest.mark.parametrize("name", ["tiny", "ragged", "mid", "shipped"])
def test_dataflow_golden_path(name, sampler, modes):
    from gencomm_amd import _lib
    g = load_case(name)
    _, gen, _ = build_modules(g, DEV)
    inp = build_inputs(g, DEV)
    noise = eval_noise(g, DEV)
    st = int(g["stride"])
    n, H, W = inp["feat"].shape[0], int(g["H"]), int(g["W"])
    outs = {}
    for df in (0, 1):
        modes(tile_want=1, sampler=sampler, dataflow=df)   # 64x16 f16-pipe kernels on every level: the dataflow kernel's layer set
        with torch.no_grad(), _lib.kernel_log() as kl:
            outs[df] = gen(inp["feat"], inp["cond"], inp["record_len"], noise=noise)["pred_feature"].clone()
            torch.cuda.synchronize()
        ran_df = any("unet_dataflow_kernel" in k for k in kl.counts)
        assert ran_df == (bool(df) and W % 4 == 0), kl.counts   # a map width that is no multiple of 4 keeps the per-layer launches
        if ran_df:
            assert not any(k.startswith("conv8h_kernel") or k.startswith("down8") for k in kl.counts), kl.counts
            assert _df_error(gen, n, H, W) == 0
    assert_close(sub(outs[1], st), g["pred_feature"], RTOL, ATOL, f"{name} {sampler}: dataflow vs golden")
    err = (outs[1] - outs[0]).abs()
    assert float((err / (1e-6 + 1e-5 * outs[0].abs())).max()) <= 1.0, float(err.max())


def test_dataflow_many_agents_and_repeatability(modes):
    """32 agents x 2 levels on a 96 x 192 map (more items than resident workgroups at full resolution, fewer at half
    resolution), several calls back to back on one workspace: counters are re-zeroed per call, results repeat."""
    from gencomm_amd import GenComm, synth
    C, H, W, T, n = 32, 96, 192, 4, 32
    gen = GenComm(synth.default_gencomm_cfg(C, T)).eval()
    synth.fill_params_(gen, 3)
    gen = gen.to(DEV)
    inp = synth.make_inputs([4] * (n // 4), C, H, W, 5)
    feat, cond = torch.from_numpy(inp["feat"]).to(DEV), torch.from_numpy(inp["cond"]).to(DEV)
    modes(dataflow=0, tile8=0)   # the dataflow kernel has the 64 x 16 tile functions only: the per-layer reference must not pick 64 x 8 tiles at half resolution
    with torch.no_grad():
        ref = gen(feat, cond, [n], seed=9)["pred_feature"].clone()
    modes(dataflow=1)
    for rep in range(3):
        with torch.no_grad():
            got = gen(feat, cond, [n], seed=9)["pred_feature"]
            torch.cuda.synchronize()
        assert _df_error(gen, n, H, W) == 0
        err = (got - ref).abs()
        assert float((err / (1e-6 + 1e-5 * ref.abs())).max()) <= 1.0, (rep, float(err.max()))
