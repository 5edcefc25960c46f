"""The Enhancer's three launch structures at C = 64 (GENCOMM_MODE_ENH_FUSE 0 / 1 / 2: separate launches, Linear1 + depthwise stage
fused, + Linear2 + residual + pool sums fused) against the CPU oracle (enhancer.py:222-250, :315-333, :346-383), on shapes with
partial 8x8 tiles, and through ScenePipeline's token-major hand-over to the fusion kernel (which has to find the result where the
selected structure wrote it)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("fuse", [0, 1, 2])
@pytest.mark.parametrize("shape", [(3, 21, 37), (2, 64, 128), (1, 8, 8)])
def test_enhancer_structures_vs_oracle(modes, fuse, shape):
    from gencomm_amd import Enhancer, synth
    from oracle import torch_port as O
    n, H, W = shape
    C = 64
    modes(enh_fuse=fuse)
    enh = Enhancer(C, [8, 8], 4).eval()
    synth.fill_params_(enh, 41)
    g = torch.Generator().manual_seed(42 + H)
    x = torch.randn(n, C, H, W, generator=g) * 0.7
    with torch.no_grad():
        sd = {k: v.detach() for k, v in enh.state_dict().items()}
        ref = O.enhancer_forward(sd, x, [n])
        got = enh.to(DEV)(x.to(DEV), None, [n]).cpu()
    err = (got - ref).abs()
    tol = 1e-5 + 1e-4 * ref.abs()
    print(f"enh_fuse={fuse} {shape}: max abs err {err.max().item():.3e}, worst err/tol {(err / tol).max().item():.3f}")
    assert (err <= tol).all()


@pytest.mark.parametrize("fuse", [0, 1, 2])
def test_pipeline_token_path_finds_the_enhancer_result(modes, fuse):
    from gencomm_amd import Enhancer, GenComm, normalize_pairwise_tfm, synth
    from gencomm_amd.pipeline import ScenePipeline
    C, H, W, T, rl = 64, 22, 46, 3, [3, 1, 2]
    gen, enh = GenComm(synth.default_gencomm_cfg(C, T)).eval().to(DEV), Enhancer(C, [8, 8], 4).eval().to(DEV)
    synth.fill_params_(gen, 5)
    synth.fill_params_(enh, 6)
    inp = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_inputs(rl, C, H, W, 7, max_shift=12.0).items()}
    outs = []
    with torch.no_grad():
        affine = normalize_pairwise_tfm(inp["pairwise_t_matrix"], H * 0.8, W * 0.8, 1)
        for f, fast in ((fuse, True), (0, False)):
            modes(enh_fuse=f)
            pipe = ScenePipeline(gen, enh, rl, C, H, W, torch.device(DEV), token_fast_path=fast)
            pipe.set_affine(affine)
            outs.append(pipe.run(inp["feat"].contiguous(), inp["cond"].contiguous(), seed=3).clone())
    err = (outs[0] - outs[1]).abs()
    assert (err <= 2e-5 + 1e-4 * outs[1].abs()).all(), err.max().item()


@pytest.mark.parametrize("big", [20.0, 300.0])
def test_fused_enhancer_with_large_linear_weights_stays_finite_and_close(modes, big):
    """the fused structure's operand tables used a static weight scale of 2^12, so a Linear weight of magnitude >= 16 became
    inf in fp16 and NaN reached the output silently.  The scale is now chosen per tensor on the device (enh_wscale_kernel): a few entries of
    Linear1 / Linear2 are set to +-`big`, the result must match the oracle at the elementwise tolerance relative to its magnitude."""
    from gencomm_amd import Enhancer, synth
    from oracle import torch_port as O
    n, C, H, W = 2, 64, 16, 24
    modes(enh_fuse=2)
    enh = Enhancer(C, [8, 8], 4).eval()
    synth.fill_params_(enh, 43)
    with torch.no_grad():
        w1, w2 = enh.block_1.mlp.linear1[0].weight, enh.block_1.mlp.linear2[0].weight
        w1[3, 5], w1[200, 60], w2[7, 11], w2[40, 100] = big, -big, big, -big
    g = torch.Generator().manual_seed(44)
    x = torch.randn(n, C, H, W, generator=g) * 0.5
    with torch.no_grad():
        ref = O.enhancer_forward({k: v.detach() for k, v in enh.state_dict().items()}, x, [n])
        got = enh.to(DEV)(x.to(DEV), None, [n]).cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    tol = 1e-5 * ref.abs().max() + 1e-4 * ref.abs()
    print(f"large weights {big}: max |ref| {ref.abs().max().item():.3e}, worst err/tol {(err / tol).max().item():.3f}")
    assert (err <= tol).all()
