"""V2XViTFusion. CPU: the oracle restatement (oracle/v2xvit_port.py) against the golden vector the
reference's own module produced (tests/golden/v2xvit.npz), and the product module's state_dict keys / shapes against the
reference's. GPU: the HIP path against the same golden vector and against the oracle on other shapes."""
import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))

from helpers import GOLDEN, assert_close, load_case, sub
from gencomm_amd import synth


def _case(g):
    args = json.loads(str(g["args"]))
    rl = [int(v) for v in g["record_len"]]
    C, H, W = int(g["C"]), int(g["H"]), int(g["W"])
    inp = synth.make_inputs(rl, C, H, W, int(g["data_seed"]), max_shift=float(g["max_shift"]))
    return args, rl, C, H, W, inp


def _module(args, seed):
    from gencomm_amd.v2xvit import V2XViTFusion
    net = V2XViTFusion(args).eval()
    synth.fill_params_(net, seed)
    return net


def test_state_dict_keys_match_reference():
    g = load_case("v2xvit")
    args = json.loads(str(g["args"]))
    with open(os.path.join(GOLDEN, "v2xvit_state_dict_keys.json")) as f:
        ref = json.load(f)
    net = _module(args, 0)
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == ref
    assert list(net.state_dict().keys()) == list(ref.keys())


def test_oracle_matches_reference_golden():
    import v2xvit_port as V
    from torch_port import normalize_pairwise_tfm
    g = load_case("v2xvit")
    args, rl, C, H, W, inp = _case(g)
    net = _module(args, int(g["weight_seed"]))
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    aff = normalize_pairwise_tfm(torch.from_numpy(inp["pairwise_t_matrix"]), H * 0.8, W * 0.8, 1.0)
    out = V.v2xvit_fusion(sd, args, torch.from_numpy(inp["feat"]), rl, aff)
    assert list(out.shape) == [int(v) for v in g["fused_shape"]]
    assert_close(sub(out, int(g["stride"])), g["fused"], 1e-5, 1e-6, "v2xvit oracle vs reference")


@pytest.mark.gpu
def test_hip_v2xvit_vs_reference_golden():
    from gencomm_amd import normalize_pairwise_tfm
    g = load_case("v2xvit")
    args, rl, C, H, W, inp = _case(g)
    net = _module(args, int(g["weight_seed"])).cuda()
    aff = normalize_pairwise_tfm(torch.from_numpy(inp["pairwise_t_matrix"]), H * 0.8, W * 0.8, 1)
    with torch.no_grad():
        out = net(torch.from_numpy(inp["feat"]).cuda(), torch.tensor(rl), aff)
    assert list(out.shape) == [int(v) for v in g["fused_shape"]]
    assert_close(sub(out, int(g["stride"])), g["fused"], 1e-4, 1e-5, "v2xvit HIP vs reference golden")


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,rl,hetero", [(32, 48, [5], True), (16, 16, [1, 2], False), (64, 128, [2], True)])
def test_hip_v2xvit_vs_oracle(H, W, rl, hetero):
    import v2xvit_port as V
    from torch_port import normalize_pairwise_tfm as npt_oracle
    from gencomm_amd import normalize_pairwise_tfm
    g = load_case("v2xvit")
    args = json.loads(str(g["args"]))
    args["transformer"]["encoder"]["cav_att_config"]["use_hetero"] = hetero
    args["transformer"]["encoder"]["depth"] = 2
    C = 128
    net = _module(args, 77)
    inp = synth.make_inputs(rl, C, H, W, 78, max_shift=10.0)
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    ptm = torch.from_numpy(inp["pairwise_t_matrix"])
    if hetero:
        ref = V.v2xvit_fusion(sd, args, torch.from_numpy(inp["feat"]), rl, npt_oracle(ptm, H * 0.8, W * 0.8, 1.0))
    net = net.cuda()
    with torch.no_grad():
        out = net(torch.from_numpy(inp["feat"]).cuda(), rl, normalize_pairwise_tfm(ptm, H * 0.8, W * 0.8, 1)).cpu()
    assert torch.isfinite(out).all() and list(out.shape) == [len(rl), C, H, W]
    if hetero:
        assert_close(out.numpy(), ref.numpy(), 1e-4, 1e-5, "v2xvit HIP vs oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,rl,hetero", [(16, 16, [2, 1], True), (16, 32, [3], True)])
def test_hip_v2xvit_backward_vs_oracle_autograd(H, W, rl, hetero):
    """Gradients of every parameter the forward uses and of the input against float64 autograd through the oracle (eval mode:
    no dropout). Parameters of agent types / relations GenComm never selects get no gradient on either side."""
    import v2xvit_port as V
    from torch_port import normalize_pairwise_tfm as npt_oracle
    from gencomm_amd import normalize_pairwise_tfm
    g = load_case("v2xvit")
    args = json.loads(str(g["args"]))
    args["transformer"]["encoder"]["cav_att_config"]["use_hetero"] = hetero
    args["transformer"]["encoder"]["depth"] = 2
    C = 128
    net = _module(args, 31)
    inp = synth.make_inputs(rl, C, H, W, 32, max_shift=3.0)
    ptm = torch.from_numpy(inp["pairwise_t_matrix"])
    sd = {k: v.detach().double().requires_grad_(v.is_floating_point()) for k, v in net.state_dict().items()}
    xd = torch.from_numpy(inp["feat"]).double().requires_grad_(True)
    ref = V.v2xvit_fusion(sd, args, xd, rl, npt_oracle(ptm, H * 0.8, W * 0.8, 1.0))
    names = [k for k in sd if sd[k].requires_grad]
    rg = torch.autograd.grad((ref ** 2).mean(), [sd[k] for k in names] + [xd], allow_unused=True)
    net = net.cuda()
    x = torch.from_numpy(inp["feat"]).cuda().requires_grad_(True)
    out = net(x, rl, normalize_pairwise_tfm(ptm, H * 0.8, W * 0.8, 1))
    assert_close(out.detach().cpu().numpy(), ref.detach().numpy(), 1e-4, 1e-5, "v2xvit forward (grad mode)")
    (out ** 2).mean().backward()

    def close(name, got, want):
        scale = float(want.abs().max())
        err = float((got.detach().cpu().double() - want).abs().max())
        assert err <= 2e-3 * scale + 1e-9, (name, err, scale)
        return err / (scale + 1e-30)
    worst = close("grad input", x.grad, rg[-1])
    got = dict(net.named_parameters())
    used = 0
    for k, r in zip(names, rg[:-1]):
        if r is None or float(r.abs().max()) == 0.0:
            assert got[k].grad is None or float(got[k].grad.abs().max()) == 0.0, k
            continue
        assert got[k].grad is not None, k
        rel = close("grad " + k, got[k].grad, r)
        if float(r.abs().max()) > 1e-9:      # gradients that are zero analytically (e.g. the key bias under softmax) are noise on both sides
            worst = max(worst, rel)
        used += 1
    print(f"V2X-ViT backward {H}x{W} scenes {rl} hetero={hetero}: {used} parameter gradients + input, worst relative error {worst:.2e}")


@pytest.mark.gpu
def test_hip_v2xvit_train_mode_dropout_is_consistent():
    """Train mode: the reference's dropouts (p = 0.3 after the attention projections and inside the FeedForward) are active -- also
    for a frozen fusion net in stage 2. Same torch seed -> same masks -> identical output; the input gradient agrees with a central
    finite difference of the loss along a random direction (masks replayed through the seed)."""
    from gencomm_amd import normalize_pairwise_tfm
    g = load_case("v2xvit")
    args = json.loads(str(g["args"]))
    args["transformer"]["encoder"]["depth"] = 1
    H, W, rl, C = 16, 16, [2], 128
    net = _module(args, 5).cuda().train()
    inp = synth.make_inputs(rl, C, H, W, 6, max_shift=2.0)
    aff = normalize_pairwise_tfm(torch.from_numpy(inp["pairwise_t_matrix"]), H * 0.8, W * 0.8, 1)
    x = torch.from_numpy(inp["feat"]).cuda()

    def loss_of(xx):
        torch.manual_seed(1234)
        return (net(xx, rl, aff) ** 2).mean()
    xg = x.clone().requires_grad_(True)
    l0 = loss_of(xg)
    l0.backward()
    with torch.no_grad():
        assert float((loss_of(xg.detach().requires_grad_(True)) - l0).abs()) == 0.0          # same seed, same masks
        net.eval()
        l_eval = (net(x, rl, aff) ** 2).mean()
        net.train()
    assert abs(float(l_eval) - float(l0)) > 1e-4 * abs(float(l0))                               # dropout does something
    d = xg.grad / xg.grad.norm() * x.norm() * 0.05          # along the gradient: the largest, best-conditioned directional derivative
    fd = (float(loss_of((x + d).requires_grad_(True))) - float(loss_of((x - d).requires_grad_(True)))) / 2
    an = float((xg.grad * d).sum())
    print(f"V2X-ViT train-mode dropout: directional derivative analytic {an:.6e}, finite difference {fd:.6e}")
    assert abs(an - fd) <= 3e-2 * abs(fd) + 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("DH,heads,H,W,lens", [(32, 8, 136, 128, [3, 1]), (16, 8, 96, 200, [2, 4]), (32, 4, 16, 24, [5])])
def test_agent_attention_kernels_vs_torch(DH, heads, H, W, lens):
    """gencomm_hgt_attn_fwd directly: the streaming kernel of the large maps (every q / k / v value read once, all N x N scores in
    registers; taken from 131 072 (pixel, head) threads up) and the per-query-agent kernel of the small ones against plain torch."""
    from gencomm_amd import _lib
    from gencomm_amd.runtime import ptr, stream_ptr
    torch.manual_seed(DH + H)
    n, inner, HW = sum(lens), heads * DH, H * W
    qkv = torch.randn(n, 3 * inner, H, W)
    off = [0]
    for v in lens:
        off.append(off[-1] + v)
    q, k, v = (qkv[:, i * inner:(i + 1) * inner].reshape(n, heads, DH, HW) for i in range(3))
    ref = torch.empty(n, heads, DH, HW)
    for b in range(len(lens)):
        sl = slice(off[b], off[b + 1])
        s = torch.einsum("ihdp,jhdp->hpij", q[sl], k[sl]) / DH ** 0.5
        ref[sl] = torch.einsum("hpij,jhdp->ihdp", s.softmax(-1), v[sl])
    dev = torch.device("cuda:0")
    qd, out = qkv.to(dev).contiguous(), torch.empty(n, inner, H, W, device=dev)
    so = torch.tensor(off, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().gencomm_hgt_attn_fwd(ptr(qd), ptr(so), ptr(out), len(lens), heads, DH, HW, stream_ptr(dev)), "gencomm_hgt_attn_fwd")
    assert_close(out.cpu().numpy(), ref.reshape(n, inner, H, W).numpy(), 1e-4, 1e-5, f"agent attention DH={DH} {H}x{W} {lens}")
