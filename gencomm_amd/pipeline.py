"""``ScenePipeline`` -- the whole hot path for a fixed batch geometry with every buffer preallocated:

    GenComm (q_sample + T denoise steps)  ->  Enhancer (optional)  ->  warp + AttFusion

i.e. lines :258, :279-282 of the reference shell
(``opencood/models/heter_model_baseline_w_gencomm_stage1.py``) as three C-ABI calls on the current
stream with no per-step host allocation, H2D copy or synchronisation. Used by ``bench.py`` and by
callers that run many scenes of one shape (inference servers); the module classes
(``GenComm``/``Enhancer``/``AttFusion``) are the drop-in API, this is the fast lane under them.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from .cond_diff import GenComm
from .enhancer import Enhancer
from .fusion import MAX_AGENTS_PER_SCENE, gather_ego_thetas
from .runtime import f32c, ptr, stream_ptr


class ScenePipeline:
    def __init__(self, gencomm: GenComm, enhancer: Optional[Enhancer], record_len: Sequence[int],
                 C: int, H: int, W: int, device: torch.device, token_fast_path: Optional[bool] = None,
                 graph: bool = False):
        """``graph=True``: the launch sequence of one ``run`` (hundreds of short kernels) is captured once into a HIP
        graph and replayed; inputs are copied into pipeline-owned buffers and the Philox key lives in device memory
        (``gencomm_denoise_fwd_dseed``), so every replay draws fresh noise. Off by default: measured on MI355X / ROCm 7.2
        a replay is no faster than the eager launches (shipped shape, one scene in flight: 1.086 vs 1.095 ms; the
        ~100 dependent kernels cost ~10 us each on the GPU side either way) and replays of different streams
        overlap worse than eager launches do (metric workload 145 vs 161 scenes/s) -- DESIGN.md section 5."""
        self.gen, self.enh = gencomm, enhancer
        self.use_graph = bool(graph)
        self._graph = None
        # Enhancer -> fusion without the NCHW round trip (self.enhanced is then NOT produced)
        self.token_fast_path = (C in (64, 128, 256)) if token_fast_path is None else bool(token_fast_path)
        self.lens = [int(v) for v in record_len]
        if min(self.lens) < 1 or max(self.lens) > MAX_AGENTS_PER_SCENE:
            raise ValueError(f"each scene needs 1..{MAX_AGENTS_PER_SCENE} agents, got {self.lens}")
        self.B, self.n = len(self.lens), sum(self.lens)
        self.C, self.H, self.W = C, H, W
        self.device = torch.device(device)
        self.T = gencomm.num_timesteps
        den = gencomm.denoiser
        self.L, self.R, self.A = den.num_resolutions, den.num_res_blocks, den.attn_mask
        l = _lib.lib()
        dev = self.device
        rows, off, o = [], [0], 0
        for k in self.lens:
            rows += [o] * k
            o += k
            off.append(o)
        self.src_rows = torch.tensor(rows, dtype=torch.int32, device=dev)
        self.scene_off = torch.tensor(off, dtype=torch.int32, device=dev)
        self.theta = torch.zeros(self.n, 2, 3, dtype=torch.float64, device=dev)
        self.pred = torch.empty(self.n, C, H, W, dtype=torch.float32, device=dev)
        self.enhanced = torch.empty_like(self.pred) if enhancer is not None else self.pred
        self.fused = torch.empty(self.B, C, H, W, dtype=torch.float32, device=dev)
        ws_d = _lib.check_size(l.gencomm_denoise_workspace_bytes(self.n, C, H, W, self.L, self.R, self.A), "gencomm_denoise_workspace_bytes")
        ws_e = _lib.check_size(l.gencomm_enhancer_workspace_bytes(self.n, C, H, W), "gencomm_enhancer_workspace_bytes") if enhancer is not None else 0
        # one arena: the two stages never overlap in time on a stream
        self.ws = torch.empty(max(ws_d, ws_e, 256), dtype=torch.uint8, device=dev)
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)  # Philox key for graph replays
        if self.use_graph:
            self.in_feat = torch.empty(self.n, C, H, W, dtype=torch.float32, device=dev)
            self.in_cond = torch.empty(self.n, 2, H, W, dtype=torch.float32, device=dev)
        self.refresh_params()

    def refresh_params(self) -> None:
        """Re-pack parameters after a weight update (cheap no-op when nothing changed)."""
        prepared = self.gen.denoiser.prepared_params(self.T, self.device)
        enh_raw = self.enh._raw_params(self.device) if self.enh is not None else None
        if self._graph is not None and (prepared.data_ptr() != self.prepared.data_ptr() or
                                        (enh_raw is not None and enh_raw.data_ptr() != self.enh_raw.data_ptr())):
            self._graph = None  # parameter blobs moved: the captured pointers are stale
        self.prepared = prepared
        self.sched = self.gen._sched_table(self.device)
        self.enh_raw = enh_raw

    def set_affine(self, affine_matrix: torch.Tensor) -> None:
        """affine_matrix [B,L,L,2,3] (output of normalize_pairwise_tfm)."""
        self.theta.copy_(gather_ego_thetas(affine_matrix, self.lens))

    def run(self, feat: torch.Tensor, cond: torch.Tensor, seed: int = 0,
            noise: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
        """feat [sumN,C,H,W] f32, cond [sumN,2,H,W] f32 (device, contiguous) -> fused [B,C,H,W].
        ``self.pred`` / ``self.enhanced`` hold the intermediate stage outputs afterwards."""
        n, C, H, W = self.n, self.C, self.H, self.W
        assert feat.is_cuda and cond.is_cuda and feat.dtype == torch.float32 and cond.dtype == torch.float32
        assert feat.is_contiguous() and cond.is_contiguous()
        assert tuple(feat.shape) == (n, C, H, W) and tuple(cond.shape) == (n, 2, H, W)
        if not self.use_graph or noise is not None:
            return self._enqueue(feat, cond, seed, noise, None)
        # graph path: inputs into the captured buffers, key into device memory, replay
        if feat.data_ptr() != self.in_feat.data_ptr():
            self.in_feat.copy_(feat, non_blocking=True)
        if cond.data_ptr() != self.in_cond.data_ptr():
            self.in_cond.copy_(cond, non_blocking=True)
        self.seed_dev.fill_(int(seed) & 0x7FFFFFFFFFFFFFFF)
        if self._graph is None:
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._enqueue(self.in_feat, self.in_cond, 0, None, self.seed_dev)  # warm-up outside the capture
            cur.wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._enqueue(self.in_feat, self.in_cond, 0, None, self.seed_dev)
            self._graph = g
        self._graph.replay()
        return self.fused

    def _enqueue(self, feat, cond, seed, noise, seed_dev) -> torch.Tensor:
        l = _lib.lib()
        st = stream_ptr(self.device)
        n, C, H, W = self.n, self.C, self.H, self.W
        n0 = sn = None
        if noise is not None:
            n0, sn = f32c(noise[0]), f32c(noise[1])
        _lib.check(l.gencomm_denoise_fwd_dseed(ptr(self.prepared), ptr(self.sched), ptr(feat), n, ptr(self.src_rows), ptr(cond),
                                               ptr(self.pred), ptr(n0), ptr(sn), seed, ptr(seed_dev), n, C, H, W, self.L, self.R, self.A,
                                               self.T, ptr(self.ws), self.ws.numel(), st), "gencomm_denoise_fwd_dseed")
        if self.enh is not None and self.token_fast_path:
            # Enhancer result stays token-major in the workspace; the fusion kernel applies the channel gate
            _lib.check(l.gencomm_enhancer_fwd(ptr(self.enh_raw), ptr(self.pred), None, n, C, H, W,
                                              ptr(self.ws), self.ws.numel(), st), "gencomm_enhancer_fwd")
            _lib.check(l.gencomm_warp_attfuse_tok_fwd(ptr(self.ws), ptr(self.theta), ptr(self.scene_off), ptr(self.fused),
                                                      self.B, n, C, H, W, st), "gencomm_warp_attfuse_tok_fwd")
            return self.fused
        if self.enh is not None:
            _lib.check(l.gencomm_enhancer_fwd(ptr(self.enh_raw), ptr(self.pred), ptr(self.enhanced), n, C, H, W,
                                              ptr(self.ws), self.ws.numel(), st), "gencomm_enhancer_fwd")
        _lib.check(l.gencomm_warp_attfuse_fwd(ptr(self.enhanced), ptr(self.theta), ptr(self.scene_off), ptr(self.fused),
                                              self.B, n, C, H, W, st), "gencomm_warp_attfuse_fwd")
        return self.fused
