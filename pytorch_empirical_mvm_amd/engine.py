"""The VIOLETv2 pretraining step as a static schedule of libvmvm kernel launches (forward, losses, backward).

No autograd graph: every block has an explicit forward that pushes its backward closure on a tape; weight
gradients are accumulated by the wgrad GEMM epilogue straight into one flat f32 gradient arena (which the
data-parallel all-reduce and the fused AdamW walk as a whole), activations are bf16, statistics f32.
torch supplies device memory, the current stream and index uploads only.

Reference call stack reproduced: VIOLET_Pretrain.forward (main_pretrain.py:226-267) ->
EncVideo.forward (model.py:32-78) -> SwinTransformer3D.forward (video_swin.py:470-482) ; EncTxt (model.py:106-115) ;
go_cross x2 (model.py:204-214) ; heads + losses (main_pretrain.py:374-432, 555-567)."""
import math
import os
from collections import OrderedDict, deque

import numpy as np
import torch

from . import config as CFG
from . import kernels as K
from . import lib as L
from . import swin_index as SI
from .engine_downstream import DownstreamMixin
from .engine_fusion import FusionMixin
from .engine_heads import HeadsMixin
from .engine_swin import SwinMixin
from .switches import Switches
from .store import BF16, F32, DropScale, ParamStore, V, _acc, _dev_i32, _gout, _h2d      # noqa: F401  (re-exported: tests and tools import them from here)


class VioletEngine(SwinMixin, FusionMixin, HeadsMixin, DownstreamMixin):
    """The step schedule.  This file: construction, dropout-offset / index caches, the weight-gradient stream and `_linear_bwd`; the layers
    live in engine_swin.py / engine_fusion.py, the heads and `forward_backward` in engine_heads.py, the downstream passes in
    engine_downstream.py, the parameter arena in store.py."""
    def __init__(self, cfg, device="cuda", seed=88):
        self.cfg = cfg
        self.device = torch.device(device)
        self.store = ParamStore(CFG.param_shapes(cfg), self.device)
        self.seed = int(seed)
        self.rng_offset = 0
        self._drop_sites = None                # None = every dropout site follows `train` (forward_backward(dropout=...))
        self.last_offsets = {}                 # Philox offsets of the last pass' named dropout sites (tests recover the masks)
        self._idx_cache = {}
        self.tape = []
        self.teacher = None                 # frozen dVAE tokenizer (MVM 'vq' target), set by the model
        self.feature_teacher = None         # frozen Swin teacher (MVM '3d_feature' / '2d_feature' targets), set by the model
        self.on_swin_tail_ready = None      # data-parallel hook (dist.GradReducer.reduce_swin_tail)
        self.on_fusion_mid_ready = None     # data-parallel hook (dist.GradReducer.reduce_other_early): the backward has left fusion layer n // 2
        self.dpr = np.linspace(0, CFG.DROP_PATH_RATE, sum(cfg["depths"])).tolist()     # video_swin.py:447
        # BASELINE config 5 ("fp8 MFMA path"), OPT-IN (`fp8_forward`, bench.py --fp8): the forward GEMMs of the fusion encoder's qkv and
        # FFN-in projections on e4m3 operands (per-tensor static power-of-two scales, v_mfma_scale_f32_16x16x128_f8f6f4 / 32x32x64);
        # the backward stays bf16 on the bf16 activations.  Rounds 1-4 carried it, round 5 removed it because it never moved that
        # config's number (58 % of the step is streaming attention), round 6 restores it as what the config names -- measured beside
        # the bf16 line, never the headline.
        self.fp8 = bool(cfg.get("fp8_forward", False)) and self.device.type == "cuda"
        self.A8_SCALE = 16.0               # activations entering those GEMMs are LayerNorm outputs / the gathered pool: |x| < 28 before e4m3 saturates
        if self.fp8:
            self.store.enable_fp8()
        self.sw = Switches.from_env()       # every VMVM_* switch of the step path, read once (switches.py)
        self.store_drop_mask = self.sw.drop_mask      # fusion attention: the backward reads the forward's dropout decisions (44.8 MB per layer at C2) instead of re-evaluating Philox twice
        self.gelu_code8 = bool(cfg.get("gelu_code8", self.sw.gelu_code8))         # Swin MLPs keep GELU' as an 8-bit code
        # Weight-gradient stream: dW = dY^T X (+ the bias column sums) depend on nothing downstream in the backward, so they
        # are enqueued on a second HIP stream and run NEXT TO the main stream's chain (input-gradient GEMMs, attention / LayerNorm
        # backward): the tail rounds of the persistent GEMM grids and the memory- / latency-bound kernels between them get filled
        # with MFMA work.  Joined before the gradient exchange / the optimizer.  VMVM_WGRAD_STREAM=0: everything on one stream.
        self.wstream = None
        self.other_ready = None             # event: the non-Swin half of the previous optimizer step (agent.backward_step) has finished on wstream
        self._wpending = False              # weight-gradient launches on wstream the main stream has not waited for yet
        self._wheld = deque()               # (event on wstream behind a weight-gradient launch, its operands): kept alive until the side stream has passed it
        self._wev_pool = []
        if self.device.type == "cuda":
            self.workspace = torch.empty(192 << 20, device=self.device, dtype=torch.uint8)        # split-K slabs of the wgrad GEMMs
            K.set_workspace(self.workspace)
            if bool(cfg.get("wgrad_stream", self.sw.wgrad_stream)):
                self.wstream = torch.cuda.Stream(device=self.device)
                self.workspace_w = torch.empty(192 << 20, device=self.device, dtype=torch.uint8)  # the side stream's own split-K slabs

    # -------------------------------------------------------------- small helpers
    def _next_offset(self, n):
        o = self.rng_offset
        self.rng_offset += int(n) + 64
        return o

    def _cached(self, key, fn):
        if key not in self._idx_cache:
            self._idx_cache[key] = fn()
        return self._idx_cache[key]

    def _wgrad_launch(self, fn, operands, sync=False):
        """run fn(workspace) -- weight-gradient launches -- on the side stream behind everything enqueued so far.  The operands (dY, the
        saved forward activation, row scales) are marked as in use by that stream (`record_stream`): the caching allocator then holds
        each block back only until the side stream has passed the launch, so the backward frees memory as the tape unwinds.  (Round 3
        kept a Python reference to every operand until the join at the end of the backward: ~2 GB per fusion layer at M = 69 120 pinned
        for the whole backward.)"""
        if self.wstream is None or sync:
            fn(None)
            return
        self.wstream.wait_stream(torch.cuda.current_stream())
        with L.on_stream(self.wstream):
            fn(self.workspace_w)
            if self.sw.wgrad_hold:
                ev = self._wev_pool.pop() if self._wev_pool else torch.cuda.Event()
                ev.record()
        if self.sw.wgrad_hold:
            # Round 6: the operands are HELD (a Python reference in a FIFO) until the side stream has passed their launch -- one event per
            # launch, the oldest polled once per launch.  Round 4-5 marked them with `record_stream` instead: every block freed that way
            # leaves an event in the caching allocator, which polls ALL outstanding ones on EVERY allocation -- torch.empty cost 48 us
            # per call in the step (8 205 calls, 0.40 s of a 9-step profile: 44 ms of the host's ~65 ms per step, tools/scratch/host_profile.py).
            held = self._wheld
            held.append((ev, operands))
            while held and held[0][0].query():
                self._wev_pool.append(held.popleft()[0])
        else:
            for t in operands:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(self.wstream)
        self._wpending = True

    def _whold(self, operands):
        """weight-gradient launches were just enqueued on the side stream by a block-level call (vmvm_bert_layer_bwd): keep their operands
        alive until the side stream has passed them (the FIFO of `_wgrad_launch`)"""
        if self.wstream is None:
            return
        ev = self._wev_pool.pop() if self._wev_pool else torch.cuda.Event()
        ev.record(self.wstream)
        held = self._wheld
        held.append((ev, operands))
        while held and held[0][0].query():
            self._wev_pool.append(held.popleft()[0])
        self._wpending = True

    def _fork_event(self):
        """raw handle of the event vmvm_bert_layer_bwd records on the main stream in front of each side-stream launch"""
        ev = getattr(self, "_fork_ev", None)
        if ev is None:
            ev = self._fork_ev = torch.cuda.Event()
            ev.record()                      # (the handle exists from the first record on)
        return ev.cuda_event

    def _wgrad_join(self):
        """the main stream waits for the weight gradients in flight (before the gradient exchange / norm / AdamW read them)"""
        if self.wstream is not None and self._wpending:
            torch.cuda.current_stream().wait_stream(self.wstream)
            self._wpending = False
            # everything the main stream enqueues from here on is ordered behind the side stream: the held operands can go
            while self._wheld:
                self._wev_pool.append(self._wheld.popleft()[0])

    def _linear_bwd(self, dy, x, wname, bname, *, w=None, gw=None, gb=None, M=None, row_scale=None, rows_per_scale=0,
                    need_dx=True, dx_kw=None, wN=None, wT=None, wsync=False, cs_scale=None):
        """dW += dy^T x ; db += colsum(dy) ; dx = dy W   (all on the MFMA GEMM, no transposed copies)."""
        S = self.store
        w = S.b(wname) if w is None else w
        gw = S.g(wname) if gw is None else gw
        w2 = w.view(w.shape[0], -1)
        gw2 = gw.view(gw.shape[0], -1)
        N = wN or w2.shape[0]
        gbias = None
        if bname is not None or gb is not None:
            gbias = S.g(bname) if gb is None else gb

        def wgrad(ws):
            gb_ = gbias
            if gb_ is not None and row_scale is not None and cs_scale is None:   # per-clip DropPath weights: separate pass (the fused form has ONE scale)
                K.colsum(dy, gb_, row_scale, rows_per_scale, accumulate=True, M=M, N=N, workspace=ws)
                gb_ = None
            # db = [cs_scale *] colsum(dy) rides on the weight-gradient GEMM (dy is its A operand)
            K.gemm(dy, x, a_kmajor=False, b_kmajor=False, M=N, N=w2.shape[1], K=M or dy.shape[0], out=gw2, accumulate=True, colsum=gb_,
                   colsum_scale=cs_scale or 0.0, workspace=ws)
        self._wgrad_launch(wgrad, (dy, x, row_scale), sync=wsync)
        if not need_dx:
            return None
        wt = S.bt(wname) if (wname is not None and wT is None) else wT
        if wt is not None and wt.shape[1] == N:
            return K.gemm(dy, wt, b_kmajor=True, M=M or dy.shape[0], N=wt.shape[0], K=N, **(dx_kw or {}))
        return K.gemm(dy, w2, b_kmajor=False, M=M or dy.shape[0], N=w2.shape[1], K=N, **(dx_kw or {}))

    def _inverse_rows(self, n_rows, row_lists):
        inv = np.full(n_rows, -1, dtype=np.int32)
        off = 0
        for r in row_lists:
            rr = r.cpu().numpy()
            inv[rr] = off + np.arange(rr.size)
            off += rr.size
        return _dev_i32(inv, self.device)

    # -------------------------------------------------------------- stochastic pieces (host RNG, tiny uploads)
    def sample_negatives(self, B, rng=None):
        """main_pretrain.py:250-256 : O-1 negatives per clip from the LOCAL batch, without replacement."""
        rng = rng or np.random
        O = min(B, 4)
        return np.stack([rng.permutation([j for j in range(B) if j != i])[:O - 1] for i in range(B)]) if O > 1 else np.zeros((B, 0), np.int64)

    def sample_drop_path(self, B, rng=None):
        """video_swin.py:46-54 : per-sample keep mask floor(keep + U), scaled by 1/keep.  A block draws twice, in the order the reference's
        forward calls drop_path (:256 attention branch, then :248 MLP branch): one ((B,), (B,)) pair of f32 vectors per block."""
        rng = rng or np.random
        rows = []
        for p in self.dpr:
            keep = 1.0 - p
            for _ in range(2):
                rows.append(np.floor(keep + rng.rand(B)) / keep if p > 0 else np.ones(B))
        host = np.stack(rows).astype(np.float32)                       # [2 * blocks, B]
        # one upload: the scales, and per draw the kept clips' scales (then zeros) and the kept / dropped clip lists (-1 behind their entries)
        nr = host.shape[0]
        pack_f = np.zeros((2, nr, B), np.float32)
        pack_i = np.full((2, nr, B), -1, np.int32)
        pack_f[0] = host
        for i in range(nr):
            k = np.flatnonzero(host[i] != 0)
            d = np.flatnonzero(host[i] == 0)
            pack_i[0, i, :k.size] = k
            pack_i[1, i, :d.size] = d
            pack_f[1, i, :k.size] = host[i, k]
        tf = _h2d(torch.from_numpy(pack_f), self.device)
        ti = _h2d(torch.from_numpy(pack_i), self.device)

        def mk(i):
            return DropScale(tf[0, i], host[i], ti[0, i], ti[1, i], tf[1, i])
        return [(mk(2 * i), mk(2 * i + 1)) for i in range(nr // 2)]
