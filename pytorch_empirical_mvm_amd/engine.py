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
from collections import OrderedDict

import numpy as np
import torch

from . import config as CFG
from . import kernels as K
from . import swin_index as SI

BF16, F32 = torch.bfloat16, torch.float32


# ----------------------------------------------------------------------------------------------------
# parameter arena
# ----------------------------------------------------------------------------------------------------
class ParamStore:
    """All parameters in ONE flat f32 buffer (+ grad, Adam m/v, bf16 compute copy), laid out by optimizer
    group (agent.py:84-113) so clip / AdamW / all-reduce are a handful of launches over contiguous memory."""
    FROZEN = ("enc_img.emb_odr", "emb_task")   # never receive a gradient on the built paths (emb_task: task token off)
    PAD = 16                 # elements: bf16 views 32-byte, fp8 views 16-byte aligned (DMA chunks)
    TAIL = 1 << 16

    def __init__(self, shapes, device, frozen=False):
        self.device = device
        self.frozen = frozen                # frozen teacher arena: parameters + bf16 copy only (no grad / Adam state / W^T copies)
        order = sorted(shapes.keys(), key=lambda n: (4 if n in self.FROZEN else CFG.param_group(n)))   # stable
        self.index = OrderedDict()
        off = 0
        bounds = [0]
        cur_g = 0
        for n in order:
            g = 4 if n in self.FROZEN else CFG.param_group(n)
            while cur_g < g:
                bounds.append(off)
                cur_g += 1
            cnt = int(np.prod(shapes[n]))
            self.index[n] = (off, cnt, tuple(shapes[n]))
            off += -(-cnt // self.PAD) * self.PAD
        while cur_g < 5:
            bounds.append(off)
            cur_g += 1
        self.total = off
        self.segments = [(bounds[i], bounds[i + 1]) for i in range(5)]       # 4 optimizer groups + frozen
        self.n_trainable = bounds[4]
        self.flat = torch.zeros(off + self.TAIL, device=device, dtype=F32)
        self.grad = None if frozen else torch.zeros(off + self.TAIL, device=device, dtype=F32)
        self.m = None if frozen else torch.zeros(off, device=device, dtype=F32)
        self.v = None if frozen else torch.zeros(off, device=device, dtype=F32)
        self.shadow = torch.zeros(off + self.TAIL, device=device, dtype=BF16)
        self.shadowT = torch.zeros(off + self.TAIL, device=device, dtype=BF16) if (torch.device(device).type == "cuda" and not frozen) else None
        self.tmap, self.ttable = {}, None
        self.swin_tail = self._swin_tail_ranges()

    def _swin_tail_ranges(self):
        """[(a, e)] inside the two Swin segments covering the parameters of the LAST TWO stages + the final norm: their
        gradients are final once the backward has left stage n-2 (93% of Swin-B's parameters, with the two memory-bound early
        stages and the patch embedding still to run), so their all-reduce can start there.  Empty when the arena order does
        not keep them contiguous at the end of a segment."""
        stages = sorted({int(n.split(".")[3]) for n in self.index if n.startswith("enc_img.swin.layers.")})
        if len(stages) < 3:
            return []
        lo = stages[-2]
        def is_tail(n):
            if n.startswith("enc_img.swin.norm."):
                return True
            return n.startswith("enc_img.swin.layers.") and int(n.split(".")[3]) >= lo
        out = []
        for gi in (0, 2):
            a, e = self.segments[gi]
            names = [n for n, (o, c, _) in self.index.items() if a <= o < e]
            tail = [n for n in names if is_tail(n)]
            if not tail:
                continue
            split = min(self.index[n][0] for n in tail)
            if all(is_tail(n) for n in names if self.index[n][0] >= split):
                out.append((split, e))
        return out

    def _view(self, buf, n, shape=None):
        o, c, s = self.index[n]
        return buf[o:o + c].view(shape or s)

    pending = None          # event of an optimizer tail still running on the engine's second stream (agent.backward_step): it updates the
                            # non-Swin parameters, their bf16 / W^T copies and zeroes their gradients beside the next Video-Swin forward

    def sync_pending(self):
        """make the current stream wait for that tail.  engine.encode() calls it before the first non-Swin parameter of a step is read;
        every other reader / writer of non-Swin flat / shadow / grad outside the step (refresh_*, load / save, broadcasts, tests that poke
        S.p() / S.g() directly, a second backward_step without a forward) goes through here as well."""
        ev = self.pending
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self.pending = None

    def p(self, n, shape=None):
        return self._view(self.flat, n, shape)

    def g(self, n, shape=None):
        return self._view(self.grad, n, shape)

    def b(self, n, shape=None):
        return self._view(self.shadow, n, shape)

    def fused(self, buf, names, shape):
        """view over ADJACENT parameters (e.g. BERT query/key/value -> one [3H,H] GEMM operand)."""
        o0 = self.index[names[0]][0]
        o = o0
        for n in names:
            assert self.index[n][0] == o, f"{n} is not adjacent in the arena"
            o += self.index[n][1]
        return buf[o0:o].view(shape)

    W8_SCALE = 512.0        # static per-tensor scale of the fp8 weight copies (|w| up to 0.875 before e4m3 saturates at 448)

    def refresh_shadow(self):
        self.sync_pending()
        K.cast_bf16(self.flat[:self.total], self.shadow[:self.total])
        self.refresh_transposed()
        if getattr(self, "shadow8", None) is not None:
            K.cast_fp8(self.shadow[:self.total8], self.W8_SCALE, out=self.shadow8[:self.total8])

    def enable_fp8(self):
        """allocate the e4m3 copy of the arena (BASELINE config 5's fp8 forward GEMMs); refreshed with the bf16 copy"""
        self.sync_pending()
        if getattr(self, "shadow8", None) is None and self.device.type == "cuda":
            self.total8 = -(-self.total // 8) * 8
            self.shadow8 = torch.zeros(self.total8 + self.TAIL, device=self.device, dtype=torch.uint8)
            K.cast_fp8(self.shadow[:self.total8], self.W8_SCALE, out=self.shadow8[:self.total8])

    def b8(self, n, shape=None):
        return self._view(self.shadow8, n, shape)

    def fused8(self, names, shape):
        return self.fused(self.shadow8, names, shape)

    # ---- W^T copies (bf16) of every Linear weight: dgrad dX = dY W then runs as a k-major x k-major GEMM
    def build_transpose_table(self):
        ents = []
        done = set()
        names = list(self.index)
        for n in names:
            o, c, shp = self.index[n]
            if n in done or not n.endswith("weight") or len(shp) < 2 or "embeddings" in n or "patch_embed" in n:
                continue
            N_, K_ = shp[0], int(np.prod(shp[1:]))
            if n.endswith("attention.self.query.weight"):                      # fused [3H,H] (query,key,value adjacent)
                kn, vn = n.replace("query", "key"), n.replace("query", "value")
                if self.index[kn][0] == o + c and self.index[vn][0] == o + 2 * c:
                    N_ *= 3
                    done.update((kn, vn))
            if N_ % 8 or K_ % 8 or N_ < 8:
                continue
            self.tmap[n] = (o, N_, K_)
            for tr in range(-(-N_ // 64)):
                for tc in range(-(-K_ // 64)):
                    ents.append((o, N_, K_, (tr << 16) | tc))
        self.ttable = torch.tensor(ents, dtype=torch.int32, device=self.device).contiguous() if ents else None
        # the same table split by optimizer group family (swin = segments 0 / 2, other = 1 / 3): the two halves of the optimizer tail
        # can then run on different streams (agent.backward_step)
        def in_swin(o):
            return any(a <= o < e for a, e in (self.segments[0], self.segments[2]))
        sw = [e_ for e_ in ents if in_swin(e_[0])]
        ot = [e_ for e_ in ents if not in_swin(e_[0])]
        self.ttable_part = {"swin": torch.tensor(sw, dtype=torch.int32, device=self.device).contiguous() if sw else None,
                            "other": torch.tensor(ot, dtype=torch.int32, device=self.device).contiguous() if ot else None}

    def refresh_transposed(self, which=None):
        """W^T copies from the bf16 shadow; which = None (all) / "swin" / "other" (one optimizer group family)"""
        if self.device.type != "cuda" or self.frozen:
            return
        if self.ttable is None and not self.tmap:
            self.build_transpose_table()
        t = self.ttable if which is None else self.ttable_part[which]
        if t is not None:
            K.transpose_batched(self.shadow, self.shadowT, t)

    def bt(self, n):
        """W^T view [K,N] (or None when the weight has no transposed copy)."""
        if n not in self.tmap:
            return None
        o, N_, K_ = self.tmap[n]
        return self.shadowT[o:o + N_ * K_].view(K_, N_)

    def load_state(self, sd):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)          # (an optimizer tail may still be updating part of the arena on the second stream)
            self.pending = None
        for n, (o, c, s) in self.index.items():
            if n in sd:
                self.flat[o:o + c].copy_(sd[n].reshape(-1).to(self.device, F32))
        self.refresh_shadow()

    def state_dict(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)          # (part of the optimizer tail may still be running on the engine's second stream)
            self.pending = None
        return OrderedDict((n, self.p(n).detach().clone()) for n in self.index)


class V:
    """activation + its gradient slot"""
    __slots__ = ("t", "g")

    def __init__(self, t):
        self.t, self.g = t, None


class DropScale:
    """One DropPath draw of a Swin block branch (video_swin.py:46-54): dev = f32 (B,) scales (0 or 1 / keep) on the device; host = the same
    on the host (which clips were dropped is known WITHOUT a device round trip: the draw happens on the host); kept / dropped = int32
    device lists of the kept (then -1 up to B entries) and of the dropped clip indices; dev_kept = the kept clips' scales, then zeros."""
    __slots__ = ("dev", "host", "kept", "dropped", "dev_kept", "n_kept", "scale")

    def __init__(self, dev, host=None, kept=None, dropped=None, dev_kept=None):
        self.dev = dev
        self.host = dev.detach().float().cpu().numpy() if host is None else host       # (explicit tensors from tests: one small D2H)
        nz = np.flatnonzero(self.host != 0)
        self.n_kept = int(nz.size)
        self.scale = float(self.host[nz[0]]) if nz.size and np.all(self.host[nz] == self.host[nz[0]]) else None      # the ONE scale of the kept clips
        if kept is None:
            B = self.host.size
            lists = np.full((2, B), -1, np.int32)
            lists[0, :nz.size] = nz
            lists[1, :B - nz.size] = np.flatnonzero(self.host == 0)
            t = _dev_i32(lists, dev.device)
            kept, dropped = t[0], t[1]
            sc = np.zeros(B, np.float32)
            sc[:nz.size] = self.host[nz]
            dev_kept = torch.from_numpy(sc).to(dev.device)
        self.kept, self.dropped, self.dev_kept = kept, dropped, dev_kept

    def take(self, rows_per_clip, B):
        """-> (n, clip list, scales): the clips a branch runs on -- the kept ones, then padding clips (list entry -1: every row of theirs is
        a -1 entry of the row maps = zeros in, nothing out; scale 0) until n * rows_per_clip is a multiple of 64: the row count is the K
        dimension of the branch's weight-gradient GEMMs, whose direct-to-LDS kernels need whole K tiles."""
        need = 64 // math.gcd(rows_per_clip, 64)
        return min(B, -(-self.n_kept // need) * need), self.kept, self.dev_kept


def _acc(v, g):
    v.g = g if v.g is None else K.add_bf16(v.g, g)


def _h2d(t, device):
    """host -> device without stalling the launch queue: a copy from pageable memory blocks the host until everything already
    enqueued has run (the staging copy is stream-ordered), which drains the GPU at the start of every step; pinned + non_blocking
    lets the host keep running ahead (the caching host allocator keeps the pinned block alive until the copy has executed)."""
    if torch.device(device).type == "cuda":
        return t.pin_memory().to(device, non_blocking=True)
    return t.to(device)


def _dev_i32(a, device):
    return _h2d(torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)), device)


# ----------------------------------------------------------------------------------------------------
# engine
# ----------------------------------------------------------------------------------------------------
class VioletEngine:
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
        self.dpr = np.linspace(0, CFG.DROP_PATH_RATE, sum(cfg["depths"])).tolist()     # video_swin.py:447
        # BASELINE config 5 ("fp8 MFMA path"): forward GEMMs of the fusion encoder's qkv and FFN-in projections on e4m3 operands
        # (per-tensor static scales, v_mfma_scale_f32_16x16x128_f8f6f4); backward stays bf16 on the bf16 activations
        self.fp8 = bool(cfg.get("fp8_forward", False)) and self.device.type == "cuda"
        self.store_drop_mask = os.environ.get("VMVM_DROP_MASK", "1") != "0"      # fusion attention: the backward reads the forward's dropout decisions (44.8 MB per layer at C2) instead of re-evaluating Philox twice
        self.gelu_code8 = bool(cfg.get("gelu_code8", os.environ.get("VMVM_GELU_CODE8", "1") != "0"))         # Swin MLPs keep GELU' as an 8-bit code
        self.A8_SCALE = 16.0
        if self.fp8:
            self.store.enable_fp8()
        # Weight-gradient stream: dW = dY^T X (+ the bias column sums) depend on nothing downstream in the backward, so they
        # are enqueued on a second HIP stream and run NEXT TO the main stream's chain (input-gradient GEMMs, attention / LayerNorm
        # backward): the tail rounds of the persistent GEMM grids and the memory- / latency-bound kernels between them get filled
        # with MFMA work.  Joined before the gradient exchange / the optimizer.  VMVM_WGRAD_STREAM=0: everything on one stream.
        self.wstream = None
        self.other_ready = None             # event: the non-Swin half of the previous optimizer step (agent.backward_step) has finished on wstream
        self._wpending = False              # weight-gradient launches on wstream the main stream has not waited for yet
        if self.device.type == "cuda":
            self.workspace = torch.empty(192 << 20, device=self.device, dtype=torch.uint8)        # split-K slabs of the wgrad GEMMs
            K.set_workspace(self.workspace)
            if bool(cfg.get("wgrad_stream", os.environ.get("VMVM_WGRAD_STREAM", "1") != "0")):
                self.wstream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("VMVM_WGRAD_PRIO", "0")))
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
        with torch.cuda.stream(self.wstream):
            fn(self.workspace_w)
        for t in operands:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(self.wstream)
        self._wpending = True

    def _wgrad_join(self):
        """the main stream waits for the weight gradients in flight (before the gradient exchange / norm / AdamW read them)"""
        if self.wstream is not None and self._wpending:
            torch.cuda.current_stream().wait_stream(self.wstream)
            self._wpending = False

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
                K.colsum(dy, gb_, row_scale, rows_per_scale, accumulate=True, M=M, N=N)
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

    # -------------------------------------------------------------- Video-Swin
    def _patch_embed(self, img, cov):
        """PatchEmbed3D (video_swin.py:390-407) in one kernel (`vmvm_patch_embed_fwd`: clip read once, cover + zero frame applied on
        the way into the MFMA operands, LayerNorm as the epilogue).  The layer is 0.1% of the FLOPs but sets the precision of
        everything downstream, so the pixels enter as a bf16 hi/lo pair and the conv output stays f32 into the LayerNorm (the
        reference runs this conv in fp16 = 3 more mantissa bits than bf16).  No im2col buffer is kept: the weight gradient
        re-derives its [M,192] = [hi | lo] operand in the backward, where it lives for one GEMM."""
        S, pre = self.store, "enc_img.swin.patch_embed."
        E = self.cfg["embed_dim"]
        wb = S.b(pre + "proj.weight", (E, 96))
        x, z, mean, rstd = K.patch_embed_fwd(img, cov, wb, S.p(pre + "proj.bias"), S.p(pre + "norm.weight"), S.p(pre + "norm.bias"), 1e-5)
        out = V(x)

        def bwd():
            dz, _ = K.layernorm_bwd(out.g, z, S.p(pre + "norm.weight"), mean, rstd, S.g(pre + "norm.weight"), S.g(pre + "norm.bias"))
            K.colsum(dz, S.g(pre + "proj.bias"), accumulate=True)
            cols = K.patch_im2col(img, cov)                                   # [M,192] = [hi | lo]
            K.gemm(dz, cols, a_kmajor=False, b_kmajor=False, M=E, N=96, K=dz.shape[0], out=S.g(pre + "proj.weight", (E, 96)), accumulate=True)
        self.tape.append(bwd)
        return out

    def _swin_block(self, xv, B, dims, C, nh, pre, shifted, dp):
        S, cfg = self.store, self.cfg
        D, H, W = dims
        L = D * H * W
        win = tuple(cfg["window"])
        ws, ss = SI.get_window_size(dims, win, tuple(i // 2 for i in win) if shifted else (0, 0, 0))
        wm, (Dp, Hp, Wp) = SI.window_map(D, H, W, ws, ss)
        N = ws[0] * ws[1] * ws[2]
        Lp = wm.size
        nW = Lp // N
        dev = self.device
        reg_np = SI.region_ids(Dp, Hp, Wp, ws, ss)
        rc_np, rc0 = SI.rc_codes(N, win)
        # win_layout = 1 (include/vmvm.h, swin_index.win3_perm): the order of the tokens INSIDE a window is free -- one gather map serves
        # the LayerNorm, the projection's un-gather epilogue and the backward -- so (8,7,7) windows are laid out d-fastest and region-major,
        # which is what the win3 attention kernels assume (Toeplitz bias reads, masked score tiles skipped); rc / region follow the slots
        w3 = 1 if (SI.win3_ok(ws, ss) and os.environ.get("VMVM_WIN_LAYOUT", "1") != "0") else 0
        pm = SI.win3_perm() if w3 else None               # (applied inside the cached builders: host work once per shape, not per block call)
        src = self._cached(("wm", dims, ws, ss, w3), lambda: _dev_i32(wm.reshape(nW, N)[:, pm].reshape(-1) if w3 else wm, dev))
        reg = None if reg_np is None else self._cached(("reg", Dp, Hp, Wp, ws, ss, w3), lambda: torch.from_numpy(np.ascontiguousarray(reg_np[:, pm]) if w3 else reg_np).to(dev))
        rc = self._cached(("rc", N, win, w3), lambda: _dev_i32(rc_np[pm] if w3 else rc_np, dev))
        scale = 32 ** -0.5 if C // nh == 32 else (C // nh) ** -0.5
        # DropPath (video_swin.py:46-63): the block calls it TWICE -- on the attention branch (:256) and on the MLP branch (:248) -- with
        # independent per-sample draws; `dp` = (scale vector of the attention branch, scale vector of the MLP branch), or one vector for both
        dp, dp2 = (dp if isinstance(dp, (tuple, list)) else (dp, dp))
        # Dead clips of the attention branch: a clip whose DropPath draw is 0 gets x1 = x -- its LayerNorm, qkv, window attention and
        # projection contribute nothing, forward or backward.  The draw is known on the host, so the branch runs on the KEPT clips only:
        # the per-clip window map becomes the absolute row map of the kept clips (vmvm_expand_batch_map) for the gather-LayerNorm, the
        # projection's un-gather epilogue and the backward's gather; the dropped clips' rows are copied (vmvm_copy_batches_bf16).
        ds = dp if (isinstance(dp, DropScale) or dp is None) else DropScale(dp)
        dpv = None if ds is None else ds.dev
        ds2 = dp2 if (isinstance(dp2, DropScale) or dp2 is None) else DropScale(dp2)
        dp2 = None if ds2 is None else ds2.dev
        Bk, compact = B, False                                 # clips the attention branch runs on
        if ds is not None and os.environ.get("VMVM_DROPPATH_DCE", "1") != "0":
            Bk, kept_a, dpk_a = ds.take(math.gcd(L, Lp), B)                   # (both row counts, Bk * L and Bk * Lp, in whole K tiles)
            drop_a, nd_a = ds.dropped, B - ds.n_kept
            compact = Bk < B and ds.scale is not None
            if not compact:
                Bk = B                                         # (nothing to eliminate / VMVM_DROPPATH_DCE=0: dropped clips are scaled by 0)
        x = xv.t
        g1, b1 = S.p(pre + "norm1.weight"), S.p(pre + "norm1.bias")
        table = S.p(pre + "attn.relative_position_bias_table")
        if compact and Bk == 0:                                # every clip dropped: the branch is the identity
            x1, src_k = x, None
        else:
            if compact:
                src_k = K.expand_batch_map(src, kept_a, Bk, L)         # [Bk * Lp] absolute rows (pads stay -1)
                lnkw = dict(M=Bk * Lp, C_=C, nseg=1, src=src_k, rows_out_per_batch=Bk * Lp, rows_in_per_batch=B * L, pad_mode=0)
                mapkw = dict(row_map=src_k, map_len=Bk * Lp, map_stride=0)
                dpk = dpk_a
            else:
                src_k = None
                lnkw = dict(M=B * Lp, C_=C, nseg=1, src=src, rows_out_per_batch=Lp, rows_in_per_batch=L, pad_mode=0)
                mapkw = dict(row_map=src, map_len=Lp, map_stride=L)
                dpk = dpv
            xw, mean1, rstd1 = K.layernorm_fwd(x, g1, b1, 1e-5, **lnkw)
            qkv = K.gemm(xw, S.b(pre + "attn.qkv.weight"), bias=S.p(pre + "attn.qkv.bias"), col_scale=scale, col_scale_n=C)
            akw = dict(q_off=0, k_off=C, v_off=2 * C, bias_table=table, rc=rc, rc0=rc0, region=reg, n_win=nW, seq_scale=dpk, seqs_per_scale=nW, win_layout=w3)
            ao, lse = K.attention_fwd(qkv, Bk * nW, N, nh, C // nh, 0, scale, **akw)
            x1 = K.gemm(ao, S.b(pre + "attn.proj.weight"), bias=S.p(pre + "attn.proj.bias"), row_scale=dpk, rows_per_scale=Lp,
                        scale_bias_only=True, resid=x, out_rows=B * L, **mapkw)
            if compact:
                K.copy_batches(x, x1, drop_a, nd_a, L)                # identity path of the dropped clips
        g2, b2 = S.p(pre + "norm2.weight"), S.p(pre + "norm2.bias")
        # saved for the GELU backward: an 8-bit code of GELU'(fc1 output) (vmvm_gemm_desc.aux_code8) where the persistent kernel's
        # whole-K-tile staging applies, the bf16 pre-activation otherwise
        c8 = getattr(self, "gelu_code8", True) and C % 64 == 0 and not S.frozen
        # the MLP branch on ITS kept clips (the second, independent draw): LayerNorm through an absolute identity map of the kept clips'
        # rows, fc1 compact, fc2 scattering back through the same map (+ residual); one extra gather of d(x2) in the backward
        Bm, compact2 = B, False
        if ds2 is not None and os.environ.get("VMVM_DROPPATH_DCE", "1") not in ("0", "attn"):
            Bm, kept_m, dpk_m = ds2.take(L, B)
            drop_m, nd_m = ds2.dropped, B - ds2.n_kept
            compact2 = Bm < B and ds2.scale is not None
            if not compact2:
                Bm = B
        if compact2 and Bm == 0:
            x2 = x1
        else:
            if compact2:
                idm = self._cached(("idmap", L), lambda: _dev_i32(np.arange(L), dev))
                map_m = K.expand_batch_map(idm, kept_m, Bm, L)
                dpm = dpk_m
                y2, mean2, rstd2 = K.layernorm_fwd(x1, g2, b2, 1e-5, M=Bm * L, C_=C, nseg=1, src=map_m, rows_out_per_batch=Bm * L, rows_in_per_batch=B * L,
                                                   pad_mode=0)
                mkw = dict(row_map=map_m, map_len=Bm * L, map_stride=0, out_rows=B * L)
            else:
                map_m, dpm, mkw = None, dp2, {}
                y2, mean2, rstd2 = K.layernorm_fwd(x1, g2, b2, 1e-5)
            u = None if S.frozen else torch.empty((Bm * L, 4 * C), device=dev, dtype=torch.uint8 if c8 else BF16)     # frozen teacher: no backward, nothing saved
            h = K.gemm(y2, S.b(pre + "mlp.fc1.weight"), bias=S.p(pre + "mlp.fc1.bias"), act=1, out_preact=u, row_scale=dpm, rows_per_scale=L, code8=c8)
            x2 = K.gemm(h, S.b(pre + "mlp.fc2.weight"), bias=S.p(pre + "mlp.fc2.bias"), row_scale=dpm, rows_per_scale=L,
                        scale_bias_only=True, resid=x1, **mkw)
            if compact2:
                K.copy_batches(x1, x2, drop_m, nd_m, L)
        out = V(x2)

        def bwd():
            dx2 = out.g
            if compact2 and Bm == 0:
                dx1 = dx2
            else:
                dx2c = K.gather_rows(dx2, map_m, Bm * L) if compact2 else dx2
                # (compact: the padding clips' rows of dx2c are zeros and the kept clips share one scale -> the bias gradient stays fused)
                du = self._linear_bwd(dx2c, h, pre + "mlp.fc2.weight", pre + "mlp.fc2.bias", row_scale=dpm, rows_per_scale=L, cs_scale=ds2.scale if (compact2 or (ds2 is not None and ds2.n_kept == B)) else None,
                                      dx_kw=dict(act=3, aux=u, row_scale=dpm, rows_per_scale=L, code8=c8))
                dy2 = self._linear_bwd(du, y2, pre + "mlp.fc1.weight", pre + "mlp.fc1.bias")
                if compact2:
                    dx1, _ = K.layernorm_bwd(dy2, x1, g2, mean2, rstd2, S.g(pre + "norm2.weight"), S.g(pre + "norm2.bias"), rows_in=B * L, nseg=1,
                                             src=map_m, rows_out_per_batch=Bm * L, rows_in_per_batch=B * L, pad_mode=0, dX_add=dx2)
                    K.copy_batches(dx2, dx1, drop_m, nd_m, L)
                else:
                    dx1, _ = K.layernorm_bwd(dy2, x1, g2, mean2, rstd2, S.g(pre + "norm2.weight"), S.g(pre + "norm2.bias"), dX_add=dx2)
            if compact and Bk == 0:
                _acc(xv, dx1)
                return
            dx1w = K.gather_rows(dx1, src_k, Bk * Lp) if compact else K.gather_rows(dx1, src, B * Lp, Lp, L)
            dao = self._linear_bwd(dx1w, ao, pre + "attn.proj.weight", pre + "attn.proj.bias", row_scale=dpk, rows_per_scale=Lp,
                                   cs_scale=ds.scale if (compact or (ds is not None and ds.n_kept == B)) else None)      # (every clip kept: one scale as well)
            dqkv = K.attention_bwd(dao, qkv, ao, lse, Bk * nW, N, nh, C // nh, 0, scale,
                                   dbias_table=S.g(pre + "attn.relative_position_bias_table"), **akw)
            dxw = self._linear_bwd(dqkv, xw, pre + "attn.qkv.weight", pre + "attn.qkv.bias")
            bkw = dict(src=src_k, rows_out_per_batch=Bk * Lp, rows_in_per_batch=B * L) if compact else dict(src=src, rows_out_per_batch=Lp, rows_in_per_batch=L)
            dx, _ = K.layernorm_bwd(dxw, x, g1, mean1, rstd1, S.g(pre + "norm1.weight"), S.g(pre + "norm1.bias"), rows_in=B * L, nseg=1,
                                    pad_mode=0, dX_add=dx1, **bkw)
            if compact:
                K.copy_batches(dx1, dx, drop_a, nd_a, L)              # d(x) of the dropped clips = d(x1)
            _acc(xv, dx)
        self.tape.append(bwd)
        return out

    def _patch_merge(self, xv, B, dims, C, pre):
        S = self.store
        D, H, W = dims
        mm, (D2, H2, W2) = SI.merge_map(D, H, W)
        src = self._cached(("mm", dims), lambda: _dev_i32(mm, self.device))
        L, Lo = D * H * W, D2 * H2 * W2
        x = xv.t
        gam, bet = S.p(pre + "norm.weight"), S.p(pre + "norm.bias")
        y, mean, rstd = K.layernorm_fwd(x, gam, bet, 1e-5, M=B * Lo, C_=4 * C, nseg=4, src=src, rows_out_per_batch=Lo, rows_in_per_batch=L, pad_mode=1)
        o = K.gemm(y, S.b(pre + "reduction.weight"))
        out = V(o)

        def bwd():
            dy = self._linear_bwd(out.g, y, pre + "reduction.weight", None)
            dx, _ = K.layernorm_bwd(dy, x, gam, mean, rstd, S.g(pre + "norm.weight"), S.g(pre + "norm.bias"), rows_in=B * L, nseg=4, src=src,
                                    rows_out_per_batch=Lo, rows_in_per_batch=L, pad_mode=1)
            _acc(xv, dx)
        self.tape.append(bwd)
        return out, (D2, H2, W2)

    def swin_forward(self, img, cov, dp_all, final_norm=True):
        """img f32 (B,T,3,H,W) -> V([B*T*h*w, 8E]) channels-last tokens (after the final norm; `final_norm=False`: the last
        stage's output, what HF SwinModel reports as hidden_states[-1])."""
        cfg, S = self.cfg, self.store
        B, T, _, H, W = img.shape
        xv = self._patch_embed(img, cov)
        dims = (T, H // 4, W // 4)
        C = cfg["embed_dim"]
        blk = 0
        n_st = len(cfg["depths"])
        for i, (d, nh) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
            if i == n_st - 2 and n_st >= 3:
                # runs in the backward right after stage n-2's last block: the gradients of stages >= n-2 are final
                self.tape.append(lambda: self.on_swin_tail_ready() if self.on_swin_tail_ready is not None else None)
            for b in range(d):
                dp = None if dp_all is None else dp_all[blk]
                xv = self._swin_block(xv, B, dims, C, nh, f"enc_img.swin.layers.{i}.blocks.{b}.", b % 2 == 1, dp)
                blk += 1
            if i < len(cfg["depths"]) - 1:
                xv, dims = self._patch_merge(xv, B, dims, C, f"enc_img.swin.layers.{i}.downsample.")
                C *= 2
        if not final_norm:
            return xv, dims, C
        x = xv.t
        gam, bet = S.p("enc_img.swin.norm.weight"), S.p("enc_img.swin.norm.bias")
        y, mean, rstd = K.layernorm_fwd(x, gam, bet, 1e-5)
        out = V(y)
        inp = xv

        def bwd():
            dx, _ = K.layernorm_bwd(out.g, x, gam, mean, rstd, S.g("enc_img.swin.norm.weight"), S.g("enc_img.swin.norm.bias"))
            _acc(inp, dx)
        self.tape.append(bwd)
        return out, dims, C

    # -------------------------------------------------------------- EncVideo / EncTxt  -> one token pool
    def _drop_on(self, site, train):
        return bool(train) and (self._drop_sites is None or site in self._drop_sites)

    def encode(self, img, cov, txt, dp_all, train, odr=None):
        """returns pool V([B*Lv + NT*X, 768]) : rows [0, B*Lv) = feat_img (model.py:71), rest = feat_txt (model.py:107) of the NT =
        txt.shape[0] text sequences (NT = B in pre-training; B*O option sequences in multiple-choice QA)."""
        cfg, S, dev = self.cfg, self.store, self.device
        B, T, _, H, W = img.shape
        X = txt.shape[1]
        Hd = cfg["hidden"]
        sw, dims, C8 = self.swin_forward(img, cov, dp_all)
        self.store.sync_pending()           # everything below reads non-Swin parameters: their AdamW update ran beside the Swin forward
        self.other_ready = None
        hw = dims[1] * dims[2]
        assert dims[1] == H // 32 and dims[2] == W // 32                       # model.py:34 hard-codes //32
        Lv = T * (1 + hw)
        has_fc = "enc_img.fc.weight" in S.index
        f = K.gemm(sw.t, S.b("enc_img.fc.weight"), bias=S.p("enc_img.fc.bias")) if has_fc else sw.t
        pos = S.p("enc_img.emb_pos", (1 + cfg["max_size_patch"] ** 2, Hd))
        ln_ = S.p("enc_img.emb_len", (cfg["max_size_frame"], Hd))
        if T > cfg["max_size_frame"]:
            raise RuntimeError(f"max_size_frame ({cfg['max_size_frame']}) must be >= T ({T})  (model.py:69)")
        if odr is None:
            pre = K.encvideo_assemble(f, S.p("enc_img.emb_cls", (Hd,)), pos, ln_, B, T, hw, Hd)
        else:
            # frame-order variant (model.py:61-67; no caller in the reference sets it, inference surface only): slot i of clip b adds
            # emb_len[i] when odr[b][i] == i, else emb_odr -- one frame table per clip, the same kernel on one clip at a time
            eo = S.p("enc_img.emb_odr", (1, Hd))
            hit = torch.as_tensor([[int(p_) == i for i, p_ in enumerate(o)] for o in odr], device=dev).view(B, T, 1)
            tabs = torch.where(hit, ln_[:T].unsqueeze(0), eo.unsqueeze(0)).contiguous()                  # f32 [B, T, Hd]
            pre = torch.empty((B * T * (1 + hw), Hd), device=dev, dtype=BF16)
            for b in range(B):
                K.encvideo_assemble(f[b * T * hw:(b + 1) * T * hw], S.p("enc_img.emb_cls", (Hd,)), pos, tabs[b], 1, T, hw, Hd,
                                    out=pre[b * T * (1 + hw):(b + 1) * T * (1 + hw)])
        pool = torch.empty((B * Lv + txt.shape[0] * X, Hd), device=dev, dtype=BF16)
        gi, bi = S.p("enc_img.norm.weight"), S.p("enc_img.norm.bias")
        fi, mean_i, rstd_i = K.layernorm_fwd(pre, gi, bi, 1e-5)
        pool[:B * Lv].copy_(fi)
        # text: HF BertEmbeddings (word + position + token_type(0)) -> LayerNorm(1e-12) -> dropout(0.1)
        pt = "enc_txt.emb_txt."
        e = K.bert_embed(txt, S.p(pt + "word_embeddings.weight"), S.p(pt + "position_embeddings.weight"),
                         S.p(pt + "token_type_embeddings.weight")[0])
        gt, bt = S.p(pt + "LayerNorm.weight"), S.p(pt + "LayerNorm.bias")
        ft, mean_t, rstd_t = K.layernorm_fwd(e, gt, bt, CFG.BERT["eps"])
        p_drop = CFG.BERT["hidden_dropout"] if self._drop_on("emb", train) else 0.0
        off_t = self._next_offset(ft.numel())
        self.last_offsets["emb"] = off_t
        if p_drop > 0:
            ft = K.dropout(ft, p_drop, self.seed, off_t)
        pool[B * Lv:].copy_(ft)
        out = V(pool)

        def bwd():
            if odr is not None:
                raise RuntimeError("odr is served on the inference surface only (go_feat / EncVideo.forward); no training path of the reference sets it")
            dpool = out.g                                                       # bf16 [B*Lv + B*X, Hd]
            dft = dpool[B * Lv:]
            if p_drop > 0:
                dft = K.dropout(dft, p_drop, self.seed, off_t)
            de, _ = K.layernorm_bwd(dft, e, gt, mean_t, rstd_t, S.g(pt + "LayerNorm.weight"), S.g(pt + "LayerNorm.bias"))
            K.bert_embed_bwd(txt, de, S.g(pt + "word_embeddings.weight"), S.g(pt + "position_embeddings.weight"),
                             S.g(pt + "token_type_embeddings.weight")[0])
            dpre, _ = K.layernorm_bwd(dpool[:B * Lv], pre, gi, mean_i, rstd_i, S.g("enc_img.norm.weight"), S.g("enc_img.norm.bias"))
            df = K.encvideo_assemble_bwd(dpre, S.g("enc_img.emb_cls", (Hd,)), S.g("enc_img.emb_pos", (1 + cfg["max_size_patch"] ** 2, Hd)),
                                         S.g("enc_img.emb_len", (cfg["max_size_frame"], Hd)), B, T, hw, Hd)
            if has_fc:
                df = self._linear_bwd(df, sw.t, "enc_img.fc.weight", "enc_img.fc.bias")
            _acc(sw, df)
        self.tape.append(bwd)
        return out, Lv, hw

    # -------------------------------------------------------------- fusion encoder
    def _bert_layer(self, xv, nseq, Lq, keymask, l, train, causal_from=0, att_out=None):
        S, dev = self.store, self.device
        pre = f"trsfr.layer.{l}."
        Hd, nh = self.cfg["hidden"], CFG.BERT["heads"]
        qn = [pre + f"attention.self.{n}.weight" for n in ("query", "key", "value")]
        bn = [pre + f"attention.self.{n}.bias" for n in ("query", "key", "value")]
        Wqkv, Gqkv = S.fused(S.shadow, qn, (3 * Hd, Hd)), S.fused(S.grad, qn, (3 * Hd, Hd))
        bqkv, gbqkv = S.fused(S.flat, bn, (3 * Hd,)), S.fused(S.grad, bn, (3 * Hd,))
        p_h = CFG.BERT["hidden_dropout"] if train else 0.0
        p_a = CFG.BERT["attn_dropout"] if train else 0.0
        M = nseq * Lq
        x = xv.t
        a8 = 1.0 / (self.A8_SCALE * S.W8_SCALE)
        if self.fp8:
            qkv = K.gemm(K.cast_fp8(x, self.A8_SCALE), S.fused8(qn, (3 * Hd, Hd)), bias=bqkv, fp8=True, alpha=a8)
        else:
            qkv = K.gemm(x, Wqkv, bias=bqkv)
        o_att = self._next_offset(nseq * nh * Lq * Lq)
        akw = dict(q_off=0, k_off=Hd, v_off=2 * Hd, keymask=keymask, dropout_p=p_a, seed=self.seed, offset=o_att, causal_from=causal_from)
        if p_a > 0 and self.store_drop_mask:                     # the forward's keep / drop decisions, read back by both backward kernels
            akw["drop_mask"] = K.attention_drop_mask(nseq, Lq, nh, Hd // nh, 1, p_a, dev, causal_from=causal_from, att_colsum=att_out)
        ctx, lse = K.attention_fwd(qkv, nseq, Lq, nh, Hd // nh, 1, 1.0 / math.sqrt(Hd // nh), att_colsum=att_out, **akw)
        o1 = self._next_offset(M * Hd)
        a = K.gemm(ctx, S.b(pre + "attention.output.dense.weight"), bias=S.p(pre + "attention.output.dense.bias"), resid=x,
                   dropout_p=p_h, seed=self.seed, offset=o1)
        g1, b1 = S.p(pre + "attention.output.LayerNorm.weight"), S.p(pre + "attention.output.LayerNorm.bias")
        x1, mean1, rstd1 = K.layernorm_fwd(a, g1, b1, CFG.BERT["eps"])
        c8 = self.gelu_code8                                     # GELU' saved as an 8-bit code (as in the Swin MLPs)
        u = torch.empty((M, CFG.BERT["ffn"]), device=dev, dtype=torch.uint8 if c8 else BF16)
        if self.fp8:
            h = K.gemm(K.cast_fp8(x1, self.A8_SCALE), S.b8(pre + "intermediate.dense.weight"), bias=S.p(pre + "intermediate.dense.bias"),
                       act=1, out_preact=u, fp8=True, alpha=a8, code8=c8)
        else:
            h = K.gemm(x1, S.b(pre + "intermediate.dense.weight"), bias=S.p(pre + "intermediate.dense.bias"), act=1, out_preact=u, code8=c8)
        o2 = self._next_offset(M * Hd)
        f = K.gemm(h, S.b(pre + "output.dense.weight"), bias=S.p(pre + "output.dense.bias"), resid=x1, dropout_p=p_h, seed=self.seed, offset=o2)
        g2, b2 = S.p(pre + "output.LayerNorm.weight"), S.p(pre + "output.LayerNorm.bias")
        x2, mean2, rstd2 = K.layernorm_fwd(f, g2, b2, CFG.BERT["eps"])
        out = V(x2)

        def bwd():
            df, dfm = K.layernorm_bwd(out.g, f, g2, mean2, rstd2, S.g(pre + "output.LayerNorm.weight"), S.g(pre + "output.LayerNorm.bias"),
                                      want_dX2=p_h > 0, dropout_p=p_h, seed=self.seed, offset=o2)
            dfm = df if dfm is None else dfm
            du = self._linear_bwd(dfm, h, pre + "output.dense.weight", pre + "output.dense.bias", dx_kw=dict(act=3, aux=u, code8=c8))
            dx1 = self._linear_bwd(du, x1, pre + "intermediate.dense.weight", pre + "intermediate.dense.bias", dx_kw=dict(resid=df))
            da, dam = K.layernorm_bwd(dx1, a, g1, mean1, rstd1, S.g(pre + "attention.output.LayerNorm.weight"),
                                      S.g(pre + "attention.output.LayerNorm.bias"), want_dX2=p_h > 0, dropout_p=p_h, seed=self.seed, offset=o1)
            dam = da if dam is None else dam
            dctx = self._linear_bwd(dam, ctx, pre + "attention.output.dense.weight", pre + "attention.output.dense.bias")
            dqkv = K.attention_bwd(dctx, qkv, ctx, lse, nseq, Lq, nh, Hd // nh, 1, 1.0 / math.sqrt(Hd // nh), **akw)
            dx = self._linear_bwd(dqkv, x, None, None, w=Wqkv, gw=Gqkv, gb=gbqkv, dx_kw=dict(resid=da), wT=S.bt(qn[0]))
            _acc(xv, dx)
        self.tape.append(bwd)
        return out

    def _bert_layer_qrow(self, xv, nseq, Lq, qpos, keymask, l, train):
        """HF BertLayer `l` for sequences of which ONLY the output at position `qpos` is read -- the VTM pass reads the encoder's last
        hidden state at the text [CLS] position (main_pretrain.py:260: out[:, T*(1+hw), :]), so in the LAST fusion layer every other
        query row of those sequences is dead code, forward and backward (their d(out) is zero).  K and V of every position are still
        computed (one GEMM on the key / value rows of the fused QKV weight); the query, the attention row (`vmvm_attn_query_row_*`),
        both dense layers, the FFN and both LayerNorms run on nseq rows instead of nseq * Lq.  Same arithmetic per row as
        `_bert_layer`; returns V([nseq, H])."""
        S, dev = self.store, self.device
        pre = f"trsfr.layer.{l}."
        Hd, nh = self.cfg["hidden"], CFG.BERT["heads"]
        hd = Hd // nh
        qn = [pre + f"attention.self.{n}.weight" for n in ("query", "key", "value")]
        bn = [pre + f"attention.self.{n}.bias" for n in ("query", "key", "value")]
        Wqkv, Gqkv = S.fused(S.shadow, qn, (3 * Hd, Hd)), S.fused(S.grad, qn, (3 * Hd, Hd))
        bqkv, gbqkv = S.fused(S.flat, bn, (3 * Hd,)), S.fused(S.grad, bn, (3 * Hd,))
        WT = S.bt(qn[0])                                                  # fused W^T [H, 3H] (or None)
        p_h = CFG.BERT["hidden_dropout"] if train else 0.0
        p_a = CFG.BERT["attn_dropout"] if train else 0.0
        scale = 1.0 / math.sqrt(hd)
        x = xv.t                                                          # [nseq * Lq, H]
        rows = self._cached(("qrow", nseq, Lq, qpos), lambda: _dev_i32(np.arange(nseq) * Lq + qpos, dev))
        kv = K.gemm(x, Wqkv[Hd:], bias=bqkv[Hd:])                         # K | V of every position  [nseq * Lq, 2H]
        xc = K.gather_rows(x, rows, nseq)                                 # the query rows  [nseq, H]
        q = K.gemm(xc, Wqkv[:Hd], bias=bqkv[:Hd])
        o_att = self._next_offset(nseq * nh * Lq)
        ctx, pr, prd = K.attn_query_row_fwd(q, kv, nseq, Lq, nh, hd, scale, k_off=0, v_off=Hd, keymask=keymask, dropout_p=p_a, seed=self.seed, offset=o_att)
        o1 = self._next_offset(nseq * Hd)
        a = K.gemm(ctx, S.b(pre + "attention.output.dense.weight"), bias=S.p(pre + "attention.output.dense.bias"), resid=xc,
                   dropout_p=p_h, seed=self.seed, offset=o1)
        g1, b1 = S.p(pre + "attention.output.LayerNorm.weight"), S.p(pre + "attention.output.LayerNorm.bias")
        x1, mean1, rstd1 = K.layernorm_fwd(a, g1, b1, CFG.BERT["eps"])
        u = torch.empty((nseq, CFG.BERT["ffn"]), device=dev, dtype=BF16)
        h = K.gemm(x1, S.b(pre + "intermediate.dense.weight"), bias=S.p(pre + "intermediate.dense.bias"), act=1, out_preact=u)
        o2 = self._next_offset(nseq * Hd)
        f = K.gemm(h, S.b(pre + "output.dense.weight"), bias=S.p(pre + "output.dense.bias"), resid=x1, dropout_p=p_h, seed=self.seed, offset=o2)
        g2, b2 = S.p(pre + "output.LayerNorm.weight"), S.p(pre + "output.LayerNorm.bias")
        x2, mean2, rstd2 = K.layernorm_fwd(f, g2, b2, CFG.BERT["eps"])
        out = V(x2)

        def bwd():
            df, dfm = K.layernorm_bwd(out.g, f, g2, mean2, rstd2, S.g(pre + "output.LayerNorm.weight"), S.g(pre + "output.LayerNorm.bias"),
                                      want_dX2=p_h > 0, dropout_p=p_h, seed=self.seed, offset=o2)
            dfm = df if dfm is None else dfm
            du = self._linear_bwd(dfm, h, pre + "output.dense.weight", pre + "output.dense.bias", dx_kw=dict(act=3, aux=u))
            dx1 = self._linear_bwd(du, x1, pre + "intermediate.dense.weight", pre + "intermediate.dense.bias", dx_kw=dict(resid=df))
            da, dam = K.layernorm_bwd(dx1, a, g1, mean1, rstd1, S.g(pre + "attention.output.LayerNorm.weight"),
                                      S.g(pre + "attention.output.LayerNorm.bias"), want_dX2=p_h > 0, dropout_p=p_h, seed=self.seed, offset=o1)
            dam = da if dam is None else dam
            dctx = self._linear_bwd(dam, ctx, pre + "attention.output.dense.weight", pre + "attention.output.dense.bias")
            dq, dkv = K.attn_query_row_bwd(dctx, q, kv, pr, prd, nseq, Lq, nh, hd, scale, k_off=0, v_off=Hd)
            # query projection: d(xc) = dq Wq + da (the residual of the attention block); key / value projection: d(x) = dkv Wkv
            dxc = self._linear_bwd(dq, xc, None, None, w=Wqkv[:Hd], gw=Gqkv[:Hd], gb=gbqkv[:Hd], dx_kw=dict(resid=da),
                                   wT=None if WT is None else WT[:, :Hd])
            dx = self._linear_bwd(dkv, x, None, None, w=Wqkv[Hd:], gw=Gqkv[Hd:], gb=gbqkv[Hd:], wT=None if WT is None else WT[:, Hd:])
            dx.index_add_(0, rows.long(), dxc)                            # (nseq rows; plumbing)
            _acc(xv, dx)
        self.tape.append(bwd)
        return out

    def go_cross(self, pool, idx, keymask, nseq, Lq, train, causal_from=0, att_out=None, qrow_split=None):
        """gather the [img;txt] sequences from the token pool and run the 12 fusion layers (model.py:204-214).
        causal_from = Lv: the seq2seq mask of the smtm pass (main_pretrain.py:217-224, model.py:191-199)."""
        Hd = self.cfg["hidden"]
        x = K.gather_rows(pool.t, idx, nseq * Lq)
        xv = V(x)
        cur = xv
        nl = self.cfg["bert_layers"]
        for l in range(nl - 1 if qrow_split is not None else nl):
            cur = self._bert_layer(cur, nseq, Lq, keymask, l, train, causal_from, att_out)
        if qrow_split is None:
            return cur, xv, idx
        # last layer: the first n1 sequences in full, of the others only the row at `qpos` (see _bert_layer_qrow)
        n1, qpos = qrow_split
        if causal_from != 0 or att_out is not None:     # _bert_layer_qrow has neither the seq2seq mask nor the attention capture
            raise RuntimeError("go_cross(qrow_split=...) serves the plain key-mask pass only (no causal_from / att_out)")
        if n1 == 0:                               # every sequence: only the row at `qpos` (retrieval / open-ended QA read the text [CLS] state only)
            return (None, self._bert_layer_qrow(cur, nseq, Lq, qpos, keymask, nl - 1, train)), xv, idx
        xa, xb = V(cur.t[:n1 * Lq]), V(cur.t[n1 * Lq:])
        prev = cur

        def join():                               # runs AFTER the two halves' backward closures: d(layer input) = their rows side by side
            _acc(prev, torch.cat([xa.g, xb.g], 0))
        self.tape.append(join)
        out_a = self._bert_layer(xa, n1, Lq, keymask[:n1], nl - 1, train, causal_from, att_out)
        out_b = self._bert_layer_qrow(xb, nseq - n1, Lq, qpos, keymask[n1:], nl - 1, train)
        return (out_a, out_b), xv, idx

    @torch.no_grad()
    def get_att(self, img, txt, mask, train=True, dp_all=None, cov=None):
        """VIOLET_Pretrain.get_att (main_pretrain.py:211-215): one (img_i, txt_i) fusion pass whose attention kernels also
        accumulate the head-averaged column sums of every layer -> (B, T*(1+hw)+X) f32, the sampling weights of the 'am' masking.
        `train` keeps dropout / DropPath on, as the reference calls it from masking() with the model in train mode."""
        dev = self.device
        B, T, _, H, W = img.shape
        X = txt.shape[1]
        saved, self.tape = self.tape, []
        if train and dp_all is None:
            dp_all = self.sample_drop_path(B)
        cov_d = None if cov is None else cov.to(dev, torch.uint8).contiguous()
        pool, Lv, hw = self.encode(img.to(dev, F32).contiguous(), cov_d, txt.to(dev).contiguous(), dp_all, train)
        Lq = Lv + X
        ar_v, ar_t = np.arange(Lv), np.arange(X)
        idx1 = _dev_i32(np.concatenate([np.concatenate([i * Lv + ar_v, B * Lv + i * X + ar_t]) for i in range(B)]), dev)
        km1 = torch.cat([torch.ones(B, Lv, dtype=torch.uint8, device=dev), (mask.to(dev) != 0).to(torch.uint8)], 1).contiguous()
        att = torch.zeros((B, Lq), device=dev, dtype=F32)
        self.go_cross(pool, idx1, km1, B, Lq, train, att_out=att)
        self.tape = saved
        return att

    # -------------------------------------------------------------- MLM head (HF BertOnlyMLMHead), shared by every pass that reads it
    def _mlm_dims(self):
        Vv = self.cfg["vocab"]
        return Vv, -(-Vv // 8) * 8, -(-Vv // 4) * 4          # vocabulary, row pitch of the f32 logits, columns the GEMM writes

    def _mlm_head_fwd(self, rows, n_rows, target, loss, want_grad):
        """dense + GELU + LayerNorm + decoder (+ bias) + cross entropy(ignore -1) on `rows` [n_rows, H] (main_pretrain.py:236,560)."""
        S, dev, Hd = self.store, self.device, self.cfg["hidden"]
        pm = "fc_mtm.predictions."
        Vv, Vpad, Nlog = self._mlm_dims()
        u_ = torch.empty((n_rows, Hd), device=dev, dtype=BF16)
        t_ = K.gemm(rows, S.b(pm + "transform.dense.weight"), bias=S.p(pm + "transform.dense.bias"), act=1, out_preact=u_)
        gm, bm = S.p(pm + "transform.LayerNorm.weight"), S.p(pm + "transform.LayerNorm.bias")
        tn_, mean_, rstd_ = K.layernorm_fwd(t_, gm, bm, CFG.BERT["eps"])
        lg_ = torch.empty((n_rows, Vpad), device=dev, dtype=F32)
        K.gemm(tn_, S.b(pm + "decoder.weight"), N=Nlog, bias=S.p(pm + "bias"), out=lg_)
        dlog_ = K.cross_entropy(lg_, Vv, target, loss, want_grad=want_grad, ld_d=Vpad)
        return dict(r=rows, u=u_, t=t_, tn=tn_, mean=mean_, rstd=rstd_, logits=lg_, dlog=dlog_, n=n_rows)

    def _mlm_head_bwd(self, hd, dx_out=None):
        """head gradients (accumulated into the shared fc_mtm.* tensors); returns / writes d(rows)."""
        S, dev, Hd = self.store, self.device, self.cfg["hidden"]
        pm = "fc_mtm.predictions."
        Vv, Vpad, _ = self._mlm_dims()
        n = hd["n"]
        gm = S.p(pm + "transform.LayerNorm.weight")
        Wdec = S.b(pm + "decoder.weight")
        def dec_wgrad(ws):
            K.colsum(hd["dlog"], S.g(pm + "bias"), accumulate=True, M=n, N=Vpad)     # pad columns are zero and land in arena padding
            K.gemm(hd["dlog"], hd["tn"], a_kmajor=False, b_kmajor=False, M=Vv, N=Hd, K=n, out=S.g(pm + "decoder.weight"), accumulate=True, workspace=ws)
        self._wgrad_launch(dec_wgrad, (hd["dlog"], hd["tn"]))
        # d(tn) = dlog . W over K = the PADDED vocabulary when that is a whole number of 64-wide K tiles (30522 -> 30528): the GEMM
        # then takes the direct-to-LDS kernel instead of the K % 64 != 0 fallback (497 -> 60 us).  Invariants this relies on:
        #  (i) the pad columns [Vv, Vpad) of dlog are exact zeros (vmvm_cross_entropy writes them);
        #  (ii) the (Vpad - Vv) extra rows of the [Vpad, H] weight view lie INSIDE the bf16 arena (other parameters or its zero tail:
        #       (Vpad - Vv) * H <= ParamStore.TAIL) and are FINITE, so 0 * w = 0 -- checked after every optimizer step by the
        #       clip coefficient being finite (a non-finite parameter makes every loss NaN long before it matters here).
        # 48 output tiles and a 30528-long reduction: as an f32 accumulation the GEMM splits K over the chip (444 -> ~70 us), its
        # partial slabs going through the engine's split-K workspace and a fixed-order reduce (run-to-run deterministic).
        Kdec = Vpad if (Vpad % 64 == 0 and (Vpad - Vv) * Hd <= S.TAIL) else Vv
        dtn32 = torch.zeros((n, Hd), device=dev, dtype=F32)
        K.gemm(hd["dlog"], Wdec, b_kmajor=False, M=n, N=Hd, K=Kdec, out=dtn32, accumulate=True)
        dtn = dtn32.to(BF16)
        dt_, _ = K.layernorm_bwd(dtn, hd["t"], gm, hd["mean"], hd["rstd"], S.g(pm + "transform.LayerNorm.weight"), S.g(pm + "transform.LayerNorm.bias"))
        du_ = K.gelu_bwd(dt_, hd["u"])
        return self._linear_bwd(du_, hd["r"], pm + "transform.dense.weight", pm + "transform.dense.bias",
                                dx_kw=None if dx_out is None else dict(out=dx_out))

    # -------------------------------------------------------------- full step
    def forward_backward(self, batch, negatives=None, train=True, dp_all=None, want_outputs=False, backward=True,
                         dropout=None, on_other_grads_ready=None):
        """One pass of the hot path.  batch: img f32 (B,T,3,H,W) UN-masked, cov u8 (B,T,h,w), txt i64 (B,X) (masked ids),
        mask i64 (B,X), ans_mtm i64 (B,X).  Returns dict of loss scalars (device f32 tensors) and optional outputs."""
        cfg, S, dev = self.cfg, self.store, self.device
        img, cov, txt, mask, ans_mtm = batch["img"], batch["cov"], batch["txt"], batch["mask"], batch["ans_mtm"]
        B, T, _, H, W = img.shape
        X = txt.shape[1]
        Hd = cfg["hidden"]
        O = min(B, 4)
        self.tape = []
        if train and dp_all is None:
            dp_all = self.sample_drop_path(B)
        # dropout sites follow `train` unless overridden: dropout = False / True for all of them, or a collection of site names out of
        # {"emb" (BertEmbeddings), "fusion" (the 12 BertLayers), "vtm" (the VTM head's Dropout, main_pretrain.py:146)} -- parity tests
        # switch sites on one group at a time and feed the kernels' own masks to the oracle
        self._drop_sites = None if (dropout is None or isinstance(dropout, bool)) else frozenset(dropout)
        train = train if (dropout is None or self._drop_sites is not None) else bool(dropout)
        feat_target = batch.get("feature_target")
        if feat_target is None and self.feature_teacher is not None:
            feat_target = self.feature_teacher.features(img)     # frozen Swin teacher first: its activations are gone before the student's pile up
        pool, Lv, hw = self.encode(img, cov, txt, dp_all, train)
        Lq = Lv + X
        # ---- sequence assembly indices (pass 1: (img_i, txt_i); pass 2: (img_i, txt_i), (img_i, txt_neg) ...)
        if negatives is None:
            negatives = self.sample_negatives(B)
        ar_v, ar_t = np.arange(Lv), np.arange(X)
        idx1 = np.concatenate([np.concatenate([i * Lv + ar_v, B * Lv + i * X + ar_t]) for i in range(B)])
        pairs = []
        for i in range(B):
            pairs.append((i, i))
            for k in range(O - 1):
                pairs.append((i, int(negatives[i][k])))
        idx2 = np.concatenate([np.concatenate([i * Lv + ar_v, B * Lv + j * X + ar_t]) for i, j in pairs])
        idx1_d = _dev_i32(idx1, dev) if "smtm" in cfg.get("pretrain_tasks", ()) else None
        km_txt = (mask != 0).to(torch.uint8)
        km1 = torch.cat([torch.ones(B, Lv, dtype=torch.uint8, device=dev), km_txt], 1).contiguous()
        tj_h = np.array([j for _, j in pairs], dtype=np.int64)
        tj = _h2d(torch.from_numpy(tj_h), dev)
        km2 = torch.cat([torch.ones(B * O, Lv, dtype=torch.uint8, device=dev), km_txt[tj]], 1).contiguous()
        if backward:                            # CSR of the pass-2 sequences by their text index (the pool gradient gathers through it)
            order = np.argsort(tj_h, kind="stable")
            csr = np.concatenate([np.concatenate([[0], np.cumsum(np.bincount(tj_h, minlength=B))]), order]).astype(np.int32)
            csr_d = _dev_i32(csr, dev)
            txt_off_d, txt_list_d = csr_d[:B + 1], csr_d[B + 1:]

        # ONE fusion pass over the B sequences of pass 1 (model.py:204-214 via main_pretrain.py:233) and the B*O sequences of the VTM
        # pass (:243-259) together: sequences are independent through the encoder, so every layer kernel runs once on (1 + O) * B
        # sequences instead of twice (the B-sequence launches filled 0.2 - 0.6 of a round of the persistent GEMM grids)
        n1, n2 = B, B * O
        idx12_d = _dev_i32(np.concatenate([idx1, idx2]), dev)
        km12 = torch.cat([km1, km2], 0).contiguous()
        # In the LAST layer only the text [CLS] row of the VTM sequences is alive (the VTM head reads nothing else, :260): it runs as
        # `_bert_layer_qrow` on n2 rows instead of n2 * Lq (`go_cross(qrow_split=...)`); results are those of the full layer.
        ntape = len(self.tape)
        qrow = os.environ.get("VMVM_QROW", "1") != "0"                             # (0: the whole last layer for every sequence, for A/B runs)
        if qrow:
            (out1, out2c), in12, _ = self.go_cross(pool, idx12_d, km12, n1 + n2, Lq, self._drop_on("fusion", train), qrow_split=(n1, Lv))
            if backward:
                out1.g = torch.empty_like(out1.t)                                  # the heads write it in place
        else:
            out12, in12, _ = self.go_cross(pool, idx12_d, km12, n1 + n2, Lq, self._drop_on("fusion", train))
            cls_rows = self._cached(("cls_rows", B * O, Lq, Lv), lambda: _dev_i32(np.arange(B * O) * Lq + Lv, dev))
            out1, out2c = V(out12.t[:n1 * Lq]), V(K.gather_rows(out12.t[n1 * Lq:], cls_rows, n2))
            if backward:
                out12.g = torch.empty_like(out12.t)
                out1.g = out12.g[:n1 * Lq]
        n_fusion_closures = len(self.tape) - ntape
        use_smtm = "smtm" in cfg.get("pretrain_tasks", ())
        if use_smtm:                            # third pass under the seq2seq mask (main_pretrain.py:238-240)
            out3, in3, _ = self.go_cross(pool, idx1_d, km1, B, Lq, self._drop_on("fusion", train), causal_from=Lv)
        losses = {k: torch.zeros(1, device=dev, dtype=F32) for k in ("mtm", "vtm", "mvm", "mvm_pixel", "mvm_vq", "mvm_feature", "mvm_hog", "smtm")}
        outs = {}

        # ---- MLM head (HF BertOnlyMLMHead; main_pretrain.py:236,560) -- also the head of the smtm pass (:240,:567)
        Vv = cfg["vocab"]
        txt_rows = self._cached(("txt_rows", B, Lv, X), lambda: _dev_i32(np.concatenate([i * Lq + Lv + ar_t for i in range(B)]), dev))
        tgt_m = ans_mtm.reshape(-1).contiguous()

        def mlm_head(outv, loss):
            return self._mlm_head_fwd(K.gather_rows(outv.t, txt_rows, B * X), B * X, tgt_m, loss, backward)

        mlm_head_bwd = self._mlm_head_bwd

        h_mlm = mlm_head(out1, losses["mtm"])
        if use_smtm:
            h_smtm = mlm_head(out3, losses["smtm"])
        if want_outputs:
            outs["out_mtm"] = h_mlm["logits"][:, :Vv].reshape(B, X, Vv)
            if use_smtm:
                outs["out_smtm"] = h_smtm["logits"][:, :Vv].reshape(B, X, Vv)

        # ---- VTM head (main_pretrain.py:146-147,260-262,561)
        r_v = out2c.t                                    # [B*O, H]: the text [CLS] states of the VTM sequences
        p_fc = 0.1 if self._drop_on("vtm", train) else 0.0
        off_fc = self._next_offset(r_v.numel())
        self.last_offsets["vtm"] = off_fc
        r_vd = K.dropout(r_v, p_fc, self.seed, off_fc) if p_fc > 0 else r_v
        h_v = K.gemm(r_vd, S.b("fc.1.weight"), bias=S.p("fc.1.bias"), act=2)
        inv_temp = 1.0 / cfg["temp"]
        lg_v = K.rowdot(h_v, S.p("fc.3.weight", (2 * Hd,)), S.p("fc.3.bias"), inv_temp)           # [B*O]
        tgt_v = torch.zeros(B, dtype=torch.int64, device=dev)
        K.cross_entropy(lg_v.view(B, O), O, tgt_v, losses["vtm"], want_grad=False)
        if want_outputs:
            outs["out_vtm"] = lg_v.view(B, O)
            outs["vtm_cls"] = r_v                       # the [CLS] states the VTM head reads (tests: head gradients on the same inputs)

        # ---- MVM pixel head (main_pretrain.py:178-179,420-432)
        ps = cfg["size_patch"]
        h_, w_ = H // ps, W // ps
        targets = cfg["mvm_target"]
        use_pix, use_vq, use_hog = "pixel" in targets, "vq" in targets, "hog" in targets
        vis_rows = self._cached(("vis_rows", B, T, hw, Lq), lambda: _dev_i32(
            np.concatenate([i * Lq + t * (1 + hw) + 1 + np.arange(hw) for i in range(B) for t in range(T)]), dev))
        if use_pix or use_hog:
            r_p = K.gather_rows(out1.t, vis_rows, B * T * hw)
        if use_hog:
            # MVM HOG head (main_pretrain.py:180-183,453-468): 1x1 conv H -> ps*ps + PixelShuffle(ps) -> one map per frame; L1 against
            # the data loader's HOG maps batch["hog"] (B,T,H,W) over pixels of covered patches, / (mask.sum() + 1e-5)
            Whog = S.b("decoder_hog.0.weight", (ps * ps, Hd))
            pred_h = K.gemm(r_p, Whog, bias=S.p("decoder_hog.0.bias"))
            msum_h = (cov.to(F32).sum() * float(ps * ps)).view(1)
            dpred_h = K.pixel_l1(pred_h, batch["hog"].to(F32).contiguous(), cov.reshape(-1), msum_h, losses["mvm_hog"], B, T, h_, w_, ps,
                                 channels=1, inv_div=1.0)
        if use_pix:
            Wpix = S.b("decoder_pixel.0.weight", (3 * ps * ps, Hd))
            pred = K.gemm(r_p, Wpix, bias=S.p("decoder_pixel.0.bias"))
            mask_sum = (cov.to(F32).sum() * float(3 * ps * ps)).view(1)
            dpred = K.pixel_l1(pred, img, cov.reshape(-1), mask_sum, losses["mvm_pixel"], B, T, h_, w_, ps)
            if want_outputs:
                outs["pred_pixel"] = pred
        # ---- MVM vq head (main_pretrain.py:194-209,469-502): frozen dVAE tokens as targets; decoder_vq (1x1 conv H -> 2H) +
        # PixelShuffle(4) + fc_mvm MLP + CE.  Only covered patches carry targets (ans = -1 elsewhere), so the head runs on the
        # covered patches' rows only.  PixelShuffle is folded into a row permutation of the decoder weight: output channel
        # c*16 + (i*4+j) moves to (i*4+j)*96 + c, so one GEMM row is 16 consecutive 96-channel positions.
        n_mp = 0
        if use_vq and "vq_patch_rows" in batch:
            prow, tix = batch["vq_patch_rows"], batch["vq_tok_index"]
            n_mp = int(prow.numel())
        if use_vq and n_mp > 0:
            up = ps // 8
            cq = 2 * Hd // (up * up)
            Vq = cfg.get("size_vq", 8192)
            tokens = batch.get("vq_tokens")
            if tokens is None:
                tokens = self.teacher.extract_vq_token(img.view(B * T, 3, H, W))
            tgt_q = tokens.reshape(-1)[tix].contiguous()
            perm = self._cached(("vq_perm", Hd, up), lambda: torch.from_numpy(
                (np.arange(cq)[None, :] * (up * up) + np.arange(up * up)[:, None]).reshape(-1).astype(np.int64)).to(dev))
            Wq = S.b("decoder_vq.0.weight", (2 * Hd, Hd)).index_select(0, perm)       # (tiny; plumbing)
            bq = S.p("decoder_vq.0.bias").index_select(0, perm)
            r_q = K.gather_rows(out1.t, prow, n_mp)
            y_q = K.gemm(r_q, Wq, bias=bq)                                            # [n_mp, 16*cq]
            x_q = y_q.view(n_mp * up * up, cq)
            p_q = 0.1 if self._drop_on("heads", train) else 0.0
            off_q = self._next_offset(x_q.numel())
            x_qd = K.dropout(x_q, p_q, self.seed, off_q) if p_q > 0 else x_q
            h_q = K.gemm(x_qd, S.b("fc_mvm.1.weight"), bias=S.p("fc_mvm.1.bias"), act=2)
            lg_q = K.gemm(h_q, S.b("fc_mvm.3.weight"), bias=S.p("fc_mvm.3.bias"), out_dtype=F32)
            dlg_q = K.cross_entropy(lg_q, Vq, tgt_q, losses["mvm_vq"], want_grad=backward, ld_d=Vq)
            if want_outputs:
                outs["vq_logits"], outs["vq_targets"] = lg_q, tgt_q
                outs["vq_acc"] = (lg_q.argmax(-1) == tgt_q).float().mean()
        # ---- MVM feature head (main_pretrain.py:153-174,508-545): fc_mvm (Dropout, Linear H -> 2H, ReLU, Linear 2H -> F) on every
        # non-cls visual token; targets = the frozen Swin teacher's features of the UN-masked clip; L1 over covered patches
        use_feat = "3d_feature" in targets or "2d_feature" in targets
        if use_feat:
            tgt_f = feat_target                                                          # bf16 [B*T*hw, F], no grad
            r_f = r_p if (use_pix or use_hog) else K.gather_rows(out1.t, vis_rows, B * T * hw)
            p_f = 0.1 if self._drop_on("heads", train) else 0.0
            off_f = self._next_offset(r_f.numel())
            r_fd = K.dropout(r_f, p_f, self.seed, off_f) if p_f > 0 else r_f
            h_f = K.gemm(r_fd, S.b("fc_mvm.1.weight"), bias=S.p("fc_mvm.1.bias"), act=2)
            pred_f = K.gemm(h_f, S.b("fc_mvm.3.weight"), bias=S.p("fc_mvm.3.bias"))
            cov_sum = cov.to(F32).sum().view(1)
            dpred_f = K.feature_l1(pred_f, tgt_f, cov.reshape(-1), cov_sum, losses["mvm_feature"])
            if want_outputs:
                outs["pred_feature"], outs["feature_target"] = pred_f, tgt_f
        losses["mvm"] = losses["mvm_pixel"] + losses["mvm_vq"] + losses["mvm_feature"] + losses["mvm_hog"]
        if want_outputs:
            outs["out_mvm"] = out1.t.view(B, Lq, Hd)[:, :Lv]
        if not backward:
            self.tape = []
            return losses, outs

        # =============================== backward ===============================
        # heads -> gradients of the two encoder outputs
        use_vis = use_pix or use_feat or use_hog
        npx = B * T * hw if use_vis else 0
        dcat = torch.empty((npx + B * X, Hd), device=dev, dtype=BF16)              # [visual-token rows ; mlm rows]
        vis_filled = False
        if use_pix:
            self._linear_bwd(dpred, r_p, None, None, w=Wpix, gw=S.g("decoder_pixel.0.weight", (3 * ps * ps, Hd)), gb=S.g("decoder_pixel.0.bias"),
                             dx_kw=dict(out=dcat[:npx]), wT=S.bt("decoder_pixel.0.weight"))
            vis_filled = True
        if use_hog:
            dr_h = self._linear_bwd(dpred_h, r_p, None, None, w=Whog, gw=S.g("decoder_hog.0.weight", (ps * ps, Hd)), gb=S.g("decoder_hog.0.bias"),
                                    dx_kw=None if vis_filled else dict(out=dcat[:npx]), wT=S.bt("decoder_hog.0.weight"))
            if vis_filled:
                K.add_bf16(dcat[:npx], dr_h, out=dcat[:npx])
            vis_filled = True
        if use_feat:
            dh_f = self._linear_bwd(dpred_f, h_f, "fc_mvm.3.weight", "fc_mvm.3.bias", dx_kw=dict(act=4, aux=h_f))   # ReLU' folded into the dgrad
            dr_f = self._linear_bwd(dh_f, r_fd, "fc_mvm.1.weight", "fc_mvm.1.bias")
            if p_f > 0:
                dr_f = K.dropout(dr_f, p_f, self.seed, off_f)
            if vis_filled:
                K.add_bf16(dcat[:npx], dr_f, out=dcat[:npx])
            else:
                dcat[:npx].copy_(dr_f)
        mlm_head_bwd(h_mlm, dcat[npx:])
        if use_smtm:
            d3 = torch.empty((B * X, Hd), device=dev, dtype=BF16)
            mlm_head_bwd(h_smtm, d3)
            inv3 = self._cached(("inv3", B, Lq, Lv, X), lambda: self._inverse_rows(B * Lq, [txt_rows]))
            out3.g = K.gather_rows(d3, inv3, B * Lq)
        inv1 = self._cached(("inv1", B, T, hw, Lq, Lv, X, use_vis), lambda: self._inverse_rows(B * Lq, [vis_rows, txt_rows] if use_vis else [txt_rows]))
        K.gather_rows(dcat, inv1, B * Lq, out=out1.g)
        if use_vq and n_mp > 0:
            dh_q = self._linear_bwd(dlg_q, h_q, "fc_mvm.3.weight", "fc_mvm.3.bias", dx_kw=dict(act=4, aux=h_q))     # ReLU' folded into the dgrad
            dx_q = self._linear_bwd(dh_q, x_qd, "fc_mvm.1.weight", "fc_mvm.1.bias")
            if p_q > 0:
                dx_q = K.dropout(dx_q, p_q, self.seed, off_q)
            gWq = torch.zeros((2 * Hd, Hd), device=dev, dtype=F32)
            gbq = torch.zeros(2 * Hd, device=dev, dtype=F32)
            dr_q = self._linear_bwd(dx_q.view(n_mp, 2 * Hd), r_q, None, None, w=Wq, gw=gWq, gb=gbq, wsync=True)    # (gWq / gbq are read right below)
            S.g("decoder_vq.0.weight", (2 * Hd, Hd)).index_add_(0, perm, gWq)          # undo the PixelShuffle row permutation
            S.g("decoder_vq.0.bias").index_add_(0, perm, gbq)
            out1.g.index_add_(0, prow.long(), dr_q)                                     # covered-patch rows (unique) of the fusion output
        # VTM
        # d(vtm)/d(logits) of the (B,O) matrix in f32: positives and negatives of a clip nearly cancel, a bf16-rounded softmax
        # would add rounding noise of the size of the signal (the reference's autocast runs cross_entropy in fp32 as well)
        dlg = torch.softmax(lg_v.view(B, O), 1)
        dlg[:, 0] -= 1.0
        dlg = (dlg / B).reshape(-1).contiguous()
        dh_v = K.rowdot_bwd(h_v, S.p("fc.3.weight", (2 * Hd,)), dlg, inv_temp, S.g("fc.3.weight", (2 * Hd,)), S.g("fc.3.bias"), relu_mask=True)
        dr_v = self._linear_bwd(dh_v, r_vd, "fc.1.weight", "fc.1.bias")
        if p_fc > 0:
            dr_v = K.dropout(dr_v, p_fc, self.seed, off_fc)
        out2c.g = dr_v
        if not qrow:
            inv2 = self._cached(("inv2", B * O, Lq, Lv), lambda: self._inverse_rows(B * O * Lq, [cls_rows]))
            K.gather_rows(dr_v, inv2, B * O * Lq, out=out12.g[n1 * Lq:])

        # encoders (tape holds: encode, the merged pass' layers (, the smtm pass' layers)) -> run them back, then gather into the pool
        n_layers = cfg["bert_layers"]
        for _ in range((n_layers if use_smtm else 0) + n_fusion_closures):
            self.tape.pop()()
        g12 = in12.g
        pool.g = K.pool_grad(g12[:n1 * Lq], g12[n1 * Lq:], in3.g if use_smtm else None, B, O, Lv, X, txt_off_d, txt_list_d)
        self.tape.pop()()                       # encode backward: text embeddings + EncVideo head -> last non-Swin gradients
        if on_other_grads_ready is not None:
            on_other_grads_ready()              # data-parallel: all-reduce of the non-Swin groups overlaps the Swin backward
        while self.tape:
            self.tape.pop()()
        self._wgrad_join()
        return losses, outs

    # -------------------------------------------------------------- downstream: text-to-video retrieval
    # -------------------------------------------------------------- scaffolding shared by the downstream passes
    def _begin_pass(self, img, txt, train, dp_all):
        """new tape, DropPath draw, encoders -> (pool, Lv, X, Lq): one token pool of B clips' visual rows, then the text sequences' rows"""
        self.tape = []
        if train and dp_all is None:
            dp_all = self.sample_drop_path(img.shape[0])
        pool, Lv, _ = self.encode(img, None, txt, dp_all, train)
        X = txt.shape[1]
        return pool, Lv, X, Lv + X

    def _fuse_pairs(self, pool, key, pairs, B, mask, Lv, X, train, cls_only):
        """One fusion pass over the (clip i, text sequence j) pairs: rows gathered from the token pool, key mask = ones over the visual part
        + the text's mask.  cls_only: the head reads the text [CLS] state alone, so the last layer runs for that query row only
        (`go_cross(qrow_split=(0, Lv))`).  -> dict(out = V of the full output or of the [CLS] rows, inn, idx, n_closures)."""
        dev = self.device
        ar_v, ar_t = np.arange(Lv), np.arange(X)
        idx_d = self._cached(key, lambda: _dev_i32(np.concatenate([np.concatenate([i * Lv + ar_v, B * Lv + j * X + ar_t]) for i, j in pairs]), dev))
        km_txt = (mask != 0).to(torch.uint8)
        tj = [j for _, j in pairs]
        if tj != list(range(km_txt.shape[0])):
            km_txt = km_txt[_h2d(torch.tensor(tj), dev)]
        km = torch.cat([torch.ones(len(pairs), Lv, dtype=torch.uint8, device=dev), km_txt], 1).contiguous()
        ntape = len(self.tape)
        if cls_only:
            (_, out), inn, _ = self.go_cross(pool, idx_d, km, len(pairs), Lv + X, train, qrow_split=(0, Lv))
        else:
            out, inn, _ = self.go_cross(pool, idx_d, km, len(pairs), Lv + X, train)
        return dict(out=out, inn=inn, idx=idx_d, n_closures=len(self.tape) - ntape)

    def _cls_hidden(self, r_v, train):
        """first half of the reference's `fc` heads on the [CLS] states (Dropout(0.1), Linear H -> 2H, ReLU; main_retrieval.py:54-56,
        main_qaoe.py:42-47) -> state for `_cls_hidden_bwd`"""
        S = self.store
        p_fc = 0.1 if train else 0.0
        off_fc = self._next_offset(r_v.numel())
        r_vd = K.dropout(r_v, p_fc, self.seed, off_fc) if p_fc > 0 else r_v
        return dict(h=K.gemm(r_vd, S.b("fc.1.weight"), bias=S.p("fc.1.bias"), act=2), x=r_vd, p=p_fc, off=off_fc)

    def _cls_hidden_bwd(self, st, dh):
        """dh = d(loss)/d(hidden) with the ReLU mask already applied -> d(loss)/d([CLS] states); accumulates fc.1's gradients"""
        dr = self._linear_bwd(dh, st["x"], "fc.1.weight", "fc.1.bias")
        return K.dropout(dr, st["p"], self.seed, st["off"]) if st["p"] > 0 else dr

    def _finish_pass(self, fz, pool, dout):
        """backward of the fusion pass (its closures are the newest on the tape), the token pool's gradient through the gather, the encoders"""
        fz["out"].g = dout
        for _ in range(fz["n_closures"]):
            self.tape.pop()()
        dpool = torch.zeros((pool.t.shape[0], pool.t.shape[1]), device=self.device, dtype=F32)
        K.scatter_add_rows(fz["inn"].g, fz["idx"], dpool)
        pool.g = K.cast_bf16(dpool)
        while self.tape:
            self.tape.pop()()
        self._wgrad_join()

    # -------------------------------------------------------------- downstream: text-video retrieval
    def retrieval_forward_backward(self, img, txt, mask, train=True, backward=True, dp_all=None, dlogits=None):
        """VIOLET_Retrieval.forward + NormSoftmaxLoss (main_retrieval.py:63-85, agent.py:34-50): every (video i, text j) pair of the
        batch goes through the fusion encoder (B*B sequences gathered from one token pool), the `fc` head reads the text [CLS]
        state, and the loss is the symmetric cross entropy of the (B,B) score matrix / temp with the diagonal as targets.
        Returns (loss f32[1], scores (B,B) f32).  `dlogits` (B,B) f32 replaces d(loss)/d(scores / temp) in the backward (tests:
        the loss gradient itself is a difference of nearly equal terms whenever the scores are close, a poor probe of the
        backward path)."""
        cfg, S, dev = self.cfg, self.store, self.device
        B, Hd = img.shape[0], cfg["hidden"]
        pool, Lv, X, Lq = self._begin_pass(img, txt, train, dp_all)
        fz = self._fuse_pairs(pool, ("ret_idx", B, Lv, X), [(i, j) for i in range(B) for j in range(B)], B, mask, Lv, X, train, cls_only=True)   # (:76 reads [CLS] only)
        st = self._cls_hidden(fz["out"].t, train)
        inv_temp = 1.0 / cfg["temp"]
        lg = K.rowdot(st["h"], S.p("fc.3.weight", (2 * Hd,)), S.p("fc.3.bias"), inv_temp).view(B, B)
        tgt = torch.arange(B, dtype=torch.int64, device=dev)
        loss = torch.zeros(1, device=dev, dtype=F32)
        K.cross_entropy(lg.contiguous(), B, tgt, loss, want_grad=False)                              # -mean diag log_softmax over rows
        K.cross_entropy(lg.t().contiguous(), B, tgt, loss, want_grad=False)                          # ... and over columns
        scores = lg * cfg["temp"]
        if not backward:
            self.tape = []
            return loss, scores
        # d(loss)/d(logits) of the (B,B) matrix in f32: the useful part of this gradient is what is left after the rows / columns
        # cancel, a bf16-rounded softmax would bury it (tiny matrix: plumbing)
        eye = torch.eye(B, device=dev, dtype=F32)
        dlg = ((torch.softmax(lg, 1) - eye) / B + (torch.softmax(lg, 0) - eye) / B).reshape(-1).contiguous()
        if dlogits is not None:
            dlg = dlogits.to(dev, F32).reshape(-1).contiguous()
        dh = K.rowdot_bwd(st["h"], S.p("fc.3.weight", (2 * Hd,)), dlg, inv_temp, S.g("fc.3.weight", (2 * Hd,)), S.g("fc.3.bias"), relu_mask=True)
        self._finish_pass(fz, pool, self._cls_hidden_bwd(st, dh))
        return loss, scores

    # -------------------------------------------------------------- downstream: open-ended video QA
    def qaoe_forward_backward(self, img, txt, mask, ans, train=True, backward=True, dp_all=None):
        """VIOLET_QAOE.forward + CrossEntropyLoss(ignore_index=-1) (main_qaoe.py:49-58,72-76, agent.py:57): one (video, question)
        fusion pass, `fc` (Dropout, Linear, ReLU, Linear -> answer vocabulary) on the text [CLS] state.  Returns (loss f32[1],
        logits (B, size_vocab) f32)."""
        cfg, S, dev = self.cfg, self.store, self.device
        B, Hd, NV = img.shape[0], cfg["hidden"], int(cfg["size_vocab"])
        pool, Lv, X, Lq = self._begin_pass(img, txt, train, dp_all)
        fz = self._fuse_pairs(pool, ("qa_idx", B, Lv, X), [(i, i) for i in range(B)], B, mask, Lv, X, train, cls_only=True)
        st = self._cls_hidden(fz["out"].t, train)
        h_v = st["h"]
        NVp = -(-NV // 8) * 8
        logits = torch.zeros((B, NVp), device=dev, dtype=F32)
        K.gemm(h_v, S.b("fc.3.weight"), N=-(-NV // 4) * 4, bias=S.p("fc.3.bias"), out=logits)
        loss = torch.zeros(1, device=dev, dtype=F32)
        dlog = K.cross_entropy(logits, NV, ans.to(dev).reshape(-1).contiguous(), loss, want_grad=backward, ld_d=NVp)
        if not backward:
            self.tape = []
            return loss, logits[:, :NV]
        K.colsum(dlog, S.g("fc.3.bias"), accumulate=True, M=B, N=NVp)
        K.gemm(dlog, h_v, a_kmajor=False, b_kmajor=False, M=NV, N=2 * Hd, K=B, out=S.g("fc.3.weight"), accumulate=True)
        dh = K.gemm(dlog, S.b("fc.3.weight"), b_kmajor=False, M=B, N=2 * Hd, K=NV, act=4, aux=h_v)            # ReLU' folded in
        self._finish_pass(fz, pool, self._cls_hidden_bwd(st, dh))
        return loss, logits[:, :NV]

    # -------------------------------------------------------------- downstream: multiple-choice video QA, MLM-head form
    def qamc_mlm_forward_backward(self, img, txt, mask, mask_ans, train=True, backward=True, dp_all=None):
        """VIOLET_QAMC_MLM_Head.forward + Agent_QAMC_MLM_Head.step (main_qamc_tsv_mlm_head.py:76-109): txt / mask / mask_ans are
        (B, O, X) -- one "question + option_o + [MASK]" sequence per option, labelled true / false token id at the [MASK] position
        and -1 elsewhere.  The clip's video tokens are shared by its O sequences (B*O sequences gathered from one token pool, as the
        retrieval head's pairs), the shared MLM head (`fc_mtm`) reads every text position, cross entropy with ignore_index -1.
        Returns (loss f32[1], logits (B*O*X, vocab) f32 view).  Task token / prompt (`enable_task_token`, `enable_prompt`) are off."""
        cfg, dev = self.cfg, self.device
        B, O, Vv = img.shape[0], int(txt.shape[1]), cfg["vocab"]
        n_seq = B * O
        txt2, mask2 = txt.reshape(n_seq, -1).contiguous(), mask.reshape(n_seq, -1).contiguous()
        pool, Lv, X, Lq = self._begin_pass(img, txt2, train, dp_all)          # pool rows: B*Lv visual, then (B*O)*X text
        fz = self._fuse_pairs(pool, ("qamc_idx", B, O, Lv, X), [(s_ // O, s_) for s_ in range(n_seq)], B, mask2, Lv, X, train, cls_only=False)
        txt_rows = self._cached(("qamc_txt_rows", n_seq, Lv, X), lambda: _dev_i32(np.concatenate([i * Lq + Lv + np.arange(X) for i in range(n_seq)]), dev))
        nr = n_seq * X
        loss = torch.zeros(1, device=dev, dtype=F32)
        hd = self._mlm_head_fwd(K.gather_rows(fz["out"].t, txt_rows, nr), nr, mask_ans.to(dev).reshape(-1).contiguous(), loss, backward)
        lg_ = hd["logits"]
        if not backward:
            self.tape = []
            return loss, lg_[:, :Vv]
        dtxt = self._mlm_head_bwd(hd)
        inv = self._cached(("qamc_inv", n_seq, Lq, Lv, X), lambda: self._inverse_rows(n_seq * Lq, [txt_rows]))
        self._finish_pass(fz, pool, K.gather_rows(dtxt, inv, n_seq * Lq))
        return loss, lg_[:, :Vv]

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
