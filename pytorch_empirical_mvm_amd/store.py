"""Parameter arena (`ParamStore`), activation / DropPath records and the small host helpers of the engine (split out of engine.py in
round 4; `engine.py` re-exports every name).  See engine.py for the schedule these serve."""
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

    W8_SCALE = 512.0        # static per-tensor scale of the e4m3 weight copy (|w| up to 0.875 before e4m3 saturates at 448)

    def refresh_shadow(self):
        self.sync_pending()
        K.cast_bf16(self.flat[:self.total], self.shadow[:self.total])
        self.refresh_transposed()
        self.refresh_fp8()

    # ---- e4m3 copy of the arena (BASELINE config 5's "fp8 MFMA path": opt-in forward GEMMs of the fusion qkv / FFN-in projections)
    def enable_fp8(self):
        self.sync_pending()
        if getattr(self, "shadow8", None) is None and self.device.type == "cuda":
            self.total8 = -(-self.total // 8) * 8
            self.shadow8 = torch.zeros(self.total8 + self.TAIL, device=self.device, dtype=torch.uint8)
            self.refresh_fp8()

    def refresh_fp8(self):
        """the e4m3 copy follows the bf16 compute copy (after every optimizer step / load)"""
        if getattr(self, "shadow8", None) is not None:
            K.cast_fp8(self.shadow[:self.total8], self.W8_SCALE, out=self.shadow8[:self.total8])

    def b8(self, n, shape=None):
        return self._view(self.shadow8, n, shape)

    def fused8(self, names, shape):
        return self.fused(self.shadow8, names, shape)
        self._shadow_version = self.flat._version        # (model._sync_param_surface: a torch-side in-place update of a parameter view bumps it)

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
    """activation + its gradient slot; `gdst` = where the FIRST gradient of this activation is to be written (a view of a larger
    buffer: the two halves of the last fusion layer's input write their rows side by side instead of being concatenated afterwards)"""
    __slots__ = ("t", "g", "gdst")

    def __init__(self, t, gdst=None):
        self.t, self.g, self.gdst = t, None, gdst


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
    if v.g is None:
        if v.gdst is not None and g.data_ptr() != v.gdst.data_ptr():
            v.gdst.copy_(g)
            g = v.gdst
        v.g = g
    else:
        v.g = K.add_bf16(v.g, g)


def _gout(v):
    """`out=` for the kernel that produces the first gradient of v (None: let it allocate)"""
    return v.gdst if v.g is None else None


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
