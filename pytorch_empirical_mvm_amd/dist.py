"""Data-parallel plumbing (replaces utils/dist.py:20-75 + DDP/DeepSpeed, agent.py:195-201).

One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm) or "gloo" (CPU tests).
The only exchange of the step is the gradient sum: the flat gradient arena is reduced in two phases --
the non-Swin optimizer groups (fusion encoder, heads, embeddings: complete as soon as the fusion backward ends)
on a side stream while the Video-Swin backward still runs, then the Swin groups.  xGMI is point-to-point, so
a few LARGE messages (chunks of <= 256 MiB) are used instead of many small DDP-style buckets.

Payload (`VMVM_GRAD_WIRE`, default bf16): the f32 arena stays the local accumulator of the weight-gradient GEMMs; what goes on the
wire is a bf16 image of the finished segment (cast -> all-reduce -> cast back, all on the side stream): 450 MB per step instead
of 900 MB, ring lower bound 5.1 instead of 10.2 ms over xGMI.  The reference's DeepSpeed branch reduces fp16
gradients as well (utils/deepspeed.py:11-30).  Every rank receives the same reduced bits, so replicas stay bit-identical.
`VMVM_GRAD_WIRE=f32` keeps the full-precision exchange.

CU reservation (`VMVM_COMM_CUS`, default 16, 0 = off; nccl backend only): the persistent GEMM / LayerNorm-backward grids fill every
CU and split their tiles statically, so a collective's channel workgroups either wait for a kernel boundary or -- once resident --
push persistent workgroups into a second round.  While a collective is pending the kernels are launched `VMVM_COMM_CUS` CUs short
(vmvm_gemm_desc.reserve_cus) and RCCL is held to as many channels (NCCL_MAX_NCHANNELS, set before the process group is created,
unless the user already set it).

ZeRO-1 shape (`VMVM_ZERO1=1`, off by default; the reference's default engine is DeepSpeed ZeRO stage 1, utils/deepspeed.py:42-44 /
agent.py:196-199): the trainable arena is cut into `world` contiguous shards; every phase REDUCES each shard's piece to its owner
(same wire bytes as a reduce-scatter), the owner alone runs the clip + AdamW on its 1/world of the arena (Adam state touched: 1/world
-- the 6.75 GB per step optimizer stream of a GPU becomes 0.84 GB at 8 ranks), and the updated f32 master shards are broadcast
(= all-gather).  f32, not the bf16 compute copy: biases, LayerNorm weights, the embedding tables and the relative-position tables are
consumed in f32 by the kernels, so replicas stay bit-identical only if the masters are.  Wire per step: 2 B (gradients) + 4 B
(parameters) per element against 2 x 2 B of the bf16 all-reduce; the global gradient norm is one extra scalar all-reduce."""
import os

import torch
import torch.distributed as dist

from . import lib as L
from .switches import Switches

CHUNK_ELEMS = 64 * 1024 * 1024      # 256 MiB of f32 (128 MiB of bf16) per collective


def comm_cus():
    return Switches.from_env().comm_cus


def grad_wire():
    return Switches.from_env().grad_wire


def init_from_env(backend=None):
    """utils/dist.py:20-75 : env:// rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world < 2 and not os.environ.get("VMVM_FORCE_DIST"):
        return 0, 1, 0
    # dmabuf IPC (the pool's host driver supports nothing else: without it RCCL's first peer-to-peer set-up fails with
    # `hipIpcGetMemHandle: invalid argument`).  The launcher's environment already carries it; this only covers a bare environment, and
    # only helps when nothing has initialised the HSA runtime yet.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("VMVM_DIST_BACKEND"):            # test hook: e.g. gloo with several ranks sharing one GPU (see VMVM_SHARE_GPU)
        backend = os.environ["VMVM_DIST_BACKEND"]
    if os.environ.get("VMVM_SHARE_GPU") and torch.cuda.is_available():
        local = local % torch.cuda.device_count()      # N ranks on fewer GPUs: exercises the whole multi-rank path on a 1-GPU box (gloo)
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            if comm_cus() > 0:                       # one channel = one workgroup = one CU: keep RCCL inside the CUs the persistent grids leave free
                os.environ.setdefault("NCCL_MAX_NCHANNELS", str(comm_cus()))
            # RCCL's kernels run on the process group's own stream; high priority lets their workgroups take the first CUs that a
            # retiring GEMM workgroup frees (the persistent GEMM grids otherwise re-occupy every CU launch after launch)
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                kw["pg_options"] = opts
            except Exception:
                kw = {}
        dist.init_process_group(backend=backend, init_method="env://", **kw)
    return rank, world, local


def is_initialized():
    # VMVM_FORCE_DIST=1 keeps the collective path active at world size 1 (exercises RCCL + the side stream on one GPU)
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or bool(os.environ.get("VMVM_FORCE_DIST")))


def world_size():
    return dist.get_world_size() if is_initialized() else 1


def rank():
    return dist.get_rank() if is_initialized() else 0


def barrier():
    if is_initialized():
        dist.barrier()


def all_reduce_(t):
    if is_initialized():
        dist.all_reduce(t)
    return t


def broadcast_(t, src=0):
    if is_initialized():
        for a in range(0, t.numel(), CHUNK_ELEMS):
            dist.broadcast(t[a:a + CHUNK_ELEMS], src)
    return t


def zero1():
    return Switches.from_env().zero1


def shard_bounds(n, world, align=256):
    """`world` contiguous shards of [0, n), boundaries on multiples of `align` elements"""
    per = -(-(-(-n // world)) // align) * align
    return [(min(n, r * per), min(n, (r + 1) * per)) for r in range(world)]


def scatter_parts(a, e, world, align=256):
    """ZeRO-1 ownership of ONE reduction range [a, e) (round 5): `world` equal, `align`-ed parts -- part r = [a + r per, a + (r + 1) per)
    is what reduce_scatter_tensor hands to rank r and what it contributes to all_gather_into_tensor -- plus a tail of fewer than
    world * align elements owned by rank 0 (reduced with one small rooted reduce).  Returns (per, tail_start)."""
    per = ((e - a) // (world * align)) * align
    return per, a + world * per


def reduce_scatter_range_(buf, a, e, rank, world):
    """sum of buf[a:e] over the ranks, each rank left with ITS parts (scatter_parts) in place; returns the number of collectives.
    One reduce_scatter_tensor per CHUNK_ELEMS * world elements (in place: output = the rank's slice of the input, which is NCCL's /
    RCCL's in-place form) drives all links of every GPU, where `world` rooted reduces drive one destination's links each."""
    per, tail = scatter_parts(a, e, world)
    n = 0
    step = min(per, CHUNK_ELEMS) if per else 0
    if per:
        # the range is cut into `world` parts of `per`; a chunk takes the same [o, o + step) window of every part -> the windows are not
        # adjacent in memory unless step == per, so chunking re-packs nothing: it loops over windows of a strided view
        if step == per:
            dist.reduce_scatter_tensor(buf[a + rank * per:a + (rank + 1) * per], buf[a:a + world * per])
            n += 1
        else:
            view = buf[a:a + world * per].view(world, per)
            for o in range(0, per, step):
                w_ = min(step, per - o)
                inp = view[:, o:o + w_].contiguous()
                out = torch.empty(w_, dtype=buf.dtype, device=buf.device)
                dist.reduce_scatter_tensor(out, inp.view(-1))
                view[rank, o:o + w_].copy_(out)
                n += 1
    if e > tail:
        dist.reduce(buf[tail:e], 0)
        n += 1
    return n


def all_gather_range_(buf, a, e, rank, world):
    """every rank's parts of buf[a:e] (scatter_parts) to every rank, in place; returns the number of collectives."""
    per, tail = scatter_parts(a, e, world)
    n = 0
    if per:
        step = min(per, CHUNK_ELEMS)
        if step == per:
            dist.all_gather_into_tensor(buf[a:a + world * per], buf[a + rank * per:a + (rank + 1) * per])
            n += 1
        else:
            view = buf[a:a + world * per].view(world, per)
            for o in range(0, per, step):
                w_ = min(step, per - o)
                out = torch.empty(world * w_, dtype=buf.dtype, device=buf.device)
                dist.all_gather_into_tensor(out, view[rank, o:o + w_].contiguous())
                view[:, o:o + w_].copy_(out.view(world, w_))
                n += 1
    if e > tail:
        dist.broadcast(buf[tail:e], 0)
        n += 1
    return n


def reduce_chunks_(flat, a, e, dst):
    """sum-reduce flat[a:e] onto rank `dst` in CHUNK_ELEMS pieces; returns the number of collectives issued."""
    n = 0
    for s in range(a, e, CHUNK_ELEMS):
        dist.reduce(flat[s:min(e, s + CHUNK_ELEMS)], dst)
        n += 1
    return n


def all_reduce_chunks_(flat, a, e):
    """sum-all-reduce flat[a:e] in CHUNK_ELEMS pieces; returns the number of collectives issued."""
    n = 0
    for s in range(a, e, CHUNK_ELEMS):
        dist.all_reduce(flat[s:min(e, s + CHUNK_ELEMS)])
        n += 1
    return n


def _minus(rng, holes):
    """[a, e) without the (disjoint, sorted or not) ranges in `holes` that lie inside it -> list of ranges"""
    a, e = rng
    out, pos = [], a
    for lo, hi in sorted(holes):
        if hi <= a or lo >= e:
            continue
        if lo > pos:
            out.append((pos, lo))
        pos = max(pos, hi)
    if pos < e:
        out.append((pos, e))
    return out


class GradReducer:
    """Two-phase gradient all-reduce over the ParamStore arena (segments: 0 swin-decay, 1 other-decay, 2 swin-nodecay,
    3 other-nodecay).  Sums only -- the 1/world average is folded into the AdamW kernel's grad_scale."""

    def __init__(self, store, device, wire=None, reserve_cus=None):
        self.store = store
        self.cuda = torch.device(device).type == "cuda"
        self.stream = torch.cuda.Stream(device=device) if self.cuda else None
        self.pending = False
        self.tail_done = False
        self.sw = Switches.from_env()             # read once, at construction (the spawned test ranks set their environment first)
        self.wire = self.sw.grad_wire if wire is None else wire
        self.wire_buf = torch.empty_like(store.grad, dtype=torch.bfloat16) if self.wire == "bf16" else None       # bf16 image of the arena, same offsets
        nccl = dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
        # (VMVM_COMM_CUS_ANY_BACKEND: test hook -- the shared-GPU gloo test runs the short grids too)
        on = nccl or self.sw.comm_cus_any_backend
        self.reserve_cus = (self.sw.comm_cus if (on and self.cuda) else 0) if reserve_cus is None else int(reserve_cus)
        self.release_at_end = self.sw.comm_cus_release == "end"
        self.zero1 = self.sw.zero1 and dist.is_available() and dist.is_initialized()
        # ZeRO-1 (round 5): ownership follows the REDUCTION RANGES -- every range the three phases reduce (the two non-Swin groups; the
        # head and the tail of the two Swin groups) is cut into `world` equal parts, so each phase is ONE reduce_scatter_tensor per range
        # and the parameter exchange one all_gather_into_tensor per range (utils/deepspeed.py:42-44, ZeRO stage 1).  Round 4 cut the
        # whole arena into `world` contiguous shards and needed `world` rooted reduces + `world` broadcasts per phase.
        self.world = dist.get_world_size() if self.zero1 else 1
        self.rank = dist.get_rank() if self.zero1 else 0
        # Round 6: the non-Swin phase starts IN THE MIDDLE of the fusion backward.  When the backward has left fusion layer `mid` (layers run
        # last -> 0), the gradients of the heads (fc.*, fc_mtm.*, decoder_*, fc_mvm.*: written before the encoder's backward starts) and of
        # trsfr.layer.mid .. last are final: ~70 M of the 137 M non-Swin elements at C2 (two contiguous runs of the decay group: layers 6-11
        # and the heads) leave then and overlap layers 5-0, the rest (layers 0-5, embeddings, the no-decay group) after the encode backward.
        self.early = self._early_ranges()
        self.early_done = False
        self.ranges = self._reduction_ranges() if self.zero1 else [(0, store.n_trainable)]
        self.own = self.owned_ranges(self.rank) if self.zero1 else [(0, store.n_trainable)]
        self.wait_streams = []                    # further producer streams (the engine's weight-gradient stream) a reduction must wait for
        self.collectives = 0                      # issued so far (tests / profiling)
        self.wire_bytes = 0
        self.timing = None                        # list: reduce_swin_and_wait appends an event pair around the main stream's wait for the side stream

    def describe(self):
        """what the first run on a real multi-GPU node needs to verify itself (bench.py prints it as the line's "rccl" object)"""
        on = dist.is_available() and dist.is_initialized()
        waits = [a_.elapsed_time(b_) for a_, b_ in (self.timing or [])]
        return {"backend": dist.get_backend() if on else None, "world": dist.get_world_size() if on else 1, "wire": self.wire, "zero1": bool(self.zero1),
                "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"), "reserved_cus": self.reserve_cus, "cu_release": "end" if self.release_at_end else "event",
                "collectives_issued": self.collectives, "wire_bytes_issued": self.wire_bytes,
                "phases": {"other_early": [list(r) for r in self.early], "mid_layer": getattr(self, "mid_layer", None), "swin_tail": [list(r) for r in (self.store.swin_tail or [])]},
                "main_stream_wait_ms": [round(w, 3) for w in waits]}

    MIN_EARLY = 1 << 20                           # elements: shorter runs are not worth a collective of their own

    def _early_ranges(self):
        """runs of the non-Swin decay group (segment 1) whose gradients are final when the backward leaves fusion layer n // 2: the
        heads and trsfr.layer.l for l >= n // 2.  Empty for encoders of fewer than 4 layers (test configurations)."""
        S = self.store
        layers = sorted({int(n.split(".")[2]) for n in S.index if n.startswith("trsfr.layer.")})
        if len(layers) < 4:
            return []
        self.mid_layer = mid = layers[len(layers) // 2]
        a, e = S.segments[1]

        def final(n):
            if n.startswith("trsfr.layer."):
                return int(n.split(".")[2]) >= mid
            return not n.startswith(("enc_img.", "enc_txt.", "trsfr."))          # a head
        runs, cur = [], None
        for n, (o, c, _) in S.index.items():          # arena order
            if not (a <= o < e):
                continue
            end = o + -(-c // S.PAD) * S.PAD
            if final(n):
                cur = [o, end] if cur is None else [cur[0], end]
            elif cur is not None:
                runs.append(tuple(cur))
                cur = None
        if cur is not None:
            runs.append((cur[0], e))
        return [(lo, hi) for lo, hi in runs if hi - lo >= self.MIN_EARLY]

    def _reduction_ranges(self):
        """the disjoint ranges the phases reduce, in arena order (they cover [0, n_trainable))"""
        S = self.store
        tails = {e: a for a, e in (S.swin_tail or [])}
        out = []
        for gi in range(4):
            a, e = S.segments[gi]
            if e <= a:
                continue
            t = tails.get(e) if gi in (0, 2) else None
            if t is not None and a < t < e:
                out += [(a, t), (t, e)]
            elif gi == 1 and self.early:
                out += _minus((a, e), self.early) + list(self.early)
            else:
                out.append((a, e))
        return sorted(out)

    def owned_ranges(self, rank):
        """the parts of the arena rank `rank` owns under ZeRO-1 (reduced gradient, AdamW, master update), in arena order"""
        out = []
        for a, e in self.ranges:
            per, tail = scatter_parts(a, e, self.world)
            if per:
                out.append((a + rank * per, a + (rank + 1) * per))
            if e > tail and rank == 0:
                out.append((tail, e))
        return sorted(out)

    # ---- one contiguous range of the arena
    def _reduce_range(self, a, e):
        if self.zero1:                            # the range must be one of self.ranges (or a union of them): each goes to its owners by reduce-scatter
            for ra, re in self.ranges:
                if ra >= a and re <= e:
                    self._reduce_scatter(ra, re)
                else:
                    assert re <= a or ra >= e, ("ZeRO-1: a reduction cut through an ownership range", (a, e), (ra, re))
            return
        g = self.store.grad
        if self.wire_buf is None:
            self.collectives += all_reduce_chunks_(g, a, e)
            self.wire_bytes += 4 * (e - a)
            return
        w = self.wire_buf
        if self.cuda:
            from . import kernels as K            # HIP casts on the side stream (current stream inside the caller's stream context)
            K.cast_bf16(g[a:e], w[a:e])
            self.collectives += all_reduce_chunks_(w, a, e)
            K.cast_f32(w[a:e], g[a:e])
        else:                                     # CPU (gloo tests): same arithmetic through torch
            w[a:e].copy_(g[a:e])
            self.collectives += all_reduce_chunks_(w, a, e)
            g[a:e].copy_(w[a:e])
        self.wire_bytes += 2 * (e - a)

    def _reduce_scatter(self, a, e):
        """ZeRO-1: the sum of one ownership range, each rank left with its parts (f32 in the gradient arena)"""
        g, w = self.store.grad, self.wire_buf
        mine = [(lo, hi) for lo, hi in self.own if lo >= a and hi <= e]
        if w is None:
            self.collectives += reduce_scatter_range_(g, a, e, self.rank, self.world)
            self.wire_bytes += 4 * (e - a)
            return
        if self.cuda:
            from . import kernels as K
            K.cast_bf16(g[a:e], w[a:e])
            self.collectives += reduce_scatter_range_(w, a, e, self.rank, self.world)
            for lo, hi in mine:
                K.cast_f32(w[lo:hi], g[lo:hi])
        else:
            w[a:e].copy_(g[a:e])
            self.collectives += reduce_scatter_range_(w, a, e, self.rank, self.world)
            for lo, hi in mine:
                g[lo:hi].copy_(w[lo:hi])
        self.wire_bytes += 2 * (e - a)

    def gather_params(self, flat):
        """ZeRO-1: every owner contributes its updated f32 master parts (one all_gather_into_tensor per ownership range); returns the
        ranges the caller did NOT own (their bf16 compute copies have to be re-cast)."""
        for a, e in self.ranges:
            self.collectives += all_gather_range_(flat, a, e, self.rank, self.world)
            self.wire_bytes += 4 * (e - a)
        others, pos = [], 0
        for lo, hi in self.own + [(self.store.n_trainable, self.store.n_trainable)]:
            if lo > pos:
                others.append((pos, lo))
            pos = max(pos, hi)
        return others

    def _side(self, ranges):
        """run the ranges on the side stream (GPU) / inline (CPU); from here until the side stream has finished them (polled by
        kernels.reserve_cus(); at the latest reduce_swin_and_wait()) the persistent kernels are launched `reserve_cus` CUs short"""
        ranges = [(a, e) for a, e in ranges if e > a]
        if not ranges:
            return
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            for ws in self.wait_streams:
                self.stream.wait_stream(ws)
            with L.on_stream(self.stream):
                for a, e in ranges:
                    self._reduce_range(a, e)
            self.pending = True
            if self.reserve_cus > 0:
                from . import kernels as K
                ev = None
                if not self.release_at_end:       # VMVM_COMM_CUS_RELEASE=end: the reservation ends at a FIXED point (reduce_swin_and_wait), so the
                    ev = torch.cuda.Event()       # grids -- and the order of the LayerNorm / split-K partial sums -- do not depend on timing
                    ev.record(self.stream)        # behind the last collective queued so far: kernels.reserve_cus() polls it and gives the
                K.RESERVE_CUS, K.RESERVE_EVENT = self.reserve_cus, ev      # CUs back as soon as it has completed (not at the end of the backward)
        else:
            for a, e in ranges:
                self._reduce_range(a, e)

    def reduce_other_early(self):
        """called by the engine when the backward has left fusion layer n // 2 (engine_fusion.go_cross `mid_hook`): the heads and
        the upper half of the fusion layers leave now and overlap the lower half's backward"""
        if not is_initialized() or not self.early:
            return
        self._side(list(self.early))
        self.early_done = True

    def reduce_other(self):
        """called by the engine right after the last non-Swin gradient has been written: everything of the two non-Swin groups that
        `reduce_other_early` has not already sent"""
        if not is_initialized():
            return
        done = self.early if self.early_done else []
        self.early_done = False
        rest = []
        for gi in (1, 3):
            rest += _minus(self.store.segments[gi], done)
        self._side(rest)

    def reduce_swin_tail(self):
        """called by the engine when the backward leaves Swin stage n-2: stages >= n-2 (+ final norm) are final; their sum
        runs on the side stream under the two early, memory-bound stages"""
        if not is_initialized() or not self.store.swin_tail:
            return
        self._side(list(self.store.swin_tail))
        self.tail_done = True

    def reduce_swin_and_wait(self):
        if not is_initialized():
            return
        tails = {a: e for a, e in self.store.swin_tail} if getattr(self, "tail_done", False) else {}
        for gi in (0, 2):
            a, e = self.store.segments[gi]
            split = next((s for s, ee in tails.items() if ee == e and a <= s), e)      # tail of this segment already reduced?
            if split > a:
                self._reduce_range(a, split)      # nothing left to overlap with: on the main stream
        self.tail_done = False
        if self.cuda and self.pending:
            if self.timing is not None:           # (bench.py's instrumented step: how long did the main stream sit in this wait?)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                torch.cuda.current_stream().wait_stream(self.stream)
                e1.record()
                self.timing.append((e0, e1))
            else:
                torch.cuda.current_stream().wait_stream(self.stream)
            self.pending = False
        if self.cuda and self.reserve_cus > 0:
            from . import kernels as K
            K.RESERVE_CUS, K.RESERVE_EVENT = 0, None
