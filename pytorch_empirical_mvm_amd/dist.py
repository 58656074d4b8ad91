"""Data-parallel plumbing (replaces utils/dist.py:20-75 + DDP/DeepSpeed, agent.py:195-201).

One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm) or "gloo" (CPU tests).
The only exchange of the step is the gradient sum: the flat f32 gradient arena is reduced in two phases --
the non-Swin optimizer groups (fusion encoder, heads, embeddings: complete as soon as the fusion backward ends)
on a side stream while the Video-Swin backward still runs, then the Swin groups.  xGMI is point-to-point, so
a few LARGE messages (chunks of <= 256 MiB) are used instead of many small DDP-style buckets."""
import os

import torch
import torch.distributed as dist

CHUNK_ELEMS = 64 * 1024 * 1024      # 256 MiB of f32 per collective


def init_from_env(backend=None):
    """utils/dist.py:20-75 : env:// rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world < 2 and not os.environ.get("VMVM_FORCE_DIST"):
        return 0, 1, 0
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


def all_reduce_chunks_(flat, a, e):
    """sum-all-reduce flat[a:e] in CHUNK_ELEMS pieces; returns the number of collectives issued."""
    n = 0
    for s in range(a, e, CHUNK_ELEMS):
        dist.all_reduce(flat[s:min(e, s + CHUNK_ELEMS)])
        n += 1
    return n


class GradReducer:
    """Two-phase gradient all-reduce over the ParamStore arena (segments: 0 swin-decay, 1 other-decay, 2 swin-nodecay,
    3 other-nodecay).  Sums only -- the 1/world average is folded into the AdamW kernel's grad_scale."""

    def __init__(self, store, device):
        self.store = store
        self.cuda = torch.device(device).type == "cuda"
        self.stream = torch.cuda.Stream(device=device) if self.cuda else None
        self.pending = False
        self.tail_done = False

    def _run(self, segs):
        g = self.store.grad
        for gi in segs:
            a, e = self.store.segments[gi]
            if e > a:
                all_reduce_chunks_(g, a, e)

    def reduce_other(self):
        """called by the engine right after the last non-Swin gradient has been written"""
        if not is_initialized():
            return
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self._run((1, 3))
            self.pending = True
        else:
            self._run((1, 3))

    def reduce_swin_tail(self):
        """called by the engine when the backward leaves Swin stage n-2: stages >= n-2 (+ final norm) are final; their sum
        runs on the side stream under the two early, memory-bound stages"""
        if not is_initialized() or not self.store.swin_tail:
            return
        g = self.store.grad
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for a, e in self.store.swin_tail:
                    all_reduce_chunks_(g, a, e)
            self.pending = True
        else:
            for a, e in self.store.swin_tail:
                all_reduce_chunks_(g, a, e)
        self.tail_done = True

    def reduce_swin_and_wait(self):
        if not is_initialized():
            return
        g = self.store.grad
        tails = {a: e for a, e in self.store.swin_tail} if getattr(self, "tail_done", False) else {}
        for gi in (0, 2):
            a, e = self.store.segments[gi]
            split = next((s for s, ee in tails.items() if ee == e and a <= s), e)      # tail of this segment already reduced?
            if split > a:
                all_reduce_chunks_(g, a, split)
        self.tail_done = False
        if self.cuda and self.pending:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.pending = False
