#!/usr/bin/env python3
"""bench.py -- pretraining clips/sec of the VIOLETv2 step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W --batch B]         (N>1: launched by torch.distributed.run)

A "step" = one full optimizer step of the hot path over one synthetic WebVid-shape batch already resident in HBM:
rm/bm + MLM masking on the device (vmvm_masking, fresh draws each step; --host-masking moves it out of the timed region) ->
forward (Swin-B 8x224^2 + BERT embeddings + 12-layer fusion x (1+O) sequences) -> MLM/VTM/MVM-pixel losses -> backward ->
gradient all-reduce (N>1) -> clip -> AdamW; train mode (dropout + DropPath + attention dropout ON), bf16 compute.
Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed inside this run) and `cpu_baseline`
(the CPU oracle timed on this host on a bounded sample; rank 0, N=1 only)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_FLOP_PER_CLIP = 2.060e12        # 686.7 GFLOP fwd x 3 (C2/C3)
EXECUTED_FLOP_PER_CLIP = 2.060e12 - 0.068e12 - 0.09 * 3 * 281.3e9      # minus the VTM pass' dead query rows and ~9 % of the Swin blocks (DropPath draws of 0): 1.916e12
PEAK_BF16 = 2.5e15                    # dense MFMA bf16 (MI355X_MICROARCH.md)
PEAK_HBM = 8.0e12
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_roofline_gemm.txt")     # written by tools/pmc_roofline.sh (separate --pmc passes)


def pmc_traffic_bytes():
    """HBM-side bytes per launch of the roofline kernel from the COMMITTED counter passes (tools/pmc_roofline.sh -> profiles/):
    FETCH_SIZE x 2 (gfx950 counts a wide coalesced read at half its bytes, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, both in
    KiB.  None when the profile is not in the tree."""
    try:
        vals, in_gemm = {}, False
        for line in open(PMC_FILE):
            if not line.startswith(" "):
                in_gemm = "gemm_pp_kernel" in line or "gemm_pers_kernel" in line      # values follow their kernel's header line
                continue
            if in_gemm:
                for tok in line.split():
                    if tok.startswith("FETCH_SIZE=") or tok.startswith("WRITE_SIZE="):
                        k, v = tok.split("=")
                        vals.setdefault(k, float(v))
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
    except OSError:
        pass
    return None


def synth_batch(args, B, device, seed):
    """N(0,1) clipped frames, [CLS] ids [SEP] pad text, rm/bm masking p=0.15."""
    from pytorch_empirical_mvm_amd import config as CFG
    g = torch.Generator(device="cpu").manual_seed(seed)
    rng = np.random.RandomState(seed)
    T, S, X = args.size_frame, args.size_img, args.size_txt
    img = torch.randn(B, T, 3, S, S, generator=g).clamp_(-2.12, 2.64)
    txt = torch.zeros(B, X, dtype=torch.long)
    for b in range(B):
        n = rng.randint(5, 31) if X >= 32 else max(1, X - 2)
        n = min(n, X - 2)
        txt[b, 0] = CFG.TOKENS["cls"]
        txt[b, 1:1 + n] = torch.from_numpy(rng.randint(1000, CFG.BERT["vocab"], size=n))
        txt[b, 1 + n] = CFG.TOKENS["sep"]
    mask = (txt != 0).long()
    return img, txt, mask


class InStepTimers:
    """HIP events (on the launching stream) around the roofline kernels INSIDE a real training step: the fusion FFN fc1 GEMM
    (bias + GELU + saved 8-bit GELU' code; M = B*(1+O)*432 rows: the B sequences of pass 1 and the B*O of the VTM pass in one batch, N = 3072, K = 768) and the AdamW launches."""

    def __init__(self, M):
        self.M, self.gemm, self.adamw = M, [], []

    def __enter__(self):
        from pytorch_empirical_mvm_amd import kernels as K
        self.K, self.og, self.oa = K, K.gemm, K.adamw
        me = self

        def gemm(A, Bm, **kw):
            hit = kw.get("act", 0) == 1 and kw.get("out_preact") is not None and A.shape[0] == me.M and Bm.shape[0] == 3072 and A.shape[1] == 768
            if not hit:
                return me.og(A, Bm, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = me.og(A, Bm, **kw); e1.record()
            me.gemm.append((e0, e1))
            return out

        def adamw(p_, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = me.oa(p_, *a, **kw); e1.record()
            me.adamw.append((e0, e1, p_.numel(), torch.cuda.current_stream() == torch.cuda.default_stream()))
            return r
        K.gemm, K.adamw = gemm, adamw
        return self

    def __exit__(self, *exc):
        self.K.gemm, self.K.adamw = self.og, self.oa

    def results(self):
        torch.cuda.synchronize()
        g = [a.elapsed_time(b) * 1e-3 for a, b in self.gemm]
        # The AdamW launches of the Video-Swin groups run on the main stream and those of the other groups on the second stream AT THE SAME TIME
        # (agent.backward_step): per-launch durations would count the shared bandwidth twice over, so the time is the UNION of the launches'
        # intervals on the device timeline (events of both streams against the first launch's start event) and the bytes are all launches'.
        n = sum(c for _, _, c, _ in self.adamw)
        ad = 0.0
        if self.adamw:
            ref = self.adamw[0][0]
            iv = sorted((ref.elapsed_time(a), ref.elapsed_time(b)) for a, b, _, _ in self.adamw)
            cs, ce = iv[0]
            for s_, e_ in iv[1:]:
                if s_ > ce:
                    ad += ce - cs; cs, ce = s_, e_
                else:
                    ce = max(ce, e_)
            ad = (ad + ce - cs) * 1e-3
        self.adamw_launches = (sum(1 for x in self.adamw if x[3]), sum(1 for x in self.adamw if not x[3]))
        return (sum(g) / len(g) if g else None), len(g), ad, n


def cpu_baseline_worker(size, frames, threads, B=4, timed=3, warmup=1):
    """CPU oracle ('port' of the reference algorithm, validated against the reference's own outputs) on this host: full train
    steps (fwd + loss + bwd + clip + AdamW, fp32) on a BOUNDED sample of the same workload: B = 4 clips (so that every clip runs its
    1 + O = 5 fusion sequences as at B = 32), one warm-up step + `timed` timed steps, `threads` torch threads."""
    from oracle import violet_ref as R
    torch.set_num_threads(threads)
    cfg = R.make_cfg(size, T=frames, img=224, n_txt=32)
    torch.manual_seed(0)
    sd = {k: (torch.randn(s) * 0.02 if len(s) > 1 else (torch.ones(s) if "norm" in k.lower() and k.endswith("weight") else torch.zeros(s)))
          for k, s in R.param_shapes(cfg).items()}
    img = torch.randn(B, cfg["T"], 3, cfg["img"], cfg["img"]).clamp_(-2.12, 2.64)
    txt = torch.randint(1000, 30000, (B, cfg["n_txt"]))
    txt[:, 0] = 101
    mask = torch.ones_like(txt)
    mb = R.default_masking(cfg, img, txt, mask, seed=0)
    st, times = {}, []
    for i in range(warmup + timed):
        t0 = time.time()
        R.train_step(sd, cfg, mb, st, i + 1, 100, negatives=R.vtm_negatives_default(B))
        times.append(time.time() - t0)
    dt = sum(times[warmup:]) / timed
    wtxt = f"after {warmup} warm-up of {times[0]:.1f} s" if warmup else "no warm-up step"
    print(json.dumps(dict(value=round(B / dt, 5), unit="clips/s", cores=threads, kind="port", s_per_step=round(dt, 2),
                          sample=f"{timed} timed full fp32 train steps ({wtxt}) of oracle/violet_ref.py (fwd+loss+bwd+clip+AdamW), "
                                 f"Swin-{size} T={cfg['T']} 224^2, B={B} (5 fusion sequences per clip as in the GPU run), {threads} torch threads: "
                                 f"{dt:.1f} s/step")), flush=True)


def cpu_baseline(size, frames, timeout_s=600):
    """Runs the worker in CHILD processes (bounded by a timeout so the default bench run stays within minutes): the figure at up to 32
    threads (1 warm-up + 3 timed steps) and, beside it, one warmed-up timed step at 8 threads -- the thread count of the survey's probe of the
    reference itself (SURVEY 8d / BASELINE.md 3: 0.047 clips/s on 8 cores)."""
    import subprocess

    def run(threads, extra):
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--size", size, "--frames", str(frames),
                                "--threads", str(threads)] + extra, capture_output=True, text=True, timeout=timeout_s,
                               env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
            line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
            return json.loads(line)
        except Exception as e:
            return {"value": None, "unit": "clips/s", "cores": threads, "kind": "port", "sample": f"not completed within {timeout_s}s: {type(e).__name__}"}
    res = run(min(os.cpu_count() or 1, 32), [])
    if (os.cpu_count() or 1) > 8:
        r8 = run(8, ["--cpu-timed", "1", "--cpu-warmup", "1"])
        res["threads_8"] = {k: r8.get(k) for k in ("value", "unit", "cores", "s_per_step", "sample")}
    return res


def K_RELEASES():
    from pytorch_empirical_mvm_amd import kernels as K
    return K.RESERVE_RELEASES


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("VMVM_BENCH_BATCH", "32")), help="clips per GPU")
    ap.add_argument("--size", default="base")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--img", type=int, default=224, help="frame size; --size large --img 384 --frames 16 is BASELINE config 5's geometry (run at bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-masking", action="store_true", help="mask the batches on the host before the timed region (the round-1 default until device-side masking existed)")
    ap.add_argument("--mvm-target", default="pixel", help="pixel (C2/C3, the headline config), vq (C4: frozen dVAE tokenizer, random weights), 2d_feature (args_pretrain.json's own "
                    "default: frozen HF Swin-B teacher) or 3d_feature (frozen VideoSwin-B teacher)")
    ap.add_argument("--cpu-baseline-worker", action="store_true")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--cpu-timed", type=int, default=3)
    ap.add_argument("--cpu-warmup", type=int, default=1)
    ap.add_argument("--fp8", action="store_true", help="BASELINE config 5's fp8 MFMA path: fusion qkv / FFN-in forward GEMMs on e4m3 operands (opt-in; never the headline: C2 is bf16)")
    a = ap.parse_args()
    if a.cpu_baseline_worker:
        cpu_baseline_worker(a.size, a.frames, a.threads, timed=a.cpu_timed, warmup=a.cpu_warmup)
        return

    from pytorch_empirical_mvm_amd import config as CFG
    from pytorch_empirical_mvm_amd import dist as D
    from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
    from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain

    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != a.gpus:
        # inside a launcher whose world size disagrees with --gpus: every rank spawning its own nested launcher on the inherited
        # MASTER_PORT would clash and hang -- refuse instead
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']} (launch with --nproc-per-node {a.gpus})")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # not under torch.distributed.run: start it as a CHILD (nothing here has touched the GPU yet) and pass its exit code on
        import subprocess
        port = os.environ.get("MASTER_PORT", "29531")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", port,
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio into fd 1 (it came out AFTER the JSON line, at
    # exit, in the world-1 RCCL run of round 6): in a distributed run everything that is not the JSON line goes to stderr -- fd 1 is
    # pointed at fd 2 for the life of the process and the line is written to a duplicate of the original stdout.
    json_out = sys.stdout
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("VMVM_FORCE_DIST"):
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    rank, world, local = D.init_from_env("nccl")
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    args = CFG.get_args(vis_backbone_size=a.size, size_frame=a.frames, max_size_frame=max(a.frames, 6), size_img=a.img, size_txt=32,
                        mvm_target=[a.mvm_target], max_iter=10000, seed=88 + rank, fp8_forward=a.fp8)
    model = VIOLET_Pretrain(args, None, device=device)
    agent = Agent_Pretrain(args, model)
    agent.prepare_dist_model()
    agent.sched_step = 500                                   # mid-warm-up learning rate (non-trivial update)
    B = a.batch
    import random
    random.seed(88 + rank); np.random.seed(88 + rank); torch.manual_seed(88 + rank)
    batches, raw = [], []
    for i in range(2):
        img, txt, mask = synth_batch(args, B, device, 88 + rank + 1000 * i)
        raw.append((img.to(device), txt.to(device), mask.to(device)))
        mb = agent.masking(img, txt, mask, None)
        batches.append(agent.prepare_batch(mb))
    torch.cuda.synchronize()
    # masking INSIDE the timed step (device-side, vmvm_masking; fresh rm/bm draws every step); for the vq target that includes the
    # device-built index lists of the covered patches and their one count read-back
    mask_in_step = not a.host_masking
    gen = torch.Generator(device=device).manual_seed(88 + rank)


    def one_step(i):
        if mask_in_step:
            return agent.step(agent.masking_device(*raw[i % len(raw)], generator=gen), is_train=True, sync=False)
        return agent.step(batches[i % len(batches)], is_train=True, sync=False)

    for i in range(a.warmup):
        losses = one_step(i)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_first = None
    for i in range(a.steps):
        losses = one_step(i)
        if i == min(a.steps, 2) - 1:
            t_first = (time.perf_counter() - t0) / min(a.steps, 2)
    dt_host = time.perf_counter() - t0                 # the host has ISSUED the K steps (it runs ahead of the GPU; nothing in a step waits for the device)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    last = {k: float(v.item()) for k, v in losses.items()}
    # one more step with HIP events around the roofline kernels (outside the timed region: the events themselves cost nothing, the
    # step is the same work).  EVERY rank runs it: the step holds the gradient collectives.
    O = min(B, 4)
    Lq = a.frames * (1 + (a.img // 32) ** 2) + 32
    rccl = None
    if agent.comm is not None:
        agent.comm.timing = []
        c0, w0 = agent.comm.collectives, agent.comm.wire_bytes
    # (the timed steps issue a fusion layer / Swin block as ONE block-level foreign call; this step issues the same launches through the
    #  per-kernel entry points -- same kernels, same descriptors, bit-identical results: tests/test_round6_gpu.py::test_block_level_* --
    #  so that the events can sit around the one GEMM)
    blk = model.engine.sw.block_abi
    model.engine.sw.block_abi = False
    try:
        with InStepTimers(B * (1 + O) * Lq) as tm:
            one_step(a.steps)
    finally:
        model.engine.sw.block_abi = blk
    torch.cuda.synchronize()
    if agent.comm is not None:
        # the exchange of ONE step, for the first run on a real node to check itself against: group / backend as torch.distributed sees
        # them, channels and CUs left to the collectives, bytes this rank put on the wire, and how long the main stream sat waiting
        # for the side stream at the end of the backward (0 = the exchange hid under the backward completely)
        rccl = agent.comm.describe()
        rccl.update(collectives_per_step=agent.comm.collectives - c0, wire_bytes_per_step=agent.comm.wire_bytes - w0, cu_releases_by_event=K_RELEASES())
        agent.comm.timing = None
    if rank != 0:
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    clips = B * world * a.steps
    value = clips / dt
    kt, kcalls, adamw_s, adamw_n = tm.results()
    kflop = 2.0 * (B * (1 + O) * Lq) * 3072 * 768
    headline = a.size == "base" and a.frames == 8 and a.img == 224
    window = "(8,12,12)" if (a.size == "large" and a.img == 384) else "(8,7,7)"
    if headline:
        label = {"pixel": "C2", "vq": "C4 (1 GPU; frozen dVAE tokenizer on implicit-GEMM fp16 convolutions, random weights)"}.get(
            a.mvm_target, f"C2 shapes with the {a.mvm_target} target (frozen Swin-B teacher on the same kernels, random weights)")
    elif a.size == "large" and a.img == 384 and a.frames == 16:
        label = ("C5 geometry, fp8 (e4m3) forward GEMMs in the fusion encoder's qkv / FFN-in, everything else bf16" if a.fp8 else
                 "C5 geometry at bf16 (--fp8 switches the fusion qkv / FFN-in forward GEMMs to e4m3)") + "; streaming attention kernels"
    else:
        label = "non-headline shape"
    out = {
        "metric": "pretrain clips/sec (Swin-B, 8x224^2, 32 txt tok)" if headline else f"pretrain clips/sec (Swin-{a.size}, {a.frames}x{a.img}^2, 32 txt tok)", "value": round(value, 3), "unit": "clips/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "host_issue_ms_per_step": round(t_first * 1e3, 3), "host_loop_ms_per_step": round(dt_host / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16+fp8" if a.fp8 else "bf16", "data": "synthetic",
        "config": {"workload": f"{label}: VIOLETv2 pretrain step, Swin-{a.size} patch(2,4,4) window{window}, {a.frames}x{a.img}^2 frames, 32 text tokens, "
                               f"mvm_target={a.mvm_target}, MLM+VTM(O=4)+MVM, train mode (dropout/DropPath on), AdamW+clip, "
                               f"{'device-side rm/bm masking inside the timed step' if mask_in_step else 'masking before the timed region'}",
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}"},
        # two fractions of the dense bf16 MFMA peak: of the reference's ALGORITHMIC step FLOPs (2.060 TFLOP per clip -- the
        # metric's definition) and of the FLOPs this build EXECUTES (EXECUTED_FLOP_PER_CLIP: the dead query rows of the VTM pass' last fusion
        # layer, 0.068 TFLOP per clip, and on average ~9 % of the Video-Swin block FLOPs -- clip-branches whose DropPath draw is 0 -- are
        # never computed; same losses and gradients)
        "step_mfma_frac": round(value * TRAIN_FLOP_PER_CLIP / (world * PEAK_BF16), 4) if (a.mvm_target == "pixel" and headline) else None,
        "step_mfma_frac_executed": round(value * EXECUTED_FLOP_PER_CLIP / (world * PEAK_BF16), 4) if (a.mvm_target == "pixel" and headline) else None,
        "step_flop_note": "step_mfma_frac counts the reference's algorithmic 2.060 TFLOP/clip, step_mfma_frac_executed the 1.916 TFLOP/clip this build executes: not executed are "
                          "0.068 TFLOP/clip of dead query rows in the VTM pass' last fusion layer and, on average, ~9 % of the Video-Swin block FLOPs (clip-branches whose DropPath "
                          "draw is 0) -- results are those of the full computation; VMVM_QROW=0 VMVM_DROPPATH_DCE=0 executes everything as the reference formulates it "
                          "(round-5 build, one box, interleaved pairs: 106.41-106.43 ms against 102.72-102.77)",
        "host_note": "host_issue_ms_per_step = wall time the host needs to ISSUE one step while nothing holds it back (the first two timed steps: it is "
                     "then at most two steps ahead of the GPU); host_loop_ms_per_step = the same over all timed steps -- beyond ~2 steps of lead the launch "
                     "queue is full and the host waits in the launch call, so that figure follows the GPU step time and is not a cost",
        "switches": model.engine.sw.describe(),               # every VMVM_* step-path switch that differs from its default (empty: the measured-winner configuration)
        "roofline": {"bound": "mfma", "achieved": round(kflop / kt / 1e12, 1) if kt else None, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                     "frac": round(kflop / kt / PEAK_BF16, 4) if kt else None, "traffic": pmc_traffic_bytes(),
                     "kernel": f"fusion FFN fc1 GEMM + bias + GELU + saved 8-bit GELU' code (M={B * (1 + O) * Lq}, N=3072, K=768; 2*M*N*K flop per launch), "
                               f"gemm_pers_kernel<k-major,k-major,F=bias|GELU|rowscale> 128x128 persistent, two workgroups per CU, re-tiled stores; "
                               f"average of the {kcalls} launches of one real step, HIP events on the launching stream",
                     "algorithmic_bytes": int((B * (1 + O) * Lq) * 768 * 2 + 3072 * 768 * 2 + (B * (1 + O) * Lq) * 3072 * (2 + 1)),      # A + W read, bf16 output + 1-byte codes written
                     # the dominant memory-bound kernel of the step, against HBM: fused clip + AdamW over the flat arena
                     # (f32 p, g, m, v read + p, m, v written + bf16 copy written = 30 B per parameter)
                     "hbm": {"bound": "hbm", "kernel": f"adamw_kernel (clip coefficient + AdamW + bf16 copy over the parameter arena: {tm.adamw_launches[0]} launches on the main stream and "
                                                       f"{tm.adamw_launches[1]} on the second stream, running side by side; bytes of all of them over the union of their intervals on the device timeline)",
                             "achieved": round(30.0 * adamw_n / adamw_s / 1e9, 1) if adamw_s else None, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                             "frac": round(30.0 * adamw_n / adamw_s / PEAK_HBM, 4) if adamw_s else None, "algorithmic_bytes": int(30 * adamw_n)}},
        "rccl": rccl,                                 # None at world size 1 (no reducer is built: Agent_Pretrain.prepare_dist_model)
        "losses_last_step": last,
        "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),      # caching-allocator peak of this rank over the whole run
    }
    if world == 1 and not a.no_cpu_baseline and a.img == 224:
        out["cpu_baseline"] = cpu_baseline(a.size, a.frames)
    if kt is None:
        out["roofline"]["kernel"] += " (not launched at this configuration)"
    print(json.dumps(out), file=json_out, flush=True)
    if world > 1:
        torch.distributed.barrier()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
