"""The loop that drives the step: the reference's `Agent_Pretrain_YAML` (main_pretrain_yaml.py:82-214), its `MetaLoader`
(dataset.py:511-547) and `RunningMeter` (utils/logger.py:164-186), on top of this package's `Agent_Pretrain`.

Host-side control flow only -- which batch comes next, when the smoothed losses are logged, when the validation loaders run and a
checkpoint is written -- so that a user of the reference's `main_pretrain_yaml.py` finds the same cadence here:

    agent = Agent_Pretrain_YAML(args, model); agent.prepare_dist_model(); agent.save_training_meta()
    agent.run(MetaLoader({"webvid": (dl_webvid, 2), "cc3m": dl_cc3m}, distributed=world > 1), {"webvid": dl_val})

What is added to the reference's behaviour: every logging step also appends one JSON line (step, smoothed losses, clips per second over
the last logging interval, learning rates) to `<path_output>/train_log.jsonl` on rank 0 -- the machine-readable twin of the LOGGER text
(SURVEY section 5).  The reference's wandb calls have no counterpart (out of scope)."""
import json
import os
import random
import time
from collections import defaultdict

import numpy as np

from . import dist as D
from .agent import Agent_Pretrain


class RunningMeter:
    """utils/logger.py:164-186: exponentially smoothed scalar; the first value seeds it."""

    def __init__(self, name, val=None, smooth=0.99):
        self.name, self.val, self.smooth = name, val, smooth

    def __call__(self, value):
        self.val = value if self.val is None else value * (1.0 - self.smooth) + self.val * self.smooth

    def __str__(self):
        return f"{self.name}: {self.val:.4f}"


class MetaLoader:
    """dataset.py:511-547: an endless stream of (task name, batch) over several loaders.  `loaders` maps a name to a loader or to a
    (loader, ratio) pair; a task is drawn uniformly from the pool in which every name appears `ratio` times, once per `accum_steps`
    batches (gradient-accumulation groups stay on one task); an exhausted loader restarts.  With `distributed` rank 0's draw is
    broadcast, so all ranks step on the same task."""

    def __init__(self, loaders, accum_steps=1, distributed=False):
        if not isinstance(loaders, dict):
            raise TypeError("MetaLoader takes a dict name -> loader | (loader, ratio)")
        self.name2loader, self.name2iter, self.sampling_pools = {}, {}, []
        for name, entry in loaders.items():
            loader, ratio = entry if isinstance(entry, tuple) else (entry, 1)
            if not isinstance(ratio, int):
                raise TypeError(f"ratio of {name!r} must be an int")
            self.name2loader[name] = loader
            self.name2iter[name] = iter(loader)
            self.sampling_pools += [name] * ratio
        self.accum_steps, self.distributed, self.step = accum_steps, distributed, 0

    def _draw(self):
        task = random.choice(self.sampling_pools)
        if self.distributed and D.is_initialized():
            import torch.distributed as dist
            box = [task]
            dist.broadcast_object_list(box, src=0)
            task = box[0]
        return task

    def __iter__(self):
        task = self.sampling_pools[0]
        while True:
            if self.step % self.accum_steps == 0:
                task = self._draw()
            self.step += 1
            try:
                batch = next(self.name2iter[task])
            except StopIteration:
                self.name2iter[task] = iter(self.name2loader[task])
                batch = next(self.name2iter[task])
            yield task, batch


class Agent_Pretrain_YAML(Agent_Pretrain):
    """main_pretrain_yaml.py:82-194.  `run(dl_trs, dl_vls)`: a MetaLoader runs `run_meta_loader` (one stream of steps up to
    args.max_iter, evaluation + checkpoint every args.eval_step steps and once more at the end when the last step is not on the grid);
    a dict of per-dataset loaders runs `go_ep` for args.size_epoch epochs (args.iter_per_ep / args.eval_step are then dicts keyed by
    dataset)."""

    def __init__(self, args, model):
        super().__init__(args, model)
        self.task2loss = {}
        self.log = defaultdict(list)
        self.ds_tr_steps = defaultdict(int)
        self._tp_t, self._tp_clips = None, 0                 # throughput bookkeeping of the JSONL logger

    # ---- meters / logging
    def meter_loss(self, dataset, ls):
        for key, val in ls.items():
            name = f"{dataset}_ls_{key}"
            if name not in self.task2loss:
                self.task2loss[name] = RunningMeter(name)
            self.task2loss[name](val)

    def log_memory(self):
        lrs = self.current_lrs()
        mem = 0
        try:
            import torch
            mem = torch.cuda.max_memory_allocated() if torch.cuda.is_available() else 0
        except Exception:
            pass
        return f"global step: {self.global_step}, lr_swin: {lrs[0]:.2e}, lr_bert: {lrs[1]:.2e}, max memory: {mem / 2 ** 30:.2f} GiB"

    def log_train(self):
        info = self.log_memory() + "\n\t"
        for task, rm in self.task2loss.items():
            info += f" {task}: {rm.val:.6f}" if rm.val != -1 else f" {task}: -1"
        self._jsonl()
        return info

    def _jsonl(self):
        """one machine-readable line per logging step (rank 0): smoothed losses and the clips per second since the previous line"""
        now = time.time()
        rate = None
        if self._tp_t is not None and now > self._tp_t and self._tp_clips:
            rate = self._tp_clips * self.world_size / (now - self._tp_t)
        self._tp_t, self._tp_clips = now, 0
        if self.rank != 0 or not getattr(self.args, "path_output", None):
            return
        os.makedirs(self.args.path_output, exist_ok=True)
        lrs = self.current_lrs()
        rec = dict(step=self.global_step, clips_per_s=None if rate is None else round(rate, 2), lr_swin=lrs[0], lr_other=lrs[1],
                   **{k: (None if m.val is None else round(float(m.val), 6)) for k, m in self.task2loss.items()})
        with open(os.path.join(self.args.path_output, "train_log.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")

    def _say(self, msg):
        if self.rank == 0:
            print(msg, flush=True)

    # ---- one training batch: mask -> device -> step (main_pretrain_yaml.py:134-139,173-178)
    def _train_batch(self, ds_key, batch):
        batch = defaultdict(lambda: None, batch)
        masked = self.masking(batch["img"], batch["txt"], batch["mask"], batch["vq"])
        batch.update(masked)
        ls = self.step(self.prepare_batch(batch), is_train=True)
        if batch["txt"] is not None and hasattr(batch["txt"], "shape"):
            self._tp_clips += int(batch["txt"].shape[0])
        return ls

    def _validate(self, dl_vls, head, suffix):
        for ds_vl_key, dl_vl in dl_vls.items():
            res = self.evaluate(dl_vl)
            for k, v in res.items():
                self.log[f"{ds_vl_key}_{suffix}_{k}"].append(v)
            self._say(f"{head} {ds_vl_key} vl: {json.dumps(res)}")

    # ---- epoch form (main_pretrain_yaml.py:123-163)
    def go_ep(self, dl_trs, dl_vls, ep):
        for ds_tr_key, dl_tr in dl_trs.items():
            iter_per_ep, eval_step = self.args.iter_per_ep[ds_tr_key], self.args.eval_step[ds_tr_key]
            step = 0
            if hasattr(dl_tr, "__dict__"):
                dl_tr.start_iter = (ep - 1) * iter_per_ep
            for batch in dl_tr:
                if step % self.args.logging_steps == 0:
                    self._say(f"Train dataset {ds_tr_key}: {self.log_train()}")
                ls = self._train_batch(ds_tr_key, batch)
                self.meter_loss(ds_tr_key, ls)
                step += 1
                self.global_step += 1          # (as the reference: step() counts too, main_pretrain.py:595 -- two per batch; it only labels log lines)
                if step % eval_step == 0 and step:
                    self._validate(dl_vls, f"Train dataset {ds_tr_key}, ep {ep}, step {step},", "vl")
                    self.save_model(ep, ds_tr_key, step)
                if step >= iter_per_ep:
                    break
            if step % self.args.logging_steps != 0:
                self._say(f"Train dataset {ds_tr_key}:" + self.log_train())
            if step % eval_step != 0:
                self._validate(dl_vls, f"Train dataset {ds_tr_key},Ep {ep}, step {step},", "acc")
                self.save_model(ep, ds_tr_key, step)

    # ---- stream form (main_pretrain_yaml.py:165-194)
    def run_meta_loader(self, dl_trs, dl_vls):
        self._say("Start training....")
        step, ep = 0, 0
        for step, (ds_tr_key, batch) in enumerate(dl_trs):
            ep = step // self.args.iter_per_ep
            self.ds_tr_steps[ds_tr_key] += 1
            if step % self.args.logging_steps == 0:
                self._say(self.log_train() + f"\n\t\t {dict(self.ds_tr_steps)}")
            ls = self._train_batch(ds_tr_key, batch)
            self.global_step += 1
            self.meter_loss(ds_tr_key, ls)
            if step % self.args.eval_step == 0 and step:
                self._validate(dl_vls, f"Ep {ep + 1}, step {step},", "vl")
                self.save_model(ep + 1, "", step)
            if step >= self.args.max_iter:
                break
        if step % self.args.logging_steps == 0:
            self._say(self.log_train() + f"\n\t\t {dict(self.ds_tr_steps)}")
        if step % self.args.eval_step != 0 and step:
            self._validate(dl_vls, f"Ep {ep}, step {step},", "acc")
            self.save_model(ep + 1, "", step)

    def run(self, dl_trs, dl_vl):
        if isinstance(dl_trs, MetaLoader):
            self.run_meta_loader(dl_trs, dl_vl)
        else:
            self._say("Start training....")
            for ep in range(self.args.size_epoch):
                self.go_ep(dl_trs, dl_vl, ep + 1)

    def evaluate(self, dl):
        """main_pretrain_yaml.py:196-214 (= Agent_Pretrain.evaluate: eval mode, masked like training, NaN-ignoring rank-averaged means)"""
        return super().evaluate(dl)


_ = np  # (numpy is part of the reference module's surface; kept for callers that monkeypatch np.random here)
