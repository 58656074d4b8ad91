"""The loop that drives the step (pytorch_empirical_mvm_amd/loop.py) against tests/golden/loop.json, recorded from the reference's own
`MetaLoader` (dataset.py:511-547), `RunningMeter` (utils/logger.py:164-186) and `Agent_Pretrain_YAML.run_meta_loader` / `go_ep`
(main_pretrain_yaml.py:123-194) by tools/gen_goldens.py::gold_loop with the step / evaluate / save hooks replaced by recorders: the same
(task, batch) order for the same `random` seed, the same step / evaluation / checkpoint cadence, the same smoothed losses, log dict
and step counters.  Host logic only -- no GPU, no HIP library."""
import json
import os
import random
from collections import defaultdict

import torch

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loop.json")))


class _Args(dict):
    __getattr__ = dict.__getitem__


def _recorder(args):
    from pytorch_empirical_mvm_amd.loop import Agent_Pretrain_YAML

    class Rec(Agent_Pretrain_YAML):
        def __init__(self, args):                           # (no model / engine: control flow only)
            self.args, self.trace, self.n = args, [], 0
            self.task2loss, self.log, self.ds_tr_steps, self.global_step = {}, defaultdict(list), defaultdict(int), 0
            self.rank, self.world_size, self._tp_t, self._tp_clips = 0, 1, None, 0

        def masking(self, img, txt, mask, vq): return {"masked": True}
        def prepare_batch(self, b): return b
        def current_lrs(self): return [1e-5, 1e-5, 1e-5, 1e-5]

        def step(self, batch, is_train):
            self.n += 1
            self.trace.append(["step", batch["id"]])
            return {"mtm": 1.0 + 0.5 * self.n, "vtm": 0.25 * self.n, "mvm": -1}

        def evaluate(self, dl):
            self.trace.append(["eval", dl])
            return {"mtm": 0.5, "vtm": 0.75}

        def save_model(self, ep, ds, step): self.trace.append(["save", ep, ds, step])
    return Rec(args)


def _dump(a):
    return dict(trace=a.trace, meters={k: v.val for k, v in a.task2loss.items()}, log={k: v for k, v in a.log.items()},
                ds_tr_steps=dict(a.ds_tr_steps), global_step=a.global_step)


def test_meta_loader_order_matches_the_reference():
    from pytorch_empirical_mvm_amd.loop import MetaLoader
    mk = lambda n, tag: torch.utils.data.DataLoader([f"{tag}{i}" for i in range(n)], batch_size=1, collate_fn=lambda x: x[0])
    for acc in (1, 2):
        random.seed(5)
        it = iter(MetaLoader({"a": (mk(3, "a"), 2), "b": mk(2, "b")}, accum_steps=acc))
        assert [list(next(it)) for _ in range(14)] == GOLD[f"meta_accum{acc}"]


def test_running_meter_matches_the_reference():
    from pytorch_empirical_mvm_amd.loop import RunningMeter
    rm, vals = RunningMeter("x"), []
    for v in (2.0, 1.0, 4.0, -1.0):
        rm(v); vals.append(rm.val)
    assert vals == GOLD["running_meter"]


def test_run_meta_loader_cadence_matches_the_reference(tmp_path):
    for max_iter, eval_step in ((7, 3), (6, 3), (4, 10)):
        a = _recorder(_Args(iter_per_ep=4, logging_steps=2, eval_step=eval_step, max_iter=max_iter, path_output=str(tmp_path / f"o{max_iter}")))
        stream = [("ds%d" % (i % 2), {"id": i, "img": None, "txt": None, "mask": None, "vq": None}) for i in range(50)]
        a.run_meta_loader(stream, {"val": "VL"})
        assert _dump(a) == GOLD[f"run_meta_loader_{max_iter}_{eval_step}"], (max_iter, eval_step)
        lines = [json.loads(l) for l in open(tmp_path / f"o{max_iter}" / "train_log.jsonl")]       # the added JSONL log: one line per logging step
        assert len(lines) >= max_iter // 2 and all("step" in l and "clips_per_s" in l for l in lines)


def test_go_ep_cadence_matches_the_reference():
    for iter_per_ep, eval_step in ((5, 2), (4, 2)):
        a = _recorder(_Args(iter_per_ep={"d": iter_per_ep}, logging_steps=2, eval_step={"d": eval_step}, path_output=None))

        class DL(list):
            pass
        a.go_ep({"d": DL({"id": i, "img": None, "txt": None, "mask": None, "vq": None, "vid": ["v"]} for i in range(20))}, {"val": "VL"}, 2)
        assert _dump(a) == GOLD[f"go_ep_{iter_per_ep}_{eval_step}"], (iter_per_ep, eval_step)
