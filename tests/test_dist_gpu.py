"""Two data-parallel ranks through the whole GPU step.  The box has one GPU, so both ranks share cuda:0 and the collectives go
through gloo (VMVM_DIST_BACKEND / VMVM_SHARE_GPU test hooks of dist.init_from_env): everything but RCCL itself runs -- rank-0
broadcast, the reductions hooked into the backward on the side stream, the 1/world scale, replica consistency -- and bench.py's
multi-rank contract (barrier, max over ranks, ONE JSON line from rank 0)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(__file__), "..")


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(script_args, timeout, **extra):
    env = dict(os.environ, VMVM_DIST_BACKEND="gloo", VMVM_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("wire,cus", [("f32", "0"), ("bf16", "16")], ids=["f32-wire", "bf16-wire+16-CUs-reserved"])
def test_two_ranks_gradient_mean_and_identical_replicas(wire, cus):
    """the default multi-GPU configuration -- bf16 gradient payload, persistent grids 16 CUs short while a collective is pending --
    and the full-precision one: reduced gradient = mean of the per-rank gradients, = the gradient of ONE process on the concatenated
    batch with the same (block-diagonal) negatives and equalised loss denominators, replicas bit-identical after 3 optimizer steps"""
    p = _run(["tools/dp_check.py"], 500, VMVM_GRAD_WIRE=wire, VMVM_COMM_CUS=cus, VMVM_COMM_CUS_ANY_BACKEND="1")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "replicas identical=True" in p.stdout and f"wire={wire} " in p.stdout and f"reserve_cus={cus} " in p.stdout, p.stdout[-2000:]
    assert "concat-batch rel err" in p.stdout                    # SURVEY 8 a17's pin: DP-2 == one process on the concatenated batch (same negatives)
    assert "autograd-step rel err" in p.stdout                   # round 6: model(batch) ... loss.backward() drives the same exchange phases


@pytest.mark.timeout(900)
def test_bench_contract_two_ranks():
    p = _run(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--no-cpu-baseline"], 800)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    o = json.loads(lines[0])
    assert o["n_gpus"] == 2 and o["steps"] == 2 and o["scaling"] == "weak" and o["config"]["global_batch"] == 8
    assert o["value"] > 0 and abs(o["value"] - 8 * 2 / (o["ms_per_step"] * 2 / 1e3)) < 1e-2 * o["value"]
    assert "cpu_baseline" not in o and o["roofline"]["frac"] > 0
    # round 6: the line describes its own exchange (what the first run on a real node checks itself against)
    r = o["rccl"]
    assert r["world"] == 2 and r["backend"] and r["wire"] in ("bf16", "f32") and r["collectives_per_step"] >= 3
    n_train = sum(e - a for a, e in r["phases"]["other_early"])            # (12 fusion layers: the early non-Swin phase exists)
    assert n_train > 0 and r["phases"]["mid_layer"] == 6 and r["phases"]["swin_tail"]
    assert r["wire_bytes_per_step"] > 0 and len(r["main_stream_wait_ms"]) == 1 and r["main_stream_wait_ms"][0] >= 0.0


@pytest.mark.timeout(900)
def test_zero1_shape_equals_plain_data_parallel():
    """VMVM_ZERO1=1 (the reference's default engine is DeepSpeed ZeRO-1, utils/deepspeed.py:42-44): per-shard reduce -> the owner's clip +
    AdamW on 1/world of the arena -> f32 master shards broadcast.  Same checks as the plain path (own-shard gradient = mean of the ranks',
    = one process on the concatenated batch, replicas bit-identical after 3 steps, 2 + 4 bytes per element per step on the wire) and the
    parameters after the 3 steps equal those of the plain data-parallel run (same sums, same update arithmetic)."""
    out = {}
    for z in ("0", "1"):
        p = _run(["tools/dp_check.py"], 400, VMVM_GRAD_WIRE="bf16", VMVM_COMM_CUS="0", VMVM_ZERO1=z)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        assert "replicas identical=True" in p.stdout and f"zero1={z}" in p.stdout, p.stdout[-2000:]
        out[z] = float([l for l in p.stdout.splitlines() if l.startswith("param-checksum")][0].split()[1])
        per = float(p.stdout.split("wire-bytes/element/step")[1].split()[0])
        assert abs(per - (6.0 if z == "1" else 2.0)) < 0.05, per
    assert abs(out["0"] - out["1"]) <= 1e-6 * abs(out["0"]), out
