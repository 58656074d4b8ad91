"""Run-time switches of the step path, read from the environment ONCE (at engine / reducer construction) into one object.

Round 4 read `os.environ` inside per-block code (four look-ups per Video-Swin block per step, ~20 `VMVM_*` names scattered over five
modules); the host needs 60+ ms to issue a step, so none of that belongs on the step path.  Every switch is an A/B handle whose default
is the measured winner (README "Switches"); `describe()` goes into the bench line so a measurement names its configuration.
Tests flip a switch on a live engine by assigning the attribute (`engine.sw.droppath_dce = "0"`).
"""
import os
from dataclasses import dataclass, asdict, fields


def _flag(env, name, default):
    v = env.get(name)
    return default if v is None else v != "0"


@dataclass
class Switches:
    win_layout: bool = True        # VMVM_WIN_LAYOUT: region-major token order inside (8,7,7) windows + the win3 / win4 attention kernels
    ln_src_major: bool = True      # VMVM_LN_SRC_MAJOR: LayerNorm backward of the window-gathered norm1 (C <= 256) walks the source rows
    dx1_window: bool = True        # VMVM_DX1_WINDOW: norm2's backward writes d(x1) in the block's window order (no gather pass; blocks that run every clip)
    droppath_dce: str = "1"        # VMVM_DROPPATH_DCE: "1" both branches on the kept clips only, "attn" attention branch only, "0" every clip (scaled)
    qrow: bool = True              # VMVM_QROW: last fusion layer of the VTM sequences on the one query row the VTM head reads
    drop_mask: bool = True         # VMVM_DROP_MASK: fusion attention forward records its dropout decisions for the backward
    gelu_code8: bool = True        # VMVM_GELU_CODE8: GELU' saved as an 8-bit code
    wgrad_stream: bool = True      # VMVM_WGRAD_STREAM: weight gradients on a second HIP stream
    block_abi: bool = True         # VMVM_BLOCK_ABI: a fusion-encoder layer / Video-Swin block = ONE foreign call per direction (vmvm_bert_layer_*, vmvm_swin_block_*) instead of ~10 / ~22
    table_side: bool = False       # VMVM_TABLE_SIDE: a bias-table gradient that is a launch of its own (streaming windows, config 5) runs on the second stream (measured: config-5 step +1.4 .. +3 %, the chip is saturated either way -- opt-in)
    wgrad_hold: bool = True        # VMVM_WGRAD_HOLD: side-stream operands kept alive by a polled event FIFO (0: torch record_stream, rounds 4-5)
    opt_overlap: bool = True       # VMVM_OPT_OVERLAP: optimizer tail of the non-Swin groups beside the next forward
    zero1: bool = False            # VMVM_ZERO1: ZeRO-1 shape of the data-parallel step
    grad_wire: str = "bf16"        # VMVM_GRAD_WIRE: gradient payload on the wire (bf16 | f32)
    comm_cus: int = 16             # VMVM_COMM_CUS: CUs left to the collective while it is in flight
    comm_cus_any_backend: bool = False   # VMVM_COMM_CUS_ANY_BACKEND: test hook (short grids under gloo too)
    comm_cus_release: str = "event"      # VMVM_COMM_CUS_RELEASE: "event" (polled, fastest) | "end" (fixed point: bit-reproducible runs)

    @classmethod
    def from_env(cls, env=None):
        env = os.environ if env is None else env
        s = cls()
        s.win_layout = _flag(env, "VMVM_WIN_LAYOUT", True)
        s.ln_src_major = _flag(env, "VMVM_LN_SRC_MAJOR", True)
        s.dx1_window = _flag(env, "VMVM_DX1_WINDOW", True)
        s.droppath_dce = env.get("VMVM_DROPPATH_DCE", "1")
        s.qrow = _flag(env, "VMVM_QROW", True)
        s.drop_mask = _flag(env, "VMVM_DROP_MASK", True)
        s.gelu_code8 = _flag(env, "VMVM_GELU_CODE8", True)
        s.wgrad_stream = _flag(env, "VMVM_WGRAD_STREAM", True)
        s.wgrad_hold = _flag(env, "VMVM_WGRAD_HOLD", True)
        s.block_abi = _flag(env, "VMVM_BLOCK_ABI", True)
        s.table_side = _flag(env, "VMVM_TABLE_SIDE", False)
        s.opt_overlap = _flag(env, "VMVM_OPT_OVERLAP", True)
        s.zero1 = env.get("VMVM_ZERO1", "0") == "1"
        w = env.get("VMVM_GRAD_WIRE", "bf16").lower()
        if w not in ("bf16", "f32"):
            raise RuntimeError(f"VMVM_GRAD_WIRE={w!r}: expected bf16 or f32")
        s.grad_wire = w
        try:
            s.comm_cus = max(0, min(128, int(env.get("VMVM_COMM_CUS", "16"))))
        except ValueError:
            s.comm_cus = 16
        s.comm_cus_any_backend = bool(env.get("VMVM_COMM_CUS_ANY_BACKEND"))
        r = env.get("VMVM_COMM_CUS_RELEASE", "event").lower()
        if r not in ("event", "end"):
            raise RuntimeError(f"VMVM_COMM_CUS_RELEASE={r!r}: expected event or end")
        s.comm_cus_release = r
        return s

    def describe(self):
        """every switch that differs from its default (empty dict = the measured-winner configuration)"""
        d, base = asdict(self), asdict(Switches())
        return {k: v for k, v in d.items() if v != base[k]}

    def names(self):
        return [f.name for f in fields(self)]
