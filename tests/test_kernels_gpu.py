"""Every libvmvm kernel (through the C ABI) against a plain PyTorch fp32 reference of the same op, on the GPU.
The checks live in tools/gpu_check.py (also used as a development harness); tolerance 2e-2 of the tensor scale for
bf16 outputs, 1e-3..1e-5 for f32 outputs, exact for pure data movement (see each `rep(..., tol=)`)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location("gpu_check", os.path.join(os.path.dirname(__file__), "..", "tools", "gpu_check.py"))


@pytest.fixture(scope="module")
def gc():
    import torch
    m = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(m)
    torch.manual_seed(0)
    return m


@pytest.mark.parametrize("group", ["check_probe", "check_gemm_layouts", "check_gemm_colsum", "check_gemm_fp16_conv", "check_dvae_passes", "check_pool_grad", "check_gemm_fp8", "check_gemm_big", "check_gemm_p3", "check_gemm_epilogues", "check_gemm_round_split", "check_ln", "check_ln_gather",
                                   "check_attn_window", "check_attn_window_spike", "check_attn_window_mask_boundary", "check_attn_window_nonfinite", "check_attn_bert", "check_attn_query_row", "check_attn_stream", "check_attn_seq2seq", "check_attn_colsum", "check_misc"])
def test_kernel_group(gc, group):
    import torch
    gc.RESULTS.clear()
    getattr(gc, group)()
    torch.cuda.synchronize()
    bad = [r for r in gc.RESULTS if r[3]]
    assert gc.RESULTS and not bad, bad
