"""MI355X-native VIOLETv2 (EmpiricalMVM) pretraining step: hand-written gfx950 HIP kernels behind a C ABI
(libvmvm.so), driven from PyTorch-ROCm (device memory / streams / torch.distributed only)."""
__all__ = ["lib", "kernels", "swin_index"]
