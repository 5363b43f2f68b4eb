"""prego_amd: MI355X-native step_recognition hot path of PREGO."""
import os as _os

# dmabuf IPC for multi-process GPU work (RCCL, CUDA-tensor sharing) on this pool.  ROCr reads the variable at hsa_init,
# i.e. at the first HIP call of the process, so it has to be in the environment before anything touches the GPU: set it at
# package import, which precedes every prego_amd entry point (main.py, bench.py, the tests).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__all__ = ["config", "weights", "registry"]
