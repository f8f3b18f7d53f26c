"""humanoid_mujoco_amd — MI355X-native batched humanoid physics-step / rollout engine.

The package holds only what the hot path needs: ``csrc/`` (HIP kernels + the C-ABI of
``libhb.so``, declared in ``include/hb.h``) and the ctypes host binding in ``engine.py``
plus the VecEnv-shaped adapter in ``vecenv.py``.
"""
from .engine import (Batch, HbError, Model, lib, LIB_PATH, STATE_INTEGRATION, STATE_PHYSICS, STATE_QPOS, STATE_QVEL,  # noqa: F401
                     STATE_TIME, STATE_WARMSTART, STATE_XFRC_APPLIED, WARN_BADQACC, WARN_BADQPOS, WARN_BADQVEL,
                     WARN_CNSTRFULL, WARN_CONTACTFULL)

from .vecenv import VecEnv  # noqa: E402,F401

__all__ = ["Batch", "Model", "VecEnv", "HbError", "lib", "LIB_PATH"]
