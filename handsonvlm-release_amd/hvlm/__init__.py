"""hvlm — MI355X-native HandsOnVLM hot path (host side).

Layout: `_lib` (ctypes binding of the C ABI), `ops` (tensor-level wrappers), and the mirror of the
reference's model interface (`model`, `clip`, `visual_to_tokens`, `traj_decoder`, `traj_eval`).
"""
__all__ = ["_lib"]
