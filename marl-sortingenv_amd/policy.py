"""Device forward of the reference's actor-critic policy.

The reference trains `MaskablePPO(MaskableActorCriticPolicy, net_arch=dict(pi=[32, 32], vf=[32, 32]))`
(src/training.py:115-131): two separate 2x32 tanh MLPs, a linear action head and a linear value head; invalid
actions get logit -1e8 before the softmax.  `MlpPolicy.forward` evaluates that network for a whole batch of
observations on the f32 matrix cores (`mse_policy_forward`, csrc/mse_policy.hip) and samples actions with the
engine's counter-based stream, so `env.step(policy.forward(obs, mask)["action"])` never leaves the device.
PyTorch is plumbing here (device memory, stream); no torch op computes anything on this path.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional

import numpy as np
import torch

from ._lib import check, load_library

HIDDEN = 32
# torch.nn.Linear tensors in the order include/mse.h documents, under the names SB3's state_dict gives them
SB3_KEYS = [
    "mlp_extractor.policy_net.0.weight", "mlp_extractor.policy_net.0.bias",
    "mlp_extractor.policy_net.2.weight", "mlp_extractor.policy_net.2.bias",
    "action_net.weight", "action_net.bias",
    "mlp_extractor.value_net.0.weight", "mlp_extractor.value_net.0.bias",
    "mlp_extractor.value_net.2.weight", "mlp_extractor.value_net.2.bias",
    "value_net.weight", "value_net.bias",
]


def _shapes(obs_dim: int, n_actions: int):
    H = HIDDEN
    return [(H, obs_dim), (H,), (H, H), (H,), (n_actions, H), (n_actions,),
            (H, obs_dim), (H,), (H, H), (H,), (1, H), (1,)]


class MlpPolicy:
    """weights: mapping from the SB3 state_dict names (SB3_KEYS) to arrays / tensors of the documented shapes."""

    PRECISIONS = {"auto": 0, "f32": 1, "f16x3": 2}

    def __init__(self, obs_dim: int, n_actions: int, weights: Mapping[str, object], device: int | str | torch.device = 0,
                 library: Optional[str] = None, precision: str = "auto"):
        if not torch.cuda.is_available():
            raise RuntimeError("MlpPolicy needs a HIP device: there is no CPU fallback")
        self.L = load_library(library)
        self.obs_dim, self.n_actions = int(obs_dim), int(n_actions)
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        parts = []
        for key, shape in zip(SB3_KEYS, _shapes(self.obs_dim, self.n_actions)):
            if key not in weights:
                raise KeyError(f"missing weight {key!r}")
            w = weights[key]
            w = w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else np.asarray(w)
            if tuple(w.shape) != tuple(shape):
                raise ValueError(f"{key}: shape {tuple(w.shape)}, expected {tuple(shape)}")
            parts.append(np.ascontiguousarray(w, dtype=np.float32).ravel())
        blob = np.concatenate(parts)
        assert blob.size == self.L.mse_policy_num_weights(self.obs_dim, self.n_actions)
        self.weights = {k: p.copy() for k, p in zip(SB3_KEYS, parts)}
        h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        check(self.L.mse_policy_create(C.byref(h), self.obs_dim, self.n_actions,
                                       blob.ctypes.data_as(C.POINTER(C.c_float)), dev_index))
        self._h = h
        self.set_precision(precision)

    def set_precision(self, precision: str) -> str:
        """"f32": exact f32 matrix products; "f16x3": three f16 MFMAs per product on 22-bit operand splits (logits
        within ~1e-6 of f32, 5x the matrix rate); "auto": f16x3 when the weights fit f16's range.  Returns the form
        in effect."""
        check(self.L.mse_policy_set_precision(self._h, self.PRECISIONS[precision]))
        self.precision = "f16x3" if self.L.mse_policy_precision(self._h) == 2 else "f32"
        return self.precision

    @classmethod
    def from_state_dict(cls, state_dict: Mapping[str, object], device=0, library: Optional[str] = None) -> "MlpPolicy":
        """From `model.policy.state_dict()` of the reference's (Maskable)PPO; dimensions are read off the tensors."""
        w1 = state_dict["mlp_extractor.policy_net.0.weight"]
        wa = state_dict["action_net.weight"]
        return cls(int(w1.shape[1]), int(wa.shape[0]), state_dict, device=device, library=library)

    @classmethod
    def random_init(cls, obs_dim: int, n_actions: int, seed: int = 0, device=0, library: Optional[str] = None,
                    precision: str = "auto") -> "MlpPolicy":
        """Random weights of the reference's architecture in SB3's initial scale (orthogonal-like magnitudes: hidden
        layers ~ sqrt(2) / sqrt(fan_in), action head 0.01, value head 1) - for benchmarks and smoke runs."""
        rng = np.random.default_rng(seed)
        w = {}
        for key, shape in zip(SB3_KEYS, _shapes(obs_dim, n_actions)):
            if len(shape) == 1:
                w[key] = np.zeros(shape, dtype=np.float32)
            else:
                gain = 0.01 if key == "action_net.weight" else (1.0 if key == "value_net.weight" else np.sqrt(2.0))
                w[key] = (rng.standard_normal(shape) * gain / np.sqrt(shape[1])).astype(np.float32)
        return cls(obs_dim, n_actions, w, device=device, library=library, precision=precision)

    def close(self):
        if getattr(self, "_h", None):
            self.L.mse_policy_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, obs: torch.Tensor, mask: Optional[torch.Tensor] = None, seed: int = 2024, t: int = 0,
                deterministic: bool = False, index_offset: int = 0, want_logits: bool = False,
                out: Optional[dict] = None) -> dict:
        """obs f32[N, D] (device), mask u8/bool[N, A] or None -> {"action" i32[N], "logp" f32[N], "value" f32[N]
        [, "logits" f32[N, A]]}."""
        if obs.dim() != 2 or obs.shape[1] != self.obs_dim:
            raise ValueError(f"obs must be [N, {self.obs_dim}]")
        obs = obs.to(device=self.device, dtype=torch.float32).contiguous()
        n = obs.shape[0]
        if mask is not None:
            if tuple(mask.shape) != (n, self.n_actions):
                raise ValueError(f"mask must be [N, {self.n_actions}]")
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        if out is None:  # pass a previous result back in to reuse its buffers
            out = {
                "action": torch.empty((n,), dtype=torch.int32, device=self.device),
                "logp": torch.empty((n,), dtype=torch.float32, device=self.device),
                "value": torch.empty((n,), dtype=torch.float32, device=self.device),
            }
            if want_logits:
                out["logits"] = torch.empty((n, self.n_actions), dtype=torch.float32, device=self.device)

        def ptr(x):
            return None if x is None else C.c_void_p(x.data_ptr())

        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            check(self.L.mse_policy_forward(self._h, n, int(index_offset), ptr(obs), ptr(mask), int(seed), int(t),
                                            1 if deterministic else 0, ptr(out["action"]), ptr(out["logp"]),
                                            ptr(out["value"]), ptr(out.get("logits")), stream))
        return out
