"""Single-env dashboard trace: the reference's per-env Python ledgers rebuilt from the engine's trace records.

The reference appends to three ledgers every step - `reward_data` (src/envs_train/env_super.py:402-408, 928-946),
`press_actions_per_timestep` (:631-637, 730-736; env_monolith.py:136; env_2_press.py:131) and the `bale_count`
lists (:661-687) - and its dashboard reads them off the env (utils/plotting.py:32-48).  Unbounded lists cannot live
per GPU lane, so the engine keeps O(1) summaries for all envs and offers this opt-in trace for ONE env index:
`mse_trace_begin` makes every `mse_step` write one fixed-size record (include/mse.h MSE_TRACE_*); `EnvTrace` turns
records into the reference's ledger objects, in the reference's formats.  Pure NumPy: no torch, no device code.
"""
from __future__ import annotations

import numpy as np

MATERIALS = ["A", "B", "C", "D", "E"]
# column map of one record (include/mse.h)
ACTION, R_SORT, R_PRESS, SETTING, BELT, CONT_TRUE, CONT_FALSE, CONT_E = 0, 1, 2, 3, 4, 8, 12, 16
N_LOG, LOG, N_BALE, BALE, DONE, STEP, INTERNAL, REWARD, ACC_BELT = 17, 18, 22, 23, 29, 30, 31, 32, 33
OVERFLOW = 37  # 1 + index of the material that overflowed when check_overflow ended the episode, else 0


class EnvTrace:
    """reward_data / press_actions_per_timestep / bale_count of one episode, fed record by record."""

    def __init__(self, bale_standard_size: int = 200, bale_remainder_threshold: float = 0.5):
        self.bale_standard_size = int(bale_standard_size)
        self.bale_remainder_threshold = float(bale_remainder_threshold)
        self.reset()

    def reset(self):
        """What Env_Super.reset() does to the ledgers (env_super.py:392, 400-408)."""
        self.reward_data = {"Accuracy": [], "Setting": [], "Belt_Occupancy": [], "Reward": [], "Belt_Proportions": []}
        self.press_actions_per_timestep = []
        self.bale_count = {m: [] for m in MATERIALS}
        self.records = []

    # env_super.py:661-687: the list form of press_bale; q arrives as the stored integer int(q * 100)
    def _press_bale(self, material: str, n: int, q: int):
        bales = self.bale_count[material]
        S = self.bale_standard_size
        full, rem = n // S, n % S
        for _ in range(full):
            bales.append((S, q))
        if rem > 0:
            if rem > S * self.bale_remainder_threshold:
                bales.append((rem, q))
            elif bales:
                size, quality = bales[-1]
                bales[-1] = (size + rem, quality)
            else:
                bales.append((rem, q))

    def append(self, rec):
        rec = np.asarray(rec, dtype=np.float64)
        self.records.append(rec.copy())
        # press_bale calls come first in a step (check_press_status), in the reference's order
        for b in range(int(rec[N_BALE])):
            mat, n, q = (int(rec[BALE + 3 * b + c]) for c in range(3))
            self._press_bale(MATERIALS[mat], n, q)
        # press_actions_per_timestep: (0, None) | (press_id, material_id) | (111|222, material name)
        for j in range(int(rec[N_LOG])):
            code, mat = int(rec[LOG + 2 * j]), int(rec[LOG + 2 * j + 1])
            if code == 0:
                self.press_actions_per_timestep.append((0, None))
            elif code in (111, 222):
                self.press_actions_per_timestep.append((code, MATERIALS[mat]))
            else:
                self.press_actions_per_timestep.append((code, mat))
        # _log_step_data (env_super.py:928-946)
        rd = self.reward_data
        r_sort, r_press = float(rec[R_SORT]), float(rec[R_PRESS])
        rd["Reward"].append((r_sort, r_press))
        rd.setdefault("Total", []).append(r_sort + r_press)
        rd["Setting"].append(int(rec[SETTING]))
        belt = [int(v) for v in rec[BELT:BELT + 4]]
        total = sum(belt)
        # belt_occupancy is the input occupancy of the batch now on the belt: round(sum / 100, 2) (env_super.py:442, 456)
        rd["Belt_Occupancy"].append(round(total / 100, 2) if total else 0.0)
        rd["Belt_Proportions"].append({m: (belt[k] / total if total > 0 else 0) for k, m in enumerate(MATERIALS[:4])})
        for k, m in enumerate(MATERIALS):
            rd.setdefault(f"{m}_True", []).append(int(rec[CONT_TRUE + k]) if k < 4 else int(rec[CONT_E]))
            rd.setdefault(f"{m}_False", []).append(int(rec[CONT_FALSE + k]) if k < 4 else 0)

    def extend(self, records):
        for r in np.asarray(records, dtype=np.float64).reshape(-1, len(records[0]) if len(records) else 40):
            self.append(r)
        return self

    def as_arrays(self):
        """The ledgers as plain arrays, keys as the fixtures' `ledger_*` (oracle/ref_harness.py ledgers)."""
        rd = self.reward_data
        out = {
            "reward": np.asarray(rd["Reward"], dtype=np.float64).reshape(-1, 2),
            "total": np.asarray(rd.get("Total", []), dtype=np.float64),
            "setting": np.asarray(rd["Setting"], dtype=np.int64),
            "belt_occupancy": np.asarray(rd["Belt_Occupancy"], dtype=np.float64),
            "belt_proportions": np.asarray([[float(p[m]) for m in "ABCD"] for p in rd["Belt_Proportions"]],
                                           dtype=np.float64).reshape(-1, 4),
            "true": np.asarray([rd.get(f"{m}_True", []) for m in MATERIALS], dtype=np.int64).T.reshape(-1, 5),
            "false": np.asarray([rd.get(f"{m}_False", []) for m in MATERIALS], dtype=np.int64).T.reshape(-1, 5),
        }
        log = [(int(c), -1 if m is None else (MATERIALS.index(m) if isinstance(m, str) else int(m)))
               for c, m in self.press_actions_per_timestep]
        out["press_log"] = np.asarray(log, dtype=np.int64).reshape(-1, 2)
        for m in MATERIALS:
            out[f"bales_{m}"] = np.asarray(self.bale_count[m], dtype=np.int64).reshape(-1, 2)
        return out
