"""Builds libmse_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
SOURCES = [os.path.join(PKG_DIR, "csrc", "mse_lib.hip"), os.path.join(PKG_DIR, "csrc", "mse_policy.hip")]
HEADERS = [os.path.join(PKG_DIR, "csrc", h) for h in ("mse_device.h", "mse_policy_device.h", "mse_policy_stream.h")] + \
          [os.path.join(REPO_ROOT, "include", "mse.h")]
LIB_PATH = os.environ.get("MSE_LIB_PATH") or os.path.join(PKG_DIR, "libmse_hip.so")  # override: experiments only

HIPCC_FLAGS = [
    "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
    # numpy evaluates the restated fp64 expressions with separately rounded operations
    "-ffp-contract=off", "-fno-fast-math",
]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmse_hip.so cannot be built")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    cmd = [hipcc_path(), *HIPCC_FLAGS, "-I", os.path.join(REPO_ROOT, "include"), *SOURCES, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


# Test-only variant: the near-tie window of the integer `choice` decision widened from 2^-23 to 2^-5 of the
# draws, so the literal-cdf branches (and the ring kernel's redo of a whole step) run all the time.
WIDETIE_LIB_PATH = os.path.join(PKG_DIR, "libmse_hip_widetie.so")


def build_widetie_library(force: bool = False) -> str:
    if not force and os.path.exists(WIDETIE_LIB_PATH) and \
            all(os.path.getmtime(p) <= os.path.getmtime(WIDETIE_LIB_PATH) for p in SOURCES + HEADERS):
        return WIDETIE_LIB_PATH
    cmd = [hipcc_path(), *HIPCC_FLAGS, "-DMSE_TIE_WINDOW=0x08000000u", "-I", os.path.join(REPO_ROOT, "include"),
           *SOURCES, "-o", WIDETIE_LIB_PATH]
    subprocess.run(cmd, check=True)
    return WIDETIE_LIB_PATH


if __name__ == "__main__":
    print(build_widetie_library(force=True))
    print(build_library(force=True, verbose=True))
