"""Hashes of the kernel sources, so that a measured figure stored beside a kernel's name
(profiles/traffic.json) can be recognised as stale once that kernel's source has changed."""
from __future__ import annotations

import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")

# kernel-name prefix -> the files its device code comes from
_SOURCES = {
    "k_fold": ["mvo_kernels.h", "mvo_wave.h", "mvo_device.h"],
    "k_pass": ["mvo_kernels.h", "mvo_wave.h", "mvo_device.h"],
    "k_reduce_slots": ["mvo_kernels.h"],
    "k_lm4": ["mvo_lm4.h", "mvo_lm.h", "mvo_grid.h", "mvo_kernels.h", "mvo_wave.h", "mvo_device.h"],
    "k_lm": ["mvo_lm.h", "mvo_grid.h", "mvo_kernels.h", "mvo_wave.h", "mvo_device.h"],
    "k_scaler": ["mvo_scaler.hip", "mvo_wave.h"],
    "k_tracks": ["mvo_tracks.hip", "mvo_wave.h"],
    "k_ba": ["mvo_ba.hip", "mvo_wave.h"],
    "k_epi": ["mvo_epipolar.hip", "mvo_eig3.h", "mvo_wave.h"],
}


def source_hash(kernel: str):
    """sha256 (first 16 hex digits) over the source files of `kernel` (matched by name prefix), or
    None for a kernel this table does not know."""
    k = kernel.replace("void ", "").strip()
    for prefix, files in _SOURCES.items():
        if k.startswith(prefix):
            h = hashlib.sha256()
            for f in files:
                with open(os.path.join(_CSRC, f), "rb") as fh:
                    h.update(fh.read())
            return h.hexdigest()[:16]
    return None
