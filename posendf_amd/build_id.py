"""Identity of the native sources of the FUSED DISTANCE KERNELS (what profiles/traffic.json describes): the first 16 hex
digits of the SHA-256 over the kernel sources, their shared headers and the launch code (names and contents, sorted).
profiles/traffic.json records it next to the PMC traffic figures; bench.py recomputes it and flags `traffic_stale` when the
profiled build is not the one that runs."""
from __future__ import annotations

import hashlib
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUSED_KERNEL_SOURCES = ("pndf_kernel.hip", "pndf_kernel_split.hip", "pndf_kernel_split_x2.hip", "pndf_device.h", "pndf_layout.h",
                        "pndf_args.h", "pndf_capi.hip")


def source_id() -> str:
    h = hashlib.sha256()
    d = os.path.join(_ROOT, "posendf_amd", "csrc")
    for name in sorted(FUSED_KERNEL_SOURCES):
        h.update(name.encode())
        with open(os.path.join(d, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]
