"""Identity of the native sources the library was built from: the first 16 hex digits of the SHA-256 over every file of
posendf_amd/csrc/ and include/ (names and contents, sorted).  profiles/traffic.json records it next to the PMC traffic
figures; bench.py recomputes it and flags `traffic_stale` when the profiled build is not the one that runs."""
from __future__ import annotations

import hashlib
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_id() -> str:
    h = hashlib.sha256()
    for d in (os.path.join(_ROOT, "posendf_amd", "csrc"), os.path.join(_ROOT, "include")):
        for name in sorted(os.listdir(d)):
            path = os.path.join(d, name)
            if os.path.isfile(path) and name.endswith((".hip", ".h")):
                h.update(name.encode())
                with open(path, "rb") as f:
                    h.update(f.read())
    return h.hexdigest()[:16]
