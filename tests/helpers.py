"""Shared helpers for the test-suite (not product code)."""
import numpy as np


def checksum(a: np.ndarray) -> str:
    """Order-sensitive digest, same definition as tests/golden/make_golden.py."""
    b = np.ascontiguousarray(a).view(np.uint8).ravel()
    pad = (-len(b)) % 4
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    wds = b.view(np.uint32).astype(np.uint64)
    idx = np.arange(1, len(wds) + 1, dtype=np.uint64)
    return hex(int((wds * idx).sum(dtype=np.uint64)))
