import numpy as np


def fnv1a64(arr):
    """FNV-1a-64 over the little-endian bytes of `arr` (vectorised per byte position is not
    possible for FNV; use a chunked pure-python loop -- fine up to a few MB)."""
    b = np.frombuffer(np.ascontiguousarray(arr).tobytes(), np.uint8).tolist()
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h
