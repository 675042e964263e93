"""Compact, size-independent summary of a tensor for golden fixtures: shape,
fp64 sum / abs-sum, and up to `n` values at deterministic strided flat indices."""
import numpy as np


def sample_idx(size: int, n: int = 4096) -> np.ndarray:
    if size <= n:
        return np.arange(size)
    # odd stride walk over the flat index space
    step = size // n
    return (np.arange(n, dtype=np.int64) * step + (np.arange(n, dtype=np.int64) * 7919) % step) % size


def digest(x, n: int = 4096) -> dict:
    a = np.asarray(x, dtype=np.float32)
    flat = a.reshape(-1)
    return {
        "shape": np.asarray(a.shape, dtype=np.int64),
        "sum": np.float64(flat.astype(np.float64).sum()),
        "abssum": np.float64(np.abs(flat.astype(np.float64)).sum()),
        "samples": flat[sample_idx(flat.size, n)].copy(),
    }


def pack(prefix: str, d: dict) -> dict:
    return {f"{prefix}.{k}": v for k, v in d.items()}


def check(x, fx, prefix: str, rtol: float, atol: float):
    """Assert tensor x matches the digest stored under `prefix` in fixture fx."""
    a = np.asarray(x, dtype=np.float32)
    shape = tuple(int(s) for s in fx[f"{prefix}.shape"])
    assert tuple(a.shape) == shape, (prefix, a.shape, shape)
    flat = a.reshape(-1)
    ref = fx[f"{prefix}.samples"]
    got = flat[sample_idx(flat.size)]
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=prefix)
    tol = atol * flat.size + rtol * float(fx[f"{prefix}.abssum"])
    assert abs(float(flat.astype(np.float64).sum()) - float(fx[f"{prefix}.sum"])) <= tol, prefix
