import numpy as np


def close(a, b, tol=1e-5):
    """BASELINE north_star tolerance for fp32 results: |a-b| <= tol * max(1, |b|)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b)
    bound = tol * np.maximum(1.0, np.abs(b))
    ok = np.all(err <= bound)
    if not ok:
        i = np.unravel_index(np.argmax(err - bound), err.shape)
        print(f"mismatch at {i}: got {a[i]!r} want {b[i]!r} (err {err[i]:.3e})")
    return ok
