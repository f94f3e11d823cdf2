"""Test infrastructure only (never imported by the product): the scipy mirror of the reference's ACLAHE parameter choice,
modules/aclahe/python/ACLAHE.py:66-129 + functions.py:49-93 -- the same scipy calls the reference makes (curve_fit with
p0=(7,0.4,0.9,5), splrep / splev), so that the product's native MINPACK / spline restatement (uwip_aclahe_knee,
uwip_aclahe_select in csrc/aclahe_select.cpp) can be checked against what scipy itself answers.  The reference-generated
golden indices live in tests/golden/aclahe_knee.npz."""
BLOCK_SIZES = (2, 4, 8, 16, 32)                       # aclahe.cpp:161

# ---------------------------------------------------------------------------
# C4: parameter selection, modules/aclahe/python/ACLAHE.py:66-129 +
# functions.py:49-93.  The reference does this on the host with scipy; so does
# this mirror (same calls: curve_fit p0=(7,0.4,0.9,5), splrep/splev).  The
# sweep loop of ACLAHE.py:40-47 is broken by indentation; the
# evident intent is implemented: entropy-vs-CL curves sampled at CL = 0.5 ...
# 24.5 (49 samples, functions.graficar slices [2:51]).
# ---------------------------------------------------------------------------
def _dexp(x, p0, p1, p2, p3):
    import numpy as np
    return p0 * np.exp(-p1 * x) + p2 * np.exp(-p3 * x)


def knee_index(xs, ys) -> int:
    """DerivadaY + DerivadaX + Curvatura (functions.py:49-93) -> arg-max index, or -1
    when curve_fit does not converge (the reference would raise there)."""
    import warnings

    import numpy as np
    from scipy.interpolate import splev, splrep
    from scipy.optimize import curve_fit

    u = np.linspace(1, 49, 49)
    p0 = (7, 0.4, 0.9, 5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            popt, _ = curve_fit(_dexp, u, ys, p0)
            x22 = np.linspace(1, 25, 25)
            tck = splrep(x22, _dexp(x22, *popt))
            x222 = np.linspace(1, 25, 49)
            y220 = splev(x222, tck, der=1)
            y221 = splev(x222, tck, der=2)
            popt2, _ = curve_fit(_dexp, u, xs, p0)
            tck2 = splrep(x22, _dexp(x22, *popt2))
            y223 = splev(x222, tck2, der=1)
            y225 = splev(x222, tck2, der=2)
        except Exception:
            return -1
        k = np.abs(y223 * y221 - y220 * y225) / np.power(np.power(y223, 2) + np.power(y220, 2), 1.5)
        return int(np.argmax(k))


def select_parameters(table, entropy_at=None):
    """(BS, CL) from one frame's 5 x 51 entropy table (ACLAHE.py:66-129).
    CL is the largest of the five curvature arg-max INDICES (:92-96, as written);
    BS is the block size whose entropy at clip limit CL is largest, compared in
    float16 with the LAST maximum winning (:102-125).  ``entropy_at(bs_index, cl)``
    supplies entropies for clip limits outside the swept grid."""
    import numpy as np

    table = np.asarray(table, dtype=np.float32)
    xs = (np.arange(51, dtype=np.float32) * 0.5)[1:50]
    idx = [knee_index(xs, table[g][1:50]) for g in range(5)]
    d = max(idx)
    if d < 0:
        d = 0
    ent = np.zeros(5, np.float16)
    for g in range(5):
        if 2 * d <= 50:
            ent[g] = table[g][2 * d]
        else:
            ent[g] = entropy_at(g, float(d)) if entropy_at is not None else table[g][50]
    w = int(np.nonzero(ent == ent.max())[0][-1])
    return BLOCK_SIZES[w], int(d)
