"""Shared by the full-size / real-image GPU tests: one uint8 BGR frame through uwip_dehaze on the paths the pipe takes
in anger (fused window filters, 8-bit transmission table, recovery fused into the guided filter) against the C oracle
(oracle/dehaze_oracle.c, pinned by the reference's goldens in tests/test_oracle_dehaze.py)."""
import numpy as np
import torch

import _oracle
from uwimageproc_amd import bgdehaze as bg

TOL = 1e-9


def trunc_u8(x):
    return (np.asarray(x) * 255).astype(np.uint8)


def check_frame(ctx, orc, img, guard=True, B=None, what=""):
    """Returns a dict of counts for the caller's report.  Float stages <= 1e-9; 8-bit outputs differ only at rounding
    ties; the truncating casts of the exposure tail (BGDehaze.py:75-76) may differ only where restored*255 sits on an
    integer, and the tail is then checked on identical input."""
    t = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    Bd = None if B is None else torch.from_numpy(np.asarray(B, np.float64)).cuda()
    out_o, tap = orc.dehaze(img, 15, full=True, guard_s=guard, B=B, taps=("B", "idx", "refined", "restored", "final"))
    rep = {}
    # background light: same pixels (first-index rule on both sides) unless B is injected
    if B is None:
        Bdev, idx = bg.Background_light(ctx, t, 15, return_index=True)
        assert np.array_equal(idx.cpu().numpy()[0], tap["idx"]), what
        assert np.array_equal(Bdev.cpu().numpy()[0], tap["B"]), what
    # RC_correction through the fused path (no refined-t tap -> recovery inside the guided filter's second kernel)
    rc = bg.dehaze(ctx, t, 15, full=False, B=Bd, want_float=True)
    restored = rc["float"].cpu().numpy()[0]
    assert np.abs(restored - tap["restored"]).max() <= TOL, (what, float(np.abs(restored - tap["restored"]).max()))
    rep["rc_u8_ties"] = _oracle.assert_u8_differs_only_at_rounding_ties(rc["out"].cpu().numpy(), tap["restored"], what=what + " RC u8")
    # refined t (this tap takes the unfused recovery path)
    rt = bg.dehaze(ctx, t, 15, full=False, B=Bd, want_refined_t=True, want_float=True)
    assert np.abs(rt["refined_t"].cpu().numpy()[0] - tap["refined"]).max() <= TOL, what
    assert np.abs(rt["float"].cpu().numpy()[0] - tap["restored"]).max() <= TOL, what
    # the truncating casts: flips only where the oracle's restored*255 is (within 1e-6 of) an integer
    x = tap["restored"] * 255.0
    flips = trunc_u8(restored) != trunc_u8(tap["restored"])
    if flips.any():
        assert np.abs(x[flips] - np.rint(x[flips])).max() <= 1e-6, what
    rep["trunc_flips"] = int(flips.sum())
    # the exposure tail on identical input: the oracle's tail fed with the DEVICE's restored
    full = bg.dehaze(ctx, t, 15, full=True, B=Bd, want_float=True, guard_s=guard)
    got = full["float"].cpu().numpy()[0]
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import dehaze_oracle as dz
    exp = dz.adaptiveExp_tail(dz.normalize_input(img), restored, guard_s=guard)
    if np.isnan(exp).any():
        assert np.isnan(got).all() and full["out"].cpu().numpy().max() == 0, what
    else:
        assert np.abs(got - exp).max() <= TOL, (what, float(np.abs(got - exp).max()))
        rep["full_u8_ties"] = _oracle.assert_u8_differs_only_at_rounding_ties(full["out"].cpu().numpy(), exp, what=what + " FULL u8")
        # end to end against the oracle's own chain: identical when no truncation flipped, else the stated 2e-3 / 1 LSB
        e2e = np.abs(got - tap["final"]).max()
        assert e2e <= (TOL if rep["trunc_flips"] == 0 else 2e-3), (what, float(e2e), rep)
        assert np.abs(full["out"].cpu().numpy().astype(int) - out_o.astype(int)).max() <= 1, what
        rep["e2e_float"] = float(e2e)
    return rep, full["out"].cpu().numpy()
