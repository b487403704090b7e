"""Many SMALL calls back to back against known answers (-m gpu).  A call whose kernels take tens of microseconds returns a few microseconds after its last
kernel has written the results into pinned memory -- the host polls them instead of waiting for the stream -- and that is where a visibility race shows:
round 6's first form (a sequence word behind a system-scope fence) returned the PREVIOUS call's words about once in 50 000 calls, which the rest of the
suite saw once in ~20 runs (profiles/NOTES_r6.md section 15; tools/race_hunt.py is the long version of this test).  Every result word now travels as a
(word, call number) pair; this test keeps ~20 000 calls on it, through kept and fresh contexts, the resident set and the window table."""
import numpy as np
import pytest

import mopro_msm_hip as mh
from oracle import bn254_oracle as orc

pytestmark = pytest.mark.gpu


def test_twenty_thousand_small_calls_all_equal_their_known_answers():
    rng = np.random.default_rng(11)
    nmax = 2048
    k_all = orc.gen_scalars(4242, nmax, nonzero=True)
    bases_all = orc.gen_bases_from_logs(k_all, orc.FORM_MONT)
    s_all = orc.gen_scalars(4343, nmax)
    inst = []
    for n, zeros in [(8, False), (8, True), (16, False), (37, False), (200, True), (681, True), (2048, False)]:
        off = int(rng.integers(0, nmax - n + 1))
        b, s = bases_all[off:off + n].copy(), s_all[off:off + n].copy()
        if zeros:
            s[rng.random(n) < 0.6] = 0
        exp, einf, _ = orc.msm_pippenger(b, s, orc.FORM_MONT, None)
        inst.append((b, s, exp, bool(einf)))
    cfgs = [(0, 0), (0, mh.FLAG_NO_GLV), (11, mh.FLAG_NO_GLV), (0, mh.FLAG_WINDOW_TABLE)]
    kept = {c: mh.MsmContext(window_bits=c[0], flags=c[1]) for c in cfgs}
    calls, bad = 0, []

    def check(tag, r, exp, einf):
        nonlocal calls
        calls += 1
        if r.is_infinity != einf or not (r.affine_std == exp).all():
            bad.append((calls,) + tag)

    try:
        while calls < 20000:
            for ii in rng.permutation(len(inst)):
                b, s, exp, einf = inst[ii]
                cfg = cfgs[int(rng.integers(0, len(cfgs)))]
                fresh = rng.random() < 0.2
                ctx = mh.MsmContext(window_bits=cfg[0], flags=cfg[1]) if fresh else kept[cfg]
                tag = (len(s), cfg, "fresh" if fresh else "kept")
                check(tag + ("host",), ctx.msm(b, s, mh.FORM_MONT), exp, einf)
                if rng.random() < 0.5:
                    ctx.upload_bases(b, mh.FORM_MONT)
                    check(tag + ("resident",), ctx.msm_resident(s), exp, einf)
                    check(tag + ("host again",), ctx.msm(b, s, mh.FORM_MONT), exp, einf)
                if fresh:
                    ctx.close()
    finally:
        for c in kept.values():
            c.close()
    assert not bad, (len(bad), "of", calls, "calls returned another point:", bad[:5])
