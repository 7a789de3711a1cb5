"""The bench's synthetic workloads (sina_amd/synth.py): what --divergence-mix and --exact-rate hand the pipeline."""
import numpy as np

from sina_amd import synth


def _ref_masks(refs, r):
    return (refs.ab[int(refs.off[r]):int(refs.off[r + 1])] >> 24).astype(np.uint8)


def test_per_query_rates_cycle_over_the_queries():
    """Rates given as sequences: query i gets rate[i % len] -- a rate of zero leaves the query an exact copy of its
    source reference (what bench.py --exact-rate builds), and windows of such queries are substrings of it."""
    refs = synth.make_refs(50, length=400, width=3000, seed=5)
    qs = synth.make_queries(refs, 40, seed=9, sub=[0.0, 0.05], dele=[0.0, 0.01], ins=[0.0, 0.01])
    exact = changed = 0
    for q in range(qs.n):
        m = qs.mask[int(qs.off[q]):int(qs.off[q + 1])]
        same = np.array_equal(m, _ref_masks(refs, int(qs.src[q])))
        if q % 2 == 0:
            assert same, q
            exact += 1
        else:
            changed += not same
    assert exact == 20 and changed >= 18  # (a 400-base query at 5 % substitutions is all but never untouched)
    w = synth.make_queries(refs, 10, seed=9, sub=[0.0], dele=[0.0], ins=[0.0], window=(1.0 / 3.0, 120))
    for q in range(w.n):
        m = w.mask[int(w.off[q]):int(w.off[q + 1])].tobytes()
        assert 0 < len(m) <= 120 and m in _ref_masks(refs, int(w.src[q])).tobytes()


def test_divergence_mix_spreads_four_rates_over_a_launch():
    refs = synth.make_refs(30, length=600, width=4000, seed=3)
    qs = synth.make_queries(refs, 64, seed=4, sub=[0.005, 0.03, 0.10, 0.20], dele=[0.0] * 4, ins=[0.0] * 4)
    diff = np.zeros(4)
    for q in range(qs.n):
        m = qs.mask[int(qs.off[q]):int(qs.off[q + 1])]
        r = _ref_masks(refs, int(qs.src[q]))
        assert len(m) == len(r)  # (no insertions or deletions asked for)
        diff[q % 4] += float((m != r).mean())
    diff /= qs.n / 4
    assert diff[0] < 0.015 and 0.015 < diff[1] < 0.05 and 0.07 < diff[2] < 0.13 and 0.16 < diff[3] < 0.24, diff
