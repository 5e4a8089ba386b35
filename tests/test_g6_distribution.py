"""G6: the DISTRIBUTION of the reset passes the device draws without numpy's shuffle (include/ngw.h,
ngw_spec.n_passes) against the reference's own resets - per-cell frequencies of the pass item and of the agent cell and
the histogram of the item count over 10 000 reference resets (tests/golden/g6_<cfg>.npz, generator gen_g6.py).  The CPU
test runs the oracle's Philox mode (the restatement of the device algorithm); tests/test_hip_parity.py holds the HIP
kernels to that mode bit for bit and repeats this check on the GPU's own resets."""
import numpy as np
import pytest

from ngw_testlib import G6_CFGS, build_spec, g6_check, g6_stats, golden
import os


@pytest.mark.parametrize('cfg', G6_CFGS)
def test_oracle_philox_resets_follow_the_reference_distribution(cfg):
    from oracle.ngw_oracle import Oracle
    spec = build_spec(cfg)
    S = spec.map_size
    n = 32768
    g = dict(np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g6_%s.npz' % cfg)))
    o = Oracle(spec.compile(), n, seed=777)
    freq = np.zeros(S * S, np.int64); hist = np.zeros(S * S + 1, np.int64); agent = np.zeros(S * S, np.int64)
    for _ in range(2):                                  # two episodes per env: the stream is keyed by (env, episode)
        assert o.reset() == 0
        f, h, a = g6_stats(o.st.map, o.st.loc, int(g['item']), S)
        freq += f; hist += h; agent += a
    mean_ref, mean_got = g6_check(cfg, freq, hist, agent, 2 * n)
    assert abs(mean_ref - mean_got) <= 0.02 * max(mean_ref, 1.0) + 0.05


def test_mt_mode_keeps_numpy_call_sequence():
    """The sparse form exists in the Philox mode only: the MT19937 mode still reproduces the reference's resets, stream
    position included (that is what test_oracle_golden.py checks for every configuration; this is the one-line reminder)."""
    from oracle.ngw_oracle import MT19937, Oracle
    cfg = 'add12m'
    g = golden(cfg)
    spec = build_spec(cfg)
    o = Oracle(spec.compile(), 1)
    mt = MT19937(0)                                     # G2: seed s, three consecutive resets, then the next raw word
    for j in range(3):
        assert o.reset_mt(mt) == 0
        assert (o.st.map[0] == g['rs_map'][0, j]).all()
    assert mt.next() == g['rs_next_word'][0]
