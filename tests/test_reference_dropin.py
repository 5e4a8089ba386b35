"""The drop-in boundary against the LIVE reference (CPU; skipped where /root/reference is absent, like the fixture-integrity test).

Two differential checks the golden fixtures cannot make (tests/dropin_driver.py has the details; it runs in a subprocess because the
reference needs the stand-in `gym` package on the path, which then also becomes this package's adapter base class):
  * the reference's OWN wrapper classes (inject_novelty, LimitActions, LidarInFront, AgentMap) stacked unchanged on this package's
    single-env adapter, against the same stack on the reference env - 15 stacks x 600 steps;
  * randomised configurations without a fixture (env id x map size x one or two novelties per case), this package's inject_novelty
    against the reference's; argument errors must match text for text."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, 'gym_novel_gridworlds')),
                                reason='the reference is only present in the build container')


JOBS = {'wrappers': ('wrappers', 2024, 600), 'random2024': ('random', 2024, 24, 300), 'random77': ('random', 77, 24, 300)}


@pytest.fixture(scope='module')
def runs():
    """The three driver runs side by side (each a single-threaded Python loop of ~1 minute): started together, collected as needed."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', MPLBACKEND='Agg', OMP_NUM_THREADS='1',
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'oracle', 'gym_shim'), REFERENCE, ROOT, os.path.join(ROOT, 'tests')]))
    procs = {k: subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dropin_driver.py')] + [str(a) for a in args], env=env, cwd=ROOT,
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k, args in JOBS.items()}
    done = {}

    def get(key):
        if key not in done:
            out, err = procs[key].communicate(timeout=900)
            done[key] = (procs[key].returncode, out, err)
        rc, out, err = done[key]
        assert rc == 0, out[-3000:] + err[-3000:]
        return out
    yield get
    for p in procs.values():
        if p.poll() is None:
            p.kill()


def test_reference_wrapper_classes_run_unchanged_on_the_adapter(runs):
    out = runs('wrappers')
    last = out.strip().splitlines()[-1].split()
    assert last[0] == 'WRAPPERS_OK' and int(last[1]) >= 8 and int(last[3]) >= 4800, out[-600:]     # >= 8 novelties x 600 steps


@pytest.mark.parametrize('seed', [2024, 77])
def test_unfixtured_configurations_match_the_reference(runs, seed):
    out = runs('random%d' % seed)
    last = out.strip().splitlines()[-1].split()
    assert last[0] == 'RANDOM_OK' and int(last[1]) == 24, out[-600:]
    assert int(last[3].strip('(')) <= 12 and int(last[-2]) >= 3000, out[-600:]                     # most cases really step
