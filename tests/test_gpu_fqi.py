"""The batch path on the GPU (grlx_fqi_*, BASELINE.json configs[4]) against the oracle with the same specification
(oracle/fqi.c, portable arithmetic, the GPU's gradient summation tree): bit for bit -- the transition store, the
targets, all 101 network parameters after every batch, the rows, the RNG streams.  PARITY UNPINNED with respect to the
reference (see oracle/fqi.c D1-D4); tolerance against the oracle written here: 0 ulp."""
import numpy as np
import pytest

from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(a, b, what=""):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shapes {a.shape} vs {b.shape}"
    bad = np.nonzero(bits(a).ravel() != bits(b).ravel())[0]
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} differ, first at {bad[:5]}: {a.ravel()[bad[0]]!r} vs {b.ravel()[bad[0]]!r}"


def _both(grlx, seeds, batches, **over):
    cfg = grlx.pendulum_fqi_config(len(seeds), max_batches=batches, **over)
    r = grlx.FqiRunner(cfg, seeds)
    oracles = [ob.FqiExperiment(ob.pendulum_fqi_spec(**over), seed=int(s)) for s in seeds]
    for b in range(batches):
        r.run_batch()
        r.sync()
        for k, e in enumerate(oracles):
            row = e.run_batch()
            n = (b + 1) * cfg.batch_size
            inp, nobs, rew, tgt = r.transitions(k, 0, n)
            oin, onobs, orew, otgt = e.transitions()
            assert_bit_equal(inp, oin, f"batch {b} replica {k}: normalised inputs")
            assert_bit_equal(nobs, onobs, f"batch {b} replica {k}: next observations")
            assert_bit_equal(rew, orew, f"batch {b} replica {k}: rewards")
            assert_bit_equal(tgt, otgt, f"batch {b} replica {k}: targets of the last iteration")
            assert_bit_equal(r.params(k), e.params(), f"batch {b} replica {k}: network parameters")
            info, oinfo = r.info(k), e.info()
            assert info["n"] == n and info["iterations"] == oinfo["iterations"]
            assert_bit_equal([info["maxdelta"], info["error"]], [oinfo["maxdelta"], oinfo["error"]], f"batch {b} replica {k}: maxdelta, mse")
            assert info["rng"] == e.rng()[:2]
            bb, tt, rr = r.rows(k, b + 1)
            assert bb[b] == row.trial and tt[b] == row.steps
            assert_bit_equal(rr[b:b + 1], [row.reward], f"batch {b} replica {k}: return of the test trial")
    for e in oracles:
        e.close()
    r.close()


def test_fqi_small_batches_bit_exact(grlx):
    """Ragged sizes: 700 transitions per batch (2.7 chunks of 256), three replicas, two batches (the second rebuild runs
    over 1400 stored transitions = 5.5 chunks)."""
    _both(grlx, [1, 2, 3], 2, batch_size=700, iterations=3, epochs=40)


def test_fqi_many_chunks_bit_exact(grlx):
    """More chunks than the 64 lanes of the third reduction level (20000 transitions = 79 chunks)."""
    _both(grlx, [7], 1, batch_size=20000, iterations=2, epochs=12)


def test_fqi_several_rounds_of_chunks_on_several_replicas_bit_exact(grlx):
    """The persistent epochs kernel at a size where every wave walks several chunks (level 2 of the sum in registers): two
    batches of 6500 transitions -- the second rebuild runs over 13000 stored transitions = 204 chunks, i.e. 3-4 chunks per
    wave with a ragged last one -- on five replicas (80 resident blocks that meet once per epoch), 2 x 25 epochs each."""
    _both(grlx, [11, 12, 13, 14, 15], 2, batch_size=6500, iterations=2, epochs=25)


def test_fqi_at_the_bench_s_size_bit_exact(grlx):
    """The shape bench.py times -- 16 replicas (256 resident blocks: every CU), 100 000 transitions per batch = 1563 chunks of 64 with a
    ragged last one, 24 or 25 chunks per wave -- with few epochs so that the scalar oracle finishes in seconds: transitions, targets, all
    101 parameters, error, rows and streams of every replica against the oracle."""
    _both(grlx, list(range(101, 117)), 1, batch_size=100000, iterations=2, epochs=3)


def test_fqi_more_replicas_than_one_launch_holds(grlx):
    """18 replicas need 288 resident blocks: more than the 256 CUs hold, so the epochs run as two cooperative launches
    (16 + 2 replicas) per iteration; results per replica do not depend on the grouping."""
    seeds = list(range(21, 39))
    cfg = grlx.pendulum_fqi_config(len(seeds), max_batches=1, batch_size=640, iterations=2, epochs=6)
    r = grlx.FqiRunner(cfg, seeds)
    r.run_batch()
    r.sync()
    for k in (0, 15, 16, 17):
        e = ob.FqiExperiment(ob.pendulum_fqi_spec(batch_size=640, iterations=2, epochs=6), seed=seeds[k])
        row = e.run_batch()
        assert_bit_equal(r.params(k), e.params(), f"replica {k}: network parameters")
        assert_bit_equal(r.rows(k, 1)[2], [row.reward], f"replica {k}: return")
        e.close()
    r.close()


@pytest.mark.parametrize("hidden", [8, 16, 32, 64])
def test_fqi_other_hidden_layer_widths_bit_exact(grlx, hidden):
    """representation/parameterized/ann:hiddens = [8], [16], [32], [64] (41 .. 321 parameters: one to six per lane of the summing
    wave; 64 units run two waves per block, 32 blocks per replica): two batches of 4500 transitions (the second rebuild: 141 chunks,
    two or three per wave, ragged), two replicas."""
    _both(grlx, [3, 4], 2, batch_size=4500, iterations=2, epochs=10, hidden=hidden)


@pytest.mark.parametrize("eta", [0.7, 0.05, -0.001, -0.01])
@pytest.mark.parametrize("hidden", [20, 32])
def test_fqi_gradient_descent_and_rmsprop_bit_exact(grlx, hidden, eta):
    """representation/parameterized/ann:eta > 0 (gradient descent: W -= eta Delta / samples, ann.cpp:202-206) and eta < 0 (RMSprop,
    :214-219: a square root and two divisions per parameter and epoch), against the oracle with the same summation tree."""
    _both(grlx, [5, 6], 2, batch_size=1300, iterations=3, epochs=30, hidden=hidden, eta=eta)


@pytest.mark.parametrize("hidden", [20, 64])
def test_fqi_saturated_logistic_bit_exact(grlx, hidden):
    """The logistic's argument is clamped to +-690 (oracle/fqi.c D5), which is what lets the kernel evaluate it without range handling.  Gradient
    descent with eta = 2 blows the weights up to 10^4 within the first iteration: net inputs in the thousands, the clamp is taken on both
    sides and the units saturate to exactly 1 and to 1 / (1 + exp(690)); all of it bit-exact against the oracle."""
    _both(grlx, [5, 6, 7], 2, batch_size=500, iterations=3, epochs=40, eta=2.0, hidden=hidden)


def test_fqi_iteration_loop_stops_per_replica(grlx):
    """gamma = 0: the targets are the rewards and the second iteration changes nothing (fqi.cpp:213); the stop is taken
    on the device, per replica, without a host round trip."""
    _both(grlx, [4, 5], 1, batch_size=300, iterations=6, epochs=5, gamma=0.0)


def test_fqi_reference_yaml_first_row(grlx):
    """tests/pendulum-fqi-ann.yaml as the reference ships it (1000 transitions, 10 iterations x 500 epochs), seed 1: the first
    row is the template's first row (-3508.07, the constant-torque return); both rows equal the oracle's bit for bit."""
    import os
    cfg = grlx.pendulum_fqi_config(1)
    r = grlx.FqiRunner(cfg, [1])
    e = ob.FqiExperiment(ob.pendulum_fqi_spec(), seed=1)
    r.run_batch(); r.run_batch(); r.sync()
    rows = [e.run_batch(), e.run_batch()]
    b, t, rew = r.rows(0, 2)
    assert list(b) == [0, 1] and list(t) == [0, 1000]
    assert_bit_equal(rew, [x.reward for x in rows], "returns of the two test trials")
    assert_bit_equal(r.params(0), e.params(), "network parameters after two batches")
    template = open(os.path.join(os.path.dirname(__file__), "golden", "pendulum-fqi-ann-0.txt")).readline()
    assert "%15d%15d%15s\n" % (b[0], t[0], "%g" % rew[0]) == template
    e.close(); r.close()


def test_fqi_validation(grlx):
    capi = grlx.capi
    for over in (dict(hidden=7), dict(hidden=128), dict(eta=2.5), dict(eta=float("nan")), dict(env=1), dict(batch_size=0), dict(timeout=float("nan")), dict(action_steps=0)):
        with pytest.raises(capi.GrlxError) as ei:
            grlx.FqiRunner(grlx.pendulum_fqi_config(1, **over), [1])
        assert ei.value.code == capi.ERR_INVALID
    r = grlx.FqiRunner(grlx.pendulum_fqi_config(1, batch_size=100, iterations=1, epochs=1, max_batches=1), [1])
    r.run_batch()
    with pytest.raises(capi.GrlxError) as ei:
        r.run_batch()
    assert ei.value.code == capi.ERR_ROWS_FULL
    r.close()


def test_fqi_reference_yaml_through_the_deployer(grlx, tmp_path):
    """`grlxd -s 1 tests/pendulum-fqi-ann.yaml` (the reference's own file, unmodified): experiment/batch_learning in the C++ host layer
    lowers the graph to a grlx_fqi_config and writes `<output>-0.txt` -- the two rows test_fqi_reference_yaml_first_row checks: the first
    is the template's first row byte for byte, both equal the oracle's."""
    import os
    import subprocess
    from grl_amd import _build
    grlxd = _build.build_host()
    yaml = os.path.join(os.path.dirname(__file__), "golden", "pendulum-fqi-ann.yaml")
    res = subprocess.run([grlxd, "-s", "1", yaml], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    e = ob.FqiExperiment(ob.pendulum_fqi_spec(), seed=1)
    want = "".join(e.format_row(e.run_batch()) for _ in range(2))
    e.close()
    got = (tmp_path / "pendulum-fqi-ann-0.txt").read_text()
    assert got == want and res.stdout == want
    template = open(os.path.join(os.path.dirname(__file__), "golden", "pendulum-fqi-ann-0.txt")).read().splitlines()
    assert got.splitlines()[0] == template[0]
    # clones: `-r 3` runs seeds 1, 2, 3 and writes <output>-0@i.txt (multi.cpp:52-56); clone 0 is the run above
    res = subprocess.run([grlxd, "-s", "1", "-r", "3", "-q", yaml], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    assert (tmp_path / "pendulum-fqi-ann-0@0.txt").read_text() == want
    e = ob.FqiExperiment(ob.pendulum_fqi_spec(), seed=3)
    assert (tmp_path / "pendulum-fqi-ann-0@2.txt").read_text() == "".join(e.format_row(e.run_batch()) for _ in range(2))
    e.close()
