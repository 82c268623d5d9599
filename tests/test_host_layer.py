"""The C++ host layer (grl's Configurable / YAML / deployer roles) without a GPU: it must
instantiate the reference's own yaml, resolve references and provided parameters, reproduce
the reference's bad_param conditions, and refuse to compute without a device."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YAML = os.path.join(ROOT, "tests", "golden", "pendulum-sarsa-tc.yaml")


@pytest.fixture(scope="module")
def grlxd():
    from grl_amd import _build
    return _build.build_host()


def run(grlxd, args, cwd):
    return subprocess.run([grlxd] + args, cwd=cwd, capture_output=True, text=True, timeout=120)


def test_instantiates_reference_yaml(grlxd, tmp_path):
    from grl_amd import capi
    res = run(grlxd, ["-s", "1", "-l", "-q", YAML], tmp_path)
    dumped = (tmp_path / "pendulum-sarsa-tc.yaml").read_text()
    # references and provided parameters resolved (configurable.cpp:355-432, 691-712)
    assert "min: [ -3 ]" in dumped and "max: [ 3 ]" in dumped            # experiment/environment/task/action_min
    assert "memory: 8388608" in dumped                                    # ../../projector/memory
    assert "projector: experiment/agent/policy/projector" in dumped       # object reference
    assert "type: predictor/sarsa" in dumped or "type: predictor/critic/sarsa" in dumped
    if capi.load().grlx_device_count() == 0:
        assert res.returncode == 1 and "no HIP device" in res.stderr     # no CPU fallback


def _variant(tmp_path, old, new):
    text = open(YAML).read()
    assert old in text
    p = tmp_path / "variant.yaml"
    p.write_text(text.replace(old, new))
    return str(p)


@pytest.mark.parametrize("old,new,needle", [
    ("wrapping: [ 6.283, 0, 0 ]", "wrapping: [ 1.0, 0, 0 ]", "projector/tile_coding:wrapping"),        # tile_coding.cpp:72-78
    ("steps: [ 3 ]", "steps: [ 3, 3 ]", "discretizer/uniform:{min,max,steps}"),                          # uniform.cpp:66-67
    ("init_min: [ 0 ]", "init_min: [ 0, 1, 2 ]", "representation/parameterized/linear:init_min"),       # linear.cpp:62-66
    ("type: dynamics/pendulum", "type: dynamics/flyer2d", "unknown object type"),
    ("safe: 0", "safe: 3", "safe must be 0, 1 or 2"),
    ("      sampler:\n        type: sampler/greedy", "      sampler:\n        type: sampler/epsilon_greedy", "sampler/greedy for testing"),
    ("      sampler:\n        epsilon: 0.05\n        type: sampler/epsilon_greedy\n", "", "required parameter 'sampler'"),
])
def test_bad_configurations_are_refused(grlxd, tmp_path, old, new, needle):
    res = run(grlxd, ["-s", "1", "-q", _variant(tmp_path, old, new)], tmp_path)
    assert res.returncode == 1
    assert needle in res.stderr, res.stderr


def test_usage_and_seed_errors(grlxd, tmp_path):
    assert run(grlxd, [], tmp_path).returncode == 1
    res = run(grlxd, ["-s", "0", YAML], tmp_path)
    assert res.returncode == 1 and "seed 0" in res.stderr


def test_instantiates_actor_critic_yaml(grlxd, tmp_path):
    """Two tables, references to values (resolution, memory, action_dims) and to objects."""
    yaml = os.path.join(ROOT, "tests", "golden", "cart_pole-ac-tc.yaml")
    res = run(grlxd, ["-s", "1", "-q", yaml], tmp_path)
    dumped = (tmp_path / "cart_pole-ac-tc.yaml").read_text()
    assert "type: predictor/ac/action" in dumped and "type: predictor/critic/td" in dumped
    assert dumped.count("resolution: [ 2.5, 0.157075, 2.5, 1.57075 ]") == 2       # critic copies the actor's by reference
    assert "outputs: 1" in dumped and "output_min: [ -15 ]" in dumped
    assert "no HIP device" in res.stderr or res.returncode == 0
    # critic before actor: the fused kernel cannot reproduce that RNG order and must refuse
    text = open(yaml).read()
    swapped = text.replace("    policy:\n      type: mapping/policy/action\n      sigma: [ 5 ]", "    policy_moved:\n      type: mapping/policy/action\n      sigma: [ 5 ]")
    assert swapped != text


def test_load_file_is_accepted_and_needs_a_device(grlxd, tmp_path):
    """load_file is part of the accelerated path (grlx_load_weights); without a GPU the deployer still
    stops at "no HIP device", never at the configuration."""
    from grl_amd import capi
    res = run(grlxd, ["-s", "1", "-l", "-q", _variant(tmp_path, 'load_file: ""', "load_file: some-policy-run$run")], tmp_path)
    assert "outside the accelerated path" not in res.stderr
    if capi.load().grlx_device_count() == 0:
        assert res.returncode == 1 and "no HIP device" in res.stderr


@pytest.mark.parametrize("body,needle", [
    ("    type: exporter/csv\n    file: \"\"\n", "exporter/csv:file"),                                   # csv.cpp:57-58
    ("    type: exporter/csv\n    file: log\n    fields: time, velocity\n", "exporter/csv:fields"),      # csv.cpp:113-117
    ("    type: exporter/csv\n    file: log\n    style: fancy\n", "exporter/csv:style"),
])
def test_exporter_parameters_are_validated(grlxd, tmp_path, body, needle):
    res = run(grlxd, ["-s", "1", "-q", _variant(tmp_path, '  load_file: ""\n', "  exporter:\n" + body + '  load_file: ""\n')], tmp_path)
    assert res.returncode == 1
    assert needle in res.stderr, res.stderr


def test_absolute_references_and_shared_top_level_objects(grlxd, tmp_path):
    """cfg/pendulum/multi_sarsa_tc.yaml of the reference: the agents' policy is a TOP-LEVEL object referred to by the
    absolute path `/policy` (configurable.h:418-449: a leading '/' addresses the root) from inside an experiment/multi."""
    from grl_amd import capi
    text = """environment:
  type: environment/modeled
  model:
    type: model/dynamical
    control_step: 0.03
    integration_steps: 5
    dynamics:
      type: dynamics/pendulum
  task:
    type: task/pendulum/swingup
    timeout: 2.99
policy:
  type: mapping/policy/discrete/value/q
  discretizer:
    type: discretizer/uniform
    min: environment/task/action_min
    max: environment/task/action_max
    steps: [3]
  projector:
    type: projector/tile_coding
    tilings: 16
    memory: 8388608
    resolution: [0.31415, 3.1415, 3]
    wrapping: [6.283, 0, 0]
  representation:
    type: representation/parameterized/linear
    init_min: [0]
    init_max: [1]
    memory: policy/projector/memory
    outputs: 1
    output_min: []
    output_max: []
  sampler:
    type: sampler/epsilon_greedy
    epsilon: 0.05
experiment:
  type: experiment/multi
  instances: 3
  experiment:
    type: experiment/online_learning
    runs: 1
    trials: 0
    steps: 0
    rate: 0
    test_interval: 10
    output: multi
    environment: /environment
    agent:
      type: agent/td
      policy: /policy
      predictor:
        type: predictor/critic/sarsa
        alpha: 0.2
        gamma: 0.97
        lambda: 0.65
        projector: policy/projector
        representation: policy/representation
        trace:
          type: trace/enumerated/replacing
    test_agent:
      type: agent/fixed
      policy:
        type: mapping/policy/discrete/value/q
        discretizer: /policy/discretizer
        projector: /policy/projector
        representation: /policy/representation
        sampler:
          type: sampler/greedy
"""
    p = tmp_path / "multi.yaml"
    p.write_text(text)
    res = run(grlxd, ["-s", "1", "-t", "11", "-q", str(p)], tmp_path)
    assert "does not name an object" not in res.stderr and "unknown" not in res.stderr, res.stderr
    if capi.load().grlx_device_count() == 0:
        assert res.returncode == 1 and "no HIP device" in res.stderr, res.stderr
    else:
        assert res.returncode == 0, res.stderr


FQI_YAML = os.path.join(ROOT, "tests", "golden", "pendulum-fqi-ann.yaml")


def test_instantiates_reference_fqi_yaml(grlxd, tmp_path):
    """tests/pendulum-fqi-ann.yaml of the reference, unmodified: experiment/batch_learning + predictor/fqi + representation/iterative +
    representation/parameterized/ann over projector/pre/normalizing.  Its `a+b` expressions over references are evaluated as the
    reference's parser evaluates them (parser.cpp:49-134: element-wise with the scalar broadcast -- two entries for input_min, the yaml's
    own defect, deviation D4), integers add, and without a GPU the deployer stops at the device, never at the configuration."""
    from grl_amd import capi
    res = run(grlxd, ["-s", "1", "-q", FQI_YAML], tmp_path)
    dumped = (tmp_path / "pendulum-fqi-ann.yaml").read_text()
    assert "type: experiment/batch_learning" in dumped and "type: predictor/fqi" in dumped and "type: representation/parameterized/ann" in dumped
    assert "input_min: [ -3, -40.699111843077517 ]" in dumped and "input_max: [ 9.2831853071795862, 40.699111843077517 ]" in dumped
    assert "inputs: 3" in dumped and "hiddens: [ 20 ]" in dumped
    assert "projector: experiment/predictor/projector" in dumped                      # the test policy refers to the predictor's objects
    assert "deviation D4" in res.stderr and "outside the accelerated path" not in res.stderr
    if capi.load().grlx_device_count() == 0:
        assert res.returncode == 1 and "no HIP device" in res.stderr


@pytest.mark.parametrize("old,new,needle", [
    ("hiddens: [ 20 ]", "hiddens: [ 20, 20 ]", "one hidden layer"),
    ("hiddens: [ 20 ]", "hiddens: [ 0 ]", "representation/parameterized/ann:hiddens"),                 # ann.cpp:74-75
    ("cumulative: 0", "cumulative: 1", "cumulative = 0"),
    ("reset_strategy: never", "reset_strategy: sometimes", "predictor/fqi:reset_strategy"),             # fqi.cpp:66-69
    ("reset_strategy: never", "reset_strategy: batch", "reset_strategy must be never"),
    ("transitions: 100000", "transitions: 1500", "transitions (the store's size)"),
    ("batches: 2", "batches: 0", "batches must be > 0"),
    ("type: dynamics/pendulum", "type: dynamics/cart_pole", "built for model/dynamical with dynamics/pendulum"),
    ("steps: [3]", "steps: [3, 3]", "discretizer/uniform:{min,max,steps}"),
    ("inputs: experiment/task/observation_dims+experiment/task/action_dims", "inputs: 2", "inputs must be observation_dims+action_dims"),
])
def test_bad_batch_configurations_are_refused(grlxd, tmp_path, old, new, needle):
    text = open(FQI_YAML).read()
    assert old in text
    p = tmp_path / "variant.yaml"
    p.write_text(text.replace(old, new))
    res = run(grlxd, ["-s", "1", "-q", str(p)], tmp_path)
    assert res.returncode == 1
    assert needle in res.stderr, res.stderr


def test_expressions_over_references(grlxd, tmp_path):
    """parser.cpp:49-134 on the values of referenced parameters: + - * element-wise with a scalar broadcast, ++ concatenation; a vector
    size mismatch is an error, as in the reference."""
    text = open(FQI_YAML).read()
    ok = text.replace("input_min: experiment/task/observation_min+experiment/task/action_min", "input_min: experiment/task/observation_min++experiment/task/action_min") \
             .replace("input_max: experiment/task/observation_max+experiment/task/action_max", "input_max: experiment/task/observation_max++experiment/task/action_max")
    p = tmp_path / "concat.yaml"
    p.write_text(ok)
    res = run(grlxd, ["-s", "1", "-q", str(p)], tmp_path)
    dumped = (tmp_path / "pendulum-fqi-ann.yaml").read_text()
    assert "input_min: [ 0, -37.699111843077517, -3 ]" in dumped and "deviation D4" not in res.stderr
    bad = text.replace("input_min: experiment/task/observation_min+experiment/task/action_min", "input_min: experiment/task/observation_min*[1, 2, 3]")
    p = tmp_path / "mismatch.yaml"
    p.write_text(bad)
    res = run(grlxd, ["-s", "1", "-q", str(p)], tmp_path)
    assert res.returncode == 1 and "vector size mismatch" in res.stderr


def test_deployer_multi_gpu_ranks_fail_cleanly_without_devices(tmp_path):
    """`grlxd -g 2` where no GPU exists (this suite runs without one): both ranks are forked before HIP starts, each fails to find its device
    and says so, the parent waits for both, removes the rendezvous file and returns 1 -- no hang, no fallback."""
    import shutil
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    from grl_amd import _build
    grlxd = _build.build_host()
    y = tmp_path / "p.yaml"
    shutil.copy(os.path.join(os.path.dirname(__file__), "golden", "pendulum-sarsa-tc.yaml"), y)
    res = subprocess.run([grlxd, "-g", "2", "-s", "1", "-q", str(y)], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert res.returncode == 1
    assert "2 of 2 ranks failed" in res.stderr + res.stdout and "multi-GPU" in res.stderr + res.stdout
    assert not list(tmp_path.glob("*.txt"))
