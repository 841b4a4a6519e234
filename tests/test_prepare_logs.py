"""The reference's tests/test_prepare_logs.py:8-72 restated on this package's experiments/base/utils.py (host logic, no GPU): a new
experiment writes parameters.json; a second seed is accepted; the same seed again is refused once its returns exist; a run of the same
experiment name with a different agent parameter is refused (AssertionError in both cases, as the reference raises)."""
import json
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))


@pytest.mark.parametrize("algo", ["dqn", "isdqn", "tfdqn"])
def test_prepare_logs(tmp_path, algo):
    from experiments.base.utils import prepare_logs

    root = str(tmp_path)
    save_path = os.path.join(root, "lunar_lander", "exp_output", "_test_prepare_logs")
    base = ["--experiment_name", "_test_prepare_logs", "--disable_wandb"]
    prepare_logs("lunar_lander", algo, base + ["--seed", "1"], root=root)  # folders + parameters.json: no error
    os.makedirs(os.path.join(save_path, algo, "episode_returns_and_lengths"), exist_ok=True)
    json.dump({}, open(os.path.join(save_path, algo, "episode_returns_and_lengths", "1.json"), "w"))  # seed 1 has finished
    prepare_logs("lunar_lander", algo, base + ["--seed", "2"], root=root)  # another seed: no error
    with pytest.raises(AssertionError):  # the same seed again
        prepare_logs("lunar_lander", algo, base + ["--seed", "1"], root=root)
    parameters = json.load(open(os.path.join(save_path, "parameters.json")))
    assert set(parameters) == {"shared_parameters", algo} and "seed" not in parameters["shared_parameters"]
    name, value = [(k, v) for k, v in parameters[algo].items() if isinstance(v, int) and not isinstance(v, bool)][-1]
    with pytest.raises(AssertionError):  # same experiment, another value of an agent parameter
        prepare_logs("lunar_lander", algo, base + ["--seed", "3", f"--{name}", str(value + 1)], root=root)
    assert os.path.exists(os.path.join(save_path, "parameters.json"))
    shutil.rmtree(save_path)
