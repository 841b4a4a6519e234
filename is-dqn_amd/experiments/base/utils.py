"""Experiment bookkeeping with the reference's output layout (experiments/base/utils.py:12-144):

    experiments/<env>/exp_output/<experiment_name>/parameters.json          shared + per-algorithm flags
    experiments/<env>/exp_output/<experiment_name>/<algo>/episode_returns_and_lengths/<seed>.json
    experiments/<env>/exp_output/<experiment_name>/<algo>/models/<seed>     pickle of {"params": flax pytree}

wandb is not installed offline: a no-op logger with the same ``.log`` method stands in (``-dw`` is implied).
The duplicate-seed guard checks the file that is actually written (``<seed>.json``); the reference looks for
``<seed>.npy`` (utils.py:48 vs :125), which can never exist.
"""
import argparse
import json
import os
import pickle
import time
from typing import List

from experiments.base import parser_argument


class _NullLogger:
    def __init__(self):
        self.history = []

    def log(self, d):
        self.history.append(dict(d))


def prepare_logs(env_name: str, algo_name: str, argvs: List[str], root: str = None):
    print(f"---- Train {algo_name} on {env_name} {time.strftime('%d-%m-%Y %H:%M:%S')} ----", flush=True)
    parser = argparse.ArgumentParser(f"Train {algo_name} on {env_name}.")
    shared = parser_argument.add_base_arguments(parser)
    agent = getattr(parser_argument, f"add_{algo_name}_arguments")(parser)
    parser_argument.add_engine_arguments(parser)
    p = vars(parser.parse_args(argvs))
    p["env_name"] = env_name
    if env_name == "atari":
        p["game_name"] = p["experiment_name"].split("_")[-1]
    p["algo_name"] = algo_name
    root = root or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    p["save_path"] = os.path.join(root, env_name, "exp_output", p["experiment_name"], algo_name)
    check_experiment(p)
    store_params(p, shared, agent)
    p["wandb"] = _NullLogger()
    return p


def check_experiment(p: dict):
    returns_path = os.path.join(p["save_path"], "episode_returns_and_lengths", f"{p['seed']}.json")
    model_path = os.path.join(p["save_path"], "models", str(p["seed"]))
    assert not (os.path.exists(returns_path) or os.path.exists(model_path)), (
        "Same algorithm with same seed results already exists. Delete them and restart, or change the experiment name."
    )
    params_path = os.path.join(os.path.dirname(p["save_path"]), "parameters.json")
    if not os.path.exists(params_path):
        return
    try:
        stored = json.load(open(params_path))
    except json.JSONDecodeError:  # a sibling seed is still writing the file
        return
    known = dict(stored.get("shared_parameters", {}))
    known.update(stored.get(p["algo_name"], {}))
    for name, value in p.items():
        if name in known:
            assert known[name] == value, (
                f"The same experiment has been run with {name} = {known[name]} instead of {value}. Change the experiment name."
            )


def store_params(p: dict, shared_params: List[str], agent_params: List[str]):
    os.makedirs(p["save_path"], exist_ok=True)
    params_path = os.path.join(os.path.dirname(p["save_path"]), "parameters.json")
    stored = {}
    if os.path.exists(params_path):
        for _ in range(200):
            try:
                stored = json.load(open(params_path))
                break
            except json.JSONDecodeError:
                time.sleep(0.01)
    if "shared_parameters" not in stored:
        stored["shared_parameters"] = {k: p[k] for k in shared_params if k not in ("seed", "disable_wandb")}
    if p["algo_name"] not in stored:
        stored[p["algo_name"]] = {k: p[k] for k in agent_params}
    ordered = {"shared_parameters": stored["shared_parameters"]}
    for k in sorted(k for k in stored if k != "shared_parameters"):
        ordered[k] = stored[k]
    json.dump(ordered, open(params_path, "w"), indent=4)


def save_data(p: dict, episode_returns: list, episode_lengths: list, model, analysis_logs=None):
    ret_dir = os.path.join(p["save_path"], "episode_returns_and_lengths")
    model_dir = os.path.join(p["save_path"], "models")
    os.makedirs(ret_dir, exist_ok=True)
    os.makedirs(model_dir, exist_ok=True)
    json.dump(
        {"episode_lengths": episode_lengths, "episode_returns": episode_returns},
        open(os.path.join(ret_dir, f"{p['seed']}.json"), "w"),
        indent=4,
    )
    if model is not None:
        pickle.dump(model, open(os.path.join(model_dir, str(p["seed"])), "wb"))
    if p.get("analysis"):  # utils.py:137-144 of the reference: analysis/<seed>.json
        os.makedirs(os.path.join(p["save_path"], "analysis"), exist_ok=True)
        json.dump(analysis_logs, open(os.path.join(p["save_path"], "analysis", f"{p['seed']}.json"), "w"), indent=4)
