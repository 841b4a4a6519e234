"""Command-line flags of the reference (experiments/base/parser_argument.py:27-156, 243-248), table-driven.

Every flag keeps its short name, long name, type and default so that launch scripts written for the reference
work unchanged.  ``add_base_arguments`` / ``add_isdqn_arguments`` return the list of long names they added,
which ``store_params`` uses to split ``parameters.json`` into shared and per-algorithm sections.
"""
import argparse
from typing import List

# (short, long, kwargs)
_BASE = [
    ("-en", "--experiment_name", dict(type=str, required=True, help="Experiment name.")),
    ("-s", "--seed", dict(type=int, required=True, help="Seed of the experiment.")),
    ("-dw", "--disable_wandb", dict(action="store_true", default=False, help="Disable wandb.")),
    ("-f", "--features", dict(type=int, nargs="*", default=[100, 100], help="List of features for the Q-networks.")),
    ("-rbc", "--replay_buffer_capacity", dict(type=int, default=10_000, help="Replay Buffer capacity.")),
    ("-bs", "--batch_size", dict(type=int, default=32, help="Batch size for training.")),
    ("-n", "--update_horizon", dict(type=int, default=1, help="Value of n in n-step TD update.")),
    ("-gamma", "--gamma", dict(type=float, default=0.99, help="Discounting factor.")),
    ("-lr", "--learning_rate", dict(type=float, default=3e-4, help="Learning rate.")),
    ("-horizon", "--horizon", dict(type=int, default=1_000, help="Horizon for truncation.")),
    ("-at", "--architecture_type", dict(type=str, default="fc", choices=["cnn", "impala", "fc"], help="Type of architecture.")),
    ("-ne", "--n_epochs", dict(type=int, default=50, help="Number of epochs to perform.")),
    ("-ntspe", "--n_training_steps_per_epoch", dict(type=int, default=10_000, help="Number of training steps per epoch.")),
    ("-utd", "--data_to_update", dict(type=float, default=1, help="Number of data points to collect per online Q-network update.")),
    ("-nis", "--n_initial_samples", dict(type=int, default=1_000, help="Number of initial samples before the training starts.")),
    ("-ee", "--epsilon_end", dict(type=float, default=0.01, help="Ending value for the linear decaying epsilon used for exploration.")),
    ("-ed", "--epsilon_duration", dict(type=float, default=1_000, help="Duration of epsilon's linear decay used for exploration.")),
    ("-a", "--analysis", dict(action="store_true", default=False, help="Flag to run analysis with the algorithm (srank and dormant neurons).")),
]
_ISDQN = [
    ("-nbi", "--n_bellman_iterations", dict(type=int, default=3, help="Number of bellman iterations to train in parallel. (K)")),
    ("-ln", "--layer_norm", dict(action="store_true", default=False, help="Flag to add layer norm.")),
    ("-bn", "--batch_norm", dict(action="store_true", default=False, help="Flag to add batch norm.")),
    ("-tuf", "--target_update_frequency", dict(type=int, default=200, help="Number of training steps before updating the target Q-network. (T)")),
]
_DQN = [_ISDQN[1], _ISDQN[3]]             # add_dqn_arguments: layer_norm, target_update_frequency (parser_argument.py:229-232)
_TFDQN = [_ISDQN[1], _ISDQN[2], _ISDQN[3]]  # add_tfdqn_arguments: layer_norm, batch_norm, target_update_frequency (:235-239)
# extras of this build (not in the reference): kept out of parameters.json comparisons by living in their own group
_ENGINE = [
    ("-prec", "--precision", dict(type=str, default="bf16x3", choices=["bf16x3", "bf16"], help="MFMA precision of the HIP engine.")),
    ("-per", "--prioritized", dict(action="store_true", default=False, help="Prioritized replay (sum-tree on the GPU) with TD-error writeback.")),
    ("-hd", "--huber_delta", dict(type=float, default=0.0, help="0: squared TD error (the reference's loss); > 0: Huber loss with this delta.")),
    ("-nenvs", "--n_envs", dict(type=int, default=1, help="Host environments stepped in lockstep with one batched best_actions forward (1 = the reference's loop).")),
    ("-nworkers", "--n_env_workers", dict(type=int, default=0, help="Host worker processes stepping the -nenvs environments in parallel (0 = in this process).")),
    ("-env", "--env_backend", dict(type=str, default="ale", choices=["ale", "synthetic"], help="'synthetic' replaces ALE by random frames (no ROMs needed).")),
]


def _add(parser: argparse.ArgumentParser, table) -> List[str]:
    for short, long, kw in table:
        parser.add_argument(short, long, **kw)
    return [long.lstrip("-") for _, long, _ in table]


def add_base_arguments(parser: argparse.ArgumentParser) -> List[str]:
    return _add(parser, _BASE)


def add_isdqn_arguments(parser: argparse.ArgumentParser) -> List[str]:
    return _add(parser, _ISDQN)


def add_dqn_arguments(parser: argparse.ArgumentParser) -> List[str]:
    return _add(parser, _DQN)


def add_tfdqn_arguments(parser: argparse.ArgumentParser) -> List[str]:
    return _add(parser, _TFDQN)


def add_analysisdqn_arguments(parser: argparse.ArgumentParser) -> List[str]:
    return _add(parser, _ISDQN)  # (parser_argument.py: the analysis agents take their base agents' flags)


def add_analysistfdqn_arguments(parser: argparse.ArgumentParser) -> List[str]:
    return _add(parser, _TFDQN)


def add_engine_arguments(parser: argparse.ArgumentParser) -> List[str]:
    return _add(parser, _ENGINE)
