"""Architecture descriptor with the attribute names of the reference's Flax module
(slimdqn/networks/architectures/dqn.py:39-45).  The forward pass itself is the HIP engine
(csrc/net_kernels.hip); this class only carries the configuration and validates the scope:

  * "cnn": Conv 8x8/4 -> Conv 4x4/2 -> Conv 3x3/1, each SAME-padded, + LayerNorm(channels) + ReLU,
           flatten (h, w, c), then Dense -> LayerNorm -> ReLU per remaining feature, final Dense  (:48-74, :93-103)
  * "fc":  Dense -> LayerNorm -> ReLU per feature, final Dense                                      (:89-103)
  * "impala" and BatchNorm variants are outside the hot-path scope and raise.
"""
from typing import Sequence


class DQNNet:
    def __init__(self, features: Sequence[int], architecture_type: str, final_feature: int, layer_norm: bool = False,
                 batch_norm: bool = False):
        if architecture_type not in ("cnn", "fc"):
            raise NotImplementedError(f"architecture_type={architecture_type!r} is outside the hot-path scope")
        if batch_norm:
            raise NotImplementedError("batch_norm is outside the hot-path scope")
        if architecture_type == "cnn" and len(features) < 3:
            raise ValueError("cnn needs at least the three convolution widths")
        self.features = [int(f) for f in features]
        self.architecture_type = architecture_type
        self.final_feature = int(final_feature)
        self.layer_norm = bool(layer_norm)
        self.batch_norm = bool(batch_norm)
