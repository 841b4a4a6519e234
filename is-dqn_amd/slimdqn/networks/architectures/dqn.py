"""Architecture descriptor with the attribute names of the reference's Flax module
(slimdqn/networks/architectures/dqn.py:39-45).  The forward pass itself is the HIP engine
(csrc/net_kernels.hip); this class only carries the configuration and validates the scope:

  * "cnn": Conv 8x8/4 -> Conv 4x4/2 -> Conv 3x3/1, each SAME-padded, + LayerNorm(channels) + ReLU,
           flatten (h, w, c), then Dense -> LayerNorm -> ReLU per remaining feature, final Dense  (:48-74, :93-103)
  * "fc":  Dense -> LayerNorm -> ReLU per feature, final Dense                                      (:89-103)
  * "impala": three Stacks (Conv 3x3 -> max_pool 3x3/2 SAME -> two residual blocks [LayerNorm] -> ReLU -> Conv -> ReLU -> Conv -> +),
           [LayerNorm] -> ReLU -> flatten, then the same Dense tail                                  (:7-36, :75-88; csrc/impala.h)
  * batch_norm: flax.linen.BatchNorm behind the input scaling and behind every hidden ReLU -- ``axis=(1, 2)`` on image tensors
           (one statistic per pixel position, over batch and channels), per feature behind the flatten and the Dense layers
           (:52-53, 59-60, 66-67, 73-74, 100-101), and inside every residual block of the impala Stacks (:29-30; csrc/batchnorm.h).
"""
from typing import Sequence


class DQNNet:
    def __init__(self, features: Sequence[int], architecture_type: str, final_feature: int, layer_norm: bool = False,
                 batch_norm: bool = False):
        if architecture_type not in ("cnn", "impala", "fc"):
            raise NotImplementedError(f"architecture_type={architecture_type!r}: 'cnn', 'impala' or 'fc'")
        if architecture_type in ("cnn", "impala") and len(features) < 3:
            raise ValueError("cnn needs at least the three convolution widths")
        self.features = [int(f) for f in features]
        self.architecture_type = architecture_type
        self.final_feature = int(final_feature)
        self.layer_norm = bool(layer_norm)
        self.batch_norm = bool(batch_norm)
