"""slimdqn -- MI355X-native drop-in for the hot path of theovincent/iS-DQN.

Same import paths and public names as the reference package (slimdqn.sample_collection.*,
slimdqn.networks.isdqn.iSDQN); the arithmetic runs in hand-written HIP kernels behind the C ABI
of include/isdqn_hip.h.  There is no CPU fallback.
"""
