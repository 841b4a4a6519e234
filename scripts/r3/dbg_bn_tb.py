import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "is-dqn_amd")
from tests.test_gpu_analysis_agents import _bn_pair
for B in (8, 16, 24, 32, 48, 64):
    hip, ora, (train, ev) = _bn_pair("cnn", 2, 4, B)
    eng = hip._engine
    g = hip.three_gradients(hip.params, hip.target_params, train)
    o = ora.three_gradients(ora.params, ora.target_params, train)
    for name, gg, oo in zip(("is", "tf", "tb"), g, o):
        got = eng.internal_to_flax_grads(gg)
        out = []
        for mod, leaf in (("BatchNorm_3", "bias"), ("LayerNorm_2", "bias"), ("LayerNorm_2", "scale"), ("Conv_2", "kernel"), ("Conv_0", "kernel")):
            a, b = np.asarray(got[mod][leaf], np.float64), oo[mod][leaf].numpy().astype(np.float64)
            out.append("%s/%s %.1e" % (mod, leaf, np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)))
        print(B, name, "  ".join(out))
