"""Copy the reference's AGPR checkpoint *data* (checkpoints/<DATASET>/SRF_<s>/{train_x,train_y,model}.pt)
into small .npz fixtures.  Run in the build container only:
    python tests/golden/make_agpr_goldens.py
Loaded with torch.load(weights_only=True): nothing from the files is executed.
raw = [raw_noise, raw_constant, raw_outputscale, raw_variance].
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/checkpoints"

for name in ["BSD100/SRF_2", "BSD100/SRF_3", "Set14/SRF_3"]:
    ld = lambda f: torch.load(f"{REF}/{name}/{f}.pt", weights_only=True, map_location="cpu")
    X, y, sd = ld("train_x"), ld("train_y"), ld("model")
    raw = np.array([float(sd["likelihood.noise_covar.raw_noise"].reshape(-1)[0]),
                    float(sd["mean_module.raw_constant"].reshape(-1)[0]),
                    float(sd["covar_module.raw_outputscale"].reshape(-1)[0]),
                    float(sd["covar_module.base_kernel.raw_variance"].reshape(-1)[0])])
    keys = np.array(sorted(sd.keys()))
    out = os.path.join(HERE, "agpr_" + name.replace("/", "_") + ".npz")
    np.savez_compressed(out, train_x=X.numpy(), train_y=y.numpy(), raw=raw, state_dict_keys=keys)
    print(out, X.shape, y.shape, raw)
