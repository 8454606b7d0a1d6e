#!/usr/bin/env python3
"""Drop-in for the reference's `python train.py configurations/optimal.yaml [-d DATA -f SYNTH]`
(train.py:454-491): same positional YAML / argparse flags / INI `config` in the CWD.  New flags:
--synthetic_voxels N (fine-tune on N synthetic voxels instead of the real .npy volumes),
--mc_samples S, --devices G (launch with torchrun for G > 1)."""
import sys

import numpy as np
import torch

from qbold_vi_amd.training import train_model
from qbold_vi_amd.utils import load_arguments

if __name__ == '__main__':
    torch.manual_seed(1)   # tf.random.set_seed(1); np.random.seed(1)  (train.py:458-459)
    np.random.seed(1)
    args = load_arguments(sys.argv, entry="train")
    args.pop("use_wandb", None)  # metrics go to <save_directory>/metrics.jsonl, no network
    train_model(args)
