#!/usr/bin/env python3
"""Drop-in for the reference's `python qbold_train_model.py <config.yaml>`
(qbold_train_model.py:326-337): utils.load_arguments() -> ModelTrainer(args).train_model()."""
import sys

from qbold_vi_amd.training import ModelTrainer
from qbold_vi_amd.utils import load_arguments

if __name__ == '__main__':
    args = load_arguments(sys.argv)
    model_trainer = ModelTrainer(args)
    model_trainer.train_model()
