#!/usr/bin/env python3
"""Entry point mirroring the reference's loglinear.py __main__ (loglinear.py:107-158): estimate
OEF / DBV / R2' maps with the log-linear WLS fit and write them as NIfTI.  The reference hard-codes
its data directory and the streamlined protocol's tau grid; both are arguments here with the
reference's values as defaults."""
import argparse
import configparser
import os

import numpy as np

if __name__ == '__main__':
    config = configparser.ConfigParser()
    config.read('config')
    params = config['DEFAULT']
    parser = argparse.ArgumentParser(description='Estimate parameters using log-linear method')
    parser.add_argument('-f', default='streamlined_ase.npy', help='signal data file inside the data directory')
    parser.add_argument('-d', default='/home/data/qbold/', help='data directory')
    parser.add_argument('-o', default='wls_clip', help='output directory')
    parser.add_argument('--tau_start', default='-0.028')
    parser.add_argument('--tau_step', default='0.004')
    args = parser.parse_args()

    from qbold_vi_amd.loglinear import fit_wls, save_predictions
    params = dict(params, tau_start=args.tau_start, tau_step=args.tau_step)
    os.makedirs(args.o, exist_ok=True)
    data = np.load(os.path.join(args.d, args.f))
    oef, dbv, r2p = fit_wls(data[:, :, :, :, :-2], params=params)
    save_predictions([oef, dbv, r2p], os.path.join(args.o, os.path.splitext(args.f)[0].replace('_ase', '')),
                     transform_directory=None)
