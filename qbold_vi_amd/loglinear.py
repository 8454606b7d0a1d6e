"""Mirror of the reference's loglinear.py: the log-linear (weighted least squares) qBOLD estimate
used as a comparator for the VI model.  `fit_wls` keeps the reference's name and return convention
(loglinear.py:68-105) but runs one closed-form HIP kernel over all voxels instead of a Python loop
with one sklearn `LinearRegression` per voxel ("takes roughly 1 min per volume", :75)."""
import configparser

import numpy as np
import torch

from .ops import Context


def _context(params, device):
    return Context(params, full_model=True, include_blood=True, device=device)


def fit_wls(signals, params=None, context=None, device=None, tau_min=0.016):
    """signals [subj, X, Y, Z, T] (or any [..., T]) -> (oef, dbv, r2p), each [..., 1], clipped to
    [0.01, 0.8], [0.002, 0.25], [1e-2, 100] (loglinear.py:101-104).  `params` is the INI [DEFAULT]
    section (read from ./config when omitted, as the reference's __main__ does, :108-127)."""
    if context is None:
        if params is None:
            cp = configparser.ConfigParser()
            cp.read('config')
            params = cp['DEFAULT']
        context = _context(params, device)
    x = torch.as_tensor(np.asarray(signals, dtype=np.float32) if not torch.is_tensor(signals) else signals,
                        dtype=torch.float32, device=context.device)
    lead = tuple(x.shape[:-1])
    out = context.wls_fit(x.reshape(-1, x.shape[-1]))
    return tuple(out[:, i].reshape(lead + (1,)) for i in range(3))


def save_predictions(predictions, filename, transform_directory=None):
    """loglinear.save_predictions (:13-65) without the FSL MNI step: `<filename>_{oef,dbv,r2p}.nii.gz`,
    each [X, Y, Z, subj]."""
    import os
    from . import nifti
    template = None
    if transform_directory is not None and os.path.isfile(os.path.join(transform_directory, 'example.nii.gz')):
        template = nifti.load(os.path.join(transform_directory, 'example.nii.gz'))[1]
    for im, suffix in zip(predictions, ('_oef', '_dbv', '_r2p')):
        im = im.detach().cpu().numpy() if torch.is_tensor(im) else np.asarray(im)
        images = np.concatenate(np.split(im, im.shape[0], axis=0), axis=-1)[0]
        nifti.save(images.astype(np.float32), filename + suffix + '.nii.gz', template)
