"""Command-line / YAML configuration with the reference's schema and merge rules.

Mirrors utils.py of the reference (setup_argparser :4-43, get_defaults :47-83, load_arguments
:86-123) and the inline copy in train.py (:107-186, :461-480): same flag names, types and
defaults; YAML values override the argparse result with `type(default)(value)` when the current
value is truthy and verbatim otherwise (so PyYAML's string "2e-3" becomes a float only because the
default is a non-zero float -- SURVEY Appendix B6).  New, defaulted-off flags for the synthetic
voxel mode are added at the end.
"""
import argparse
import sys

_FLAG_TYPES = [
    ("no_units", int), ("no_pt_epochs", int), ("no_ft_epochs", int), ("student_t_df", int),
    ("crop_size", int), ("no_intermediate_layers", int), ("kl_weight", float),
    ("smoothness_weight", float), ("pt_lr", float), ("ft_lr", float), ("dropout_rate", float),
    ("im_loss_sigma", float), ("use_layer_norm", bool), ("use_r2p_loss", bool),
    ("multi_image_normalisation", bool), ("activation", None), ("misalign_prob", float),
    ("use_blood", bool), ("channelwise_gating", bool), ("full_model", bool),
    ("use_population_prior", bool), ("inv_gamma_alpha", float), ("inv_gamma_beta", float),
    ("gate_offset", float), ("resid_init_std", float), ("infer_inv_gamma", bool), ("use_mvg", bool),
    ("uniform_prop", float), ("use_swa", bool), ("adamw_decay", float), ("pt_adamw_decay", float),
    ("predict_log_data", bool),
]


def get_defaults(entry="qbold_train_model"):
    """utils.get_defaults (utils.py:47-83).  entry='train' gives train.py's variant (:150-186),
    which differs in use_population_prior (False) and has use_wandb instead of wandb_project."""
    d = dict(no_units=30, no_intermediate_layers=1, student_t_df=2, pt_lr=5e-5, ft_lr=5e-3,
             kl_weight=1.0, smoothness_weight=1.0, dropout_rate=0.0, no_pt_epochs=5, no_ft_epochs=40,
             im_loss_sigma=0.08, crop_size=16, use_layer_norm=False, activation='relu',
             use_r2p_loss=False, multi_image_normalisation=True, full_model=True, use_blood=True,
             misalign_prob=0.0, use_population_prior=True, wandb_project='', inv_gamma_alpha=0.0,
             inv_gamma_beta=0.0, gate_offset=0.0, resid_init_std=1e-1, channelwise_gating=True,
             infer_inv_gamma=False, use_mvg=False, uniform_prop=0.1, use_swa=True, adamw_decay=2e-4,
             pt_adamw_decay=2e-4, predict_log_data=True)
    if entry == "train":
        d["use_population_prior"] = False
        del d["wandb_project"]
        d["use_wandb"] = True
    return d


def setup_argparser(defaults_dict):
    p = argparse.ArgumentParser(description='Train neural network for parameter estimation')
    p.add_argument('-d', default='/home/data/qbold/', help='path to the real data directory')
    p.add_argument('-f', default='synthetic_data.npz', help='path to synthetic data file')
    for name, typ in _FLAG_TYPES:
        if typ is None:
            p.add_argument('--' + name, default=defaults_dict[name])
        else:
            p.add_argument('--' + name, type=typ, default=defaults_dict[name])
    p.add_argument('--save_directory', default=None)
    if 'wandb_project' in defaults_dict:
        p.add_argument('--wandb_project', default=defaults_dict['wandb_project'])
    if 'use_wandb' in defaults_dict:
        p.add_argument('--use_wandb', type=bool, default=defaults_dict['use_wandb'])
    # additions of this implementation (all off by default)
    p.add_argument('--synthetic_voxels', type=int, default=0,
                   help='fine-tune / evaluate on N synthetic voxels instead of the real .npy volumes')
    p.add_argument('--mc_samples', type=int, default=1, help='likelihood draws per voxel (no_samples)')
    p.add_argument('--devices', type=int, default=1, help='GPUs (one process each, torchrun)')
    return p


def merge_yaml(args, opt):
    """The reference's override loop (utils.py:109-116 = train.py:473-480)."""
    for key, val in opt.items():
        if args.get(key):
            args[key] = type(args.get(key))(val)
        else:
            args[key] = val
    return args


def load_arguments(argv=None, entry="qbold_train_model"):
    """utils.load_arguments (utils.py:86-123).  argv defaults to sys.argv; a first argument
    containing '.yaml' is the configuration file, the rest goes to argparse.  (The reference
    raises UnboundLocalError when no YAML is given -- Appendix B7; here the YAML is optional.)"""
    import yaml
    argv = list(sys.argv if argv is None else argv)
    yaml_file = None
    if len(argv) >= 2 and ".yaml" in argv[1]:
        yaml_file = argv[1]
        argv = [argv[0]] + argv[2:]   # the reference drops everything else; extra flags are allowed here
    args = vars(setup_argparser(get_defaults(entry)).parse_args(argv[1:]))
    if yaml_file is not None:
        with open(yaml_file) as fh:
            merge_yaml(args, yaml.load(fh, Loader=yaml.FullLoader))
    return args
