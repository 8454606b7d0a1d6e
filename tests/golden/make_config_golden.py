"""Generates tests/golden/merged_config_optimal.json by running the REFERENCE's own
utils.load_arguments() (pure argparse + PyYAML; it imports without TensorFlow) on the reference's
configurations/optimal.yaml.  Run in the build container only (/root/reference is not on the GPU
box); the JSON it writes is the committed fixture."""
import json
import os
import sys

REF = "/root/reference"
sys.path.insert(0, REF)
import utils as ref_utils  # noqa: E402

sys.argv = ["qbold_train_model.py", os.path.join(REF, "configurations", "optimal.yaml")]
args = ref_utils.load_arguments()
out = {k: {"value": v, "type": type(v).__name__} for k, v in sorted(args.items())}
here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "merged_config_optimal.json"), "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
print(len(out), "keys")
