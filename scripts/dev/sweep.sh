#!/bin/bash
# A/B timing of build-time options on the GPU box: each argument is one flag set; rebuilds ctx / vi / elbo kernels with
# it (variant3.sh), times the headline and the 24-tau protocol in one process each.  scripts/dev/sweep.sh "" "-DQB_GT_DEPTH=4" ...
cd $GRAFT_REPO_ROOT
for fl in "$@"; do
  bash scripts/dev/variant3.sh "-DQB_VI_PROBE $fl" "python scripts/dev/time_headline.py 0 2>&1 | grep 'sel' | tail -2 | tr '\n' ' '; python scripts/dev/time_headline.py 0 p24 2>&1 | grep sel | tail -2 | tr '\n' ' '; echo"
done
