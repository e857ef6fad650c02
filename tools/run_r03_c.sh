#!/bin/bash
# GPU box: the round-end checks (tools/run_final.sh), then the measured files of profiles/r03 (tools/make_profiles.sh)
set -u
cd $GRAFT_REPO_ROOT
bash tools/run_final.sh || exit 1
bash tools/make_profiles.sh r03
