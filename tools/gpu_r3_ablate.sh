#!/bin/bash
# (1) what the team waits cost: FFT_HIP_TEAM_ABLATE=8 skips the polls; (2) what a smaller window would be worth: gpu_r3_window.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/gpu_r3_nopoll.sh || exit 1
bash $R/tools/gpu_r3_window.sh || exit 1
