#!/bin/bash
# the three grid configs of tools/profile_round3.sh alone (re-run after the grid kernels changed)
set -u
C="--steps 1 --warmup 0 --no-cpu-baseline"
tools/profile.sh res4_grid5 $C --scene dragon_res4 --width 900 --spp-per-gpu 50 --algorithm grid --grid-size 5 || exit 1
tools/profile.sh res4_grid1 $C --scene dragon_res4 --width 900 --spp-per-gpu 50 --algorithm grid --grid-size 1 || exit 1
tools/profile.sh res4_grid25 $C --scene dragon_res4 --width 900 --spp-per-gpu 50 --algorithm grid --grid-size 25 || exit 1
