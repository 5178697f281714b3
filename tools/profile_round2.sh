#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + PMC passes (tools/profile.sh) of the BASELINE configs that are
# not the headline bench: configs[1] weekend-final, configs[2] res4 (bvh, grid 5, grid 1, grid 25), configs[4] mixed.
# Afterwards, here: for t in ...; do python tools/summarise_prof.py $t round2_$t; done
set -u
C="--steps 1 --warmup 1 --no-cpu-baseline --no-pmc"
tools/profile.sh weekend_final $C --scene weekend-final --width 1920 --spp-per-gpu 256 || exit 1
tools/profile.sh res4_bvh $C --scene dragon_res4 --width 900 --spp-per-gpu 50 || exit 1
tools/profile.sh res4_grid5 $C --scene dragon_res4 --width 900 --spp-per-gpu 50 --algorithm grid --grid-size 5 || exit 1
tools/profile.sh res4_grid1 $C --scene dragon_res4 --width 900 --spp-per-gpu 50 --algorithm grid --grid-size 1 || exit 1
tools/profile.sh res4_grid25 $C --scene dragon_res4 --width 900 --spp-per-gpu 50 --algorithm grid --grid-size 25 || exit 1
tools/profile.sh mixed $C --scene weekend-mixed --width 900 --spp-per-gpu 4096 || exit 1
