#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + PMC passes (tools/profile.sh; one step per pass) of the default bench
# workload and of the BASELINE configs that are not the headline: configs[1] weekend-final, configs[2] res4 (bvh, grid 25 / 5 / 1),
# configs[4] mixed.  Afterwards, here: for t in default weekend_final ...; do python tools/summarise_prof.py $t round3_$t; done
set -u
C="--steps 1 --warmup 0 --no-cpu-baseline"
tools/profile.sh default $C || exit 1
tools/profile.sh weekend_final $C --scene weekend-final --width 1920 --spp-per-gpu 256 || exit 1
tools/profile.sh res4_bvh $C --scene dragon_res4 --width 900 --spp-per-gpu 50 || exit 1
tools/profile.sh res4_grid5 $C --scene dragon_res4 --width 900 --spp-per-gpu 50 --algorithm grid --grid-size 5 || exit 1
tools/profile.sh res4_grid1 $C --scene dragon_res4 --width 900 --spp-per-gpu 50 --algorithm grid --grid-size 1 || exit 1
tools/profile.sh res4_grid25 $C --scene dragon_res4 --width 900 --spp-per-gpu 50 --algorithm grid --grid-size 25 || exit 1
tools/profile.sh mixed $C --scene weekend-mixed --width 900 --spp-per-gpu 4096 || exit 1
