#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the two configs whose BVH fits in LDS (configs[1] weekend-final, configs[4] mixed), after the
# LDS-resident extend variant.  Afterwards, here: for t in weekend_final mixed; do python tools/summarise_prof.py $t round3_$t; done
set -u
C="--steps 1 --warmup 0 --no-cpu-baseline"
tools/profile.sh weekend_final $C --scene weekend-final --width 1920 --spp-per-gpu 256 || exit 1
tools/profile.sh mixed $C --scene weekend-mixed --width 900 --spp-per-gpu 4096 || exit 1
