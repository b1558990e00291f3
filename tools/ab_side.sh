#!/bin/bash
# tools/ab_side.sh <out tag> <variant specs for tools/ab.py ...> : the side configurations (convolver alone 2x up / 2x down,
# 48000 -> 32000, cfg5, polynomial, half-band + convolver) through tools/ab.py, two repetitions each
tag=$1; shift
for c in "up2 --src 44100 --dst 88200" "down2 --src 88200 --dst 44100" "r23 --src 48000 --dst 32000" "cfg5 --config cfg5" "poly --src 44100 --dst 44101" "hbdown --src 176400 --dst 44100" ${AB_EXTRA:+"$AB_EXTRA"}; do
  set -- $c; name=$1; shift
  if [ -n "$AB_ONLY" ] && ! echo " $AB_ONLY " | grep -q " $name "; then continue; fi
  echo "== $name"
  python tools/ab.py --out gpurun_out/ab_${tag}_$name --reps ${AB_REPS:-2} --steps 200 --bench-args "$*" "${VARIANTS[@]:-$@}" 2>&1 | tail -${AB_TAIL:-3} | tail -n +2
done
