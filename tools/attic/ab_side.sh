#!/bin/bash
# tools/ab_side.sh <out tag> <variant specs for tools/ab.py ...> : the side configurations (convolver alone 2x up / 2x down,
# 48000 -> 32000, cfg5, polynomial, half-band + convolver) through tools/ab.py, AB_REPS (2) repetitions each.
#   tools/ab_side.sh once park1 park0:opt=park=0
#   AB_ONLY="r23 up2" tools/ab_side.sh lean new prev:lib=variants/prev.so
tag=$1; shift
variants=("$@")
for c in "up2 --src 44100 --dst 88200" "down2 --src 88200 --dst 44100" "r23 --src 48000 --dst 32000" "cfg5 --config cfg5" \
         "poly --src 44100 --dst 44101" "hbdown --src 176400 --dst 44100"; do
  name=${c%% *}; args=${c#* }
  if [ -n "$AB_ONLY" ] && ! echo " $AB_ONLY " | grep -q " $name "; then continue; fi
  echo "== $name"
  python tools/ab.py --out gpurun_out/ab_${tag}_$name --reps ${AB_REPS:-2} --steps 200 --bench-args "$args" "${variants[@]}" 2>&1 | tail -$((${#variants[@]} + 1))
done
