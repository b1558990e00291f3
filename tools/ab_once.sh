python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for c in "up2 --src 44100 --dst 88200" "down2 --src 88200 --dst 44100" "r23 --src 48000 --dst 32000" "cfg5 --config cfg5" "poly --src 44100 --dst 44101" "hbdown --src 176400 --dst 44100"; do
  set -- $c; name=$1; shift
  python tools/ab.py --out gpurun_out/ab_once_$name --reps 2 --steps 200 --bench-args "$*" park1 park0:opt=park=0 2>&1 | tail -2
done
