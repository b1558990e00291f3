python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for c in "up2 --src 44100 --dst 88200" "down2 --src 88200 --dst 44100" "r23 --src 48000 --dst 32000" "r13 --src 96000 --dst 32000" "r31 --src 44100 --dst 132300"; do
  set -- $c; name=$1; shift
  echo "== $name"
  python tools/ab.py --out gpurun_out/ab_lean_$name --reps 2 --steps 200 --bench-args "$*" new prev:lib=variants/prev.so 2>&1 | tail -2
done
