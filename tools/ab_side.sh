for args in "--src 44100 --dst 2822400 --block 1024 --channels 1024" "--src 44100 --dst 44101" "--src 48000 --dst 32000" "--src 176400 --dst 44100"; do tools/ab.sh --steps 200 --warmup 20 $args; done
