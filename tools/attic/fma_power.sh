#!/bin/bash
# tools/fma_power.sh <outdir>: clock and power under a pure fp64 FMA load (tools/ubench/fma_power.hip)
out=gpurun_out/$1; mkdir -p $out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/fma_power.hip -o $out/fma_power || exit 1
($out/fma_power > $out/fma.txt 2>&1) &
BP=$!
for i in $(seq 1 40); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ';' >> $out/smi.txt; echo >> $out/smi.txt
  sleep 0.2
  kill -0 $BP 2>/dev/null || break
done
wait $BP
cat $out/fma.txt; cat $out/smi.txt
