#!/bin/bash
# tools/clock_probe.sh <outdir>: GPU clocks / power while the cfg2 kernel runs for ~2 s
out=gpurun_out/$1; mkdir -p $out
(python bench.py --steps 40000 --warmup 50 --no-cpu > $out/bench.json 2>$out/err.log) &
BP=$!
sleep 2
for i in $(seq 1 40); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ';' >> $out/smi.txt; echo >> $out/smi.txt
  sleep 0.25
  kill -0 $BP 2>/dev/null || break
done
wait $BP
cat $out/smi.txt | cut -c1-400; python -c "import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
