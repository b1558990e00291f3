mkdir -p gpurun_out/r6final
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6final/smoke.log 2>&1; tail -1 gpurun_out/r6final/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6final/bench_driver.json 2> gpurun_out/r6final/bench_driver.err; echo rc=$?; cut -c1-700 gpurun_out/r6final/bench_driver.json
timeout 180 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu > gpurun_out/r6final/bench_g2.out 2>&1; echo "gpus2 rc=$?"; tail -3 gpurun_out/r6final/bench_g2.out | cut -c1-300
