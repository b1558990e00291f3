mkdir -p gpurun_out/r6n
timeout 900 python -m pytest tests -m gpu -x -q -k "long_filters or exact_block or rccl or fuzz" > gpurun_out/r6n/pytest_sel.log 2>&1; tail -3 gpurun_out/r6n/pytest_sel.log
for r in "32000 48000 0.5" "64000 48000 0.5"; do set -- $r
python bench.py --src $1 --dst $2 --tb $3 --steps 60 --warmup 10 --settle 0 --no-cpu 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1->$2 tb $3', l['ms_per_step'], l['roofline']['kernel'], l['roofline']['avg_kernel_ms'])"
done
python bench.py --tb 10 --atten 109.56 --steps 200 --warmup 40 --settle 0 --no-cpu 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('tb10', l['ms_per_step'], l['roofline']['kernel'], l['roofline']['avg_kernel_ms'], l['roofline'].get('launches'))"
