python -m pytest tests -m gpu -x -q 2>&1 | tail -5; python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(k) for k in d['kernels']]; print(d['all_kernels'])"
