#!/bin/bash
# headline + per-class kernel times of the product library (or HSEFR_LIB), side legs off: the A/B view of a kernel change
python bench.py --steps ${1:-40} --no-cpu-baseline --no-pipeline --no-latency --no-config5 --no-other-configs --no-sustained 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], [(k['kernel'][:14], k['ms_per_step']) for k in l['kernels']])"
