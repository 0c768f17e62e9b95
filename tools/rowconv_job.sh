python -m pytest tests/test_gpu_conv.py -x -q -k "rows_kernel" 2>&1 | tail -3
python tools/time_conv.py 64 0 64 256 0 1032 64 8 -1 1 0 | tail -1
python tools/time_conv.py 64 0 64 256 0 1032 64 8 -1 1 1 | tail -1
python tools/time_conv.py 64 0 64 256 0 1004 64 1 -1 1 1 | tail -1
python -m pytest tests/test_gpu_network.py tests/test_gpu_plans.py -x -q 2>&1 | tail -3
for v in normal large; do for rc in 0 1; do echo "== $v rowconv=$rc"; LSP_HIP_ROWCONV=$rc python bench.py --variant $v --batch 8 --dtype bf16 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(c['kernel'], c['launches'], c['ms'], c['us_per_launch'], c['gbs']) for c in d['roofline']['per_class']]"; done; done
