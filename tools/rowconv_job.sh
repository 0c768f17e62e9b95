python -m pytest tests/test_gpu_conv.py -x -q -k "band_kernel" 2>&1 | tail -4
for h in 4 2; do
python tools/time_conv.py 512 0 512 $h 0 2000 32 8 -1 1 1 | tail -1
python tools/time_conv.py 512 0 512 $h 0 0 0 8 0 1 1 | tail -1
done
python -m pytest tests/test_gpu_network.py tests/test_gpu_plans.py -x -q 2>&1 | tail -2
for v in "normal 8" "large 8"; do set -- $v; for mf in 99 8 99 8; do echo "== $1 b$2 small-level bandconv from $mf frames"; LSP_HIP_BANDCONV_MIN_FRAMES=$mf python bench.py --variant $1 --batch $2 --dtype bf16 --no-cpu-baseline --no-extra --steps 100 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [(c['kernel'],c['launches'],c['us_per_launch']) for c in d['roofline']['per_class'] if 'band' in c['kernel'] or 'reduce' in c['kernel'] or '64x64' in c['kernel'] or 'g4' in c['kernel']])"; done; done
