python -m pytest tests/test_gpu_conv.py -x -q -k "band_kernel" 2>&1 | tail -3
python tools/time_conv.py 512 0 512 32 0 2000 32 8 -1 1 1 | tail -1
python tools/time_conv.py 512 0 512 32 0 0 0 8 0 1 1 | tail -1
for v in "normal 8" "large 8"; do set -- $v; for mw in 16 32 16 32; do echo "== $1 b$2 bandconv max width $mw"; LSP_HIP_BANDCONV_MAX_WIDTH=$mw python bench.py --variant $1 --batch $2 --dtype bf16 --no-cpu-baseline --no-extra --steps 100 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [(c['kernel'],c['launches'],c['us_per_launch']) for c in d['roofline']['per_class'] if 'band' in c['kernel'] or '128x128' in c['kernel']])"; done; done
