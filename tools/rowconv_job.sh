python -m pytest tests/test_gpu_conv.py -x -q -k "band_kernel" 2>&1 | tail -5
for h in 16 8; do
python tools/time_conv.py 512 0 512 $h 0 2000 32 8 -1 1 1 | tail -1
python tools/time_conv.py 512 0 512 $h 0 0 0 8 0 1 1 | tail -1
python tools/time_conv.py 512 0 512 $h 0 2000 32 1 -1 1 1 | tail -1
python tools/time_conv.py 512 0 512 $h 0 0 0 1 0 1 1 | tail -1
done
