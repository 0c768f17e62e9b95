python -m pytest tests/test_gpu_conv.py -x -q -k "rows_kernel" 2>&1 | tail -3
python tools/time_conv.py 128 0 128 128 0 1016 128 8 -1 1 0 | tail -1
python tools/time_conv.py 128 0 128 128 0 1016 128 8 -1 1 1 | tail -1
python tools/time_conv.py 128 0 128 128 0 128 128 8 0 1 0 | tail -1
python tools/time_conv.py 128 0 128 128 0 128 128 8 0 1 1 | tail -1
python tools/time_conv.py 128 0 128 128 0 1002 128 1 -1 1 1 | tail -1
python tools/time_conv.py 128 0 128 128 0 128 128 1 0 1 1 | tail -1
