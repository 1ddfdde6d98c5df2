python -m pytest tests/test_gpu_select_audit.py -x -q -m gpu -k "every_shape" 2>&1 | tail -12
