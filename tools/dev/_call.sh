python -m pytest tests/test_gpu_guard_pages.py -x -q -m gpu 2>&1 | tail -6
