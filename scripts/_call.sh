export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cholesky" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-300
