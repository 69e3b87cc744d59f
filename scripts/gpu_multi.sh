export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | tail -8
