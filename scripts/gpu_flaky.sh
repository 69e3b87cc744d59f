#!/bin/bash
# the whole GPU parity suite several times in a row (the driver runs it once with -x: a test that fails one time in twenty matters)
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_rep$i.log 2>&1
  echo "rep $i exit $?: $(tail -1 gpurun_out/pytest_rep$i.log)"
done
