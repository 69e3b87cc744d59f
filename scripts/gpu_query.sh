#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "query or abi or export or ref_named or grid_interpolate" > gpurun_out/pytest_query.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_query.log
tail -30 gpurun_out/pytest_query.log | cut -c1-250
