#!/bin/bash
# hash-grid backward A/B on the GPU box: parity tests of the hash grid, then timing of library variants (WISP_HIP_LIB)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_0_parity.py tests/test_gpu_1_selfcheck.py -m gpu -q --tb=short -p no:cacheprovider -k "hashgrid or flagship or stress or spill" > gpurun_out/pytest_hg.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_hg.log
tail -5 gpurun_out/pytest_hg.log
CS=kaolin-wisp_amd/csrc
for rep in 1 2; do
for lib in $CS/libwisp_hip.so $(ls $CS/ab/*.so 2>/dev/null); do WISP_HIP_LIB=$PWD/$lib timeout 300 python scripts/ab_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1; done
done | tee gpurun_out/ab_hg.log
(cd /tmp && rm -rf /tmp/prof_hg && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_hg -o hg -- python "$GRAFT_REPO_ROOT/scripts/ab_kernels.py" > /dev/null 2>&1)
python - <<'PY' | tee gpurun_out/hg_stats.txt
import csv, glob
for f in glob.glob('/tmp/prof_hg/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'hashgrid' in r['Name'] or 'mlp' in r['Name']:
            print(r['Name'].split('(')[0][-60:], r['Calls'], r['AverageNs'])
PY
