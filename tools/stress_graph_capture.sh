#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/stress_graphs
export PYTHONWARNINGS=ignore
fails=0
for i in $(seq 1 ${RUNS:-8}); do
  timeout 600 python -X faulthandler -m pytest tests/test_abi.py tests/test_datasets.py tests/test_fullsize_properties_gpu.py tests/test_graphs_gpu.py -m gpu -x -q -p no:cacheprovider > gpurun_out/stress_graphs/run_$i.txt 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "run $i rc=$rc"; grep -v "dist-packages" gpurun_out/stress_graphs/run_$i.txt | head -60; else rm gpurun_out/stress_graphs/run_$i.txt; fi
done
echo "failures: $fails"
