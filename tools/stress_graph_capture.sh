#!/bin/bash
# Repeats capture-heavy HIP-graph cases in fresh processes (looking for the rare fault inside stream capture).
#   RUNS=40 bash tools/stress_graph_capture.sh      -> gpurun_out/stress_graphs/summary.txt (+ the output of failing runs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/stress_graphs
export PYTHONWARNINGS=ignore HSA_ENABLE_IPC_MODE_LEGACY=0
fails=0
for i in $(seq 1 ${RUNS:-40}); do
  for c in "graphed_full_workload_matches_eager 1" "graphed_fpn_step_is_bitwise_the_eager_step" "graphed_step_over_one_rank_rccl_group"; do
    timeout 300 python -X faulthandler -m tests.helpers.graph_cases $c > gpurun_out/stress_graphs/run.txt 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then
      fails=$((fails+1)); cp gpurun_out/stress_graphs/run.txt "gpurun_out/stress_graphs/fail_${i}_$(echo $c | cut -d' ' -f1).txt"
      echo "run $i [$c] rc=$rc"; grep -v "dist-packages" gpurun_out/stress_graphs/run.txt | tail -40
    fi
  done
done
rm -f gpurun_out/stress_graphs/run.txt
echo "runs: ${RUNS:-40} x 3 cases, failures: $fails" | tee gpurun_out/stress_graphs/summary.txt
