#!/bin/bash
# Round 4: graph replay is now the default for small steps (graphs="auto") with GModule on its own stream and one-chain
# backward graphs: fresh-process repeats of the cases that exercise exactly that.  RUNS=40 bash tools/stress_graphs_r04.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/stress_graphs
export PYTHONWARNINGS=ignore HSA_ENABLE_IPC_MODE_LEGACY=0
fails=0; n=0
for i in $(seq 1 ${RUNS:-40}); do
  for c in "graphs_auto_switches_with_the_batch_size" "graphed_full_workload_matches_eager 1" "side_streams_run_beside_the_main_stream"; do
    n=$((n+1))
    timeout 300 python -X faulthandler -m tests.helpers.graph_cases $c > gpurun_out/stress_graphs/run.txt 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then
      fails=$((fails+1)); cp gpurun_out/stress_graphs/run.txt "gpurun_out/stress_graphs/fail_${i}_$(echo $c | cut -d' ' -f1).txt"
      echo "run $i [$c] rc=$rc"; grep -v "dist-packages" gpurun_out/stress_graphs/run.txt | tail -30
    fi
  done
done
rm -f gpurun_out/stress_graphs/run.txt
echo "round 4 graph stress: $n fresh-process runs (${RUNS:-40} x 3 cases: graphs=auto switching, full workload replayed with GModule's stream, probed side streams under RCCL), failures: $fails" | tee gpurun_out/stress_graphs/summary_r04.txt
