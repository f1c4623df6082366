#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=gpurun_out/r02_tests_all
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu --durations=12 > $OUT/pytest_all.txt 2>&1
grep -n "^E  \|passed\|failed\|FAILED" $OUT/pytest_all.txt | cut -c1-300 | head -40
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
