#!/bin/bash
# the round's closing visit: full GPU suite, smoke, then scripts/gpu_round.sh stages on the same box
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/r05b; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -rP --maxfail 20 --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_gpu.log | cut -c1-300
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu_round.sh r05b bench trace pmc others
