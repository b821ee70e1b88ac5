#!/bin/bash
# the whole GPU suite + smoke of the tree as it stands
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4full; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/tests.log 2>&1; echo "tests rc=$? $(tail -1 $out/tests.log)"; grep "FAILED\|ERROR" $out/tests.log | head
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $out/smoke.log)"
