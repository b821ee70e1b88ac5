#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=5 -p no:cacheprovider 2>&1 | tail -8
