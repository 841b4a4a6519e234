#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== verbatim window probe"
timeout -k 10 200 ./scripts/probes/pk_window_probe 2048 20000 || exit 1
