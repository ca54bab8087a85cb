#!/bin/bash
# Rebuild every native artefact, then run a command on the MI355X box (avoids testing a stale .so).
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "error|Error" && exit 1
exec /usr/local/graft/bin/gpurun --timeout "${GPU_TIMEOUT:-900}" -- "$@"
