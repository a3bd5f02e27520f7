#!/bin/bash
# Run a gpurun command with the reference checkout staged on the GPU box.
#
# The GPU boxes have no /root/reference, so tests/test_install_reference.py -- install() on the reference's REAL modules --
# is skipped there.  This wrapper copies the few reference files that test imports into an UNTRACKED, git-ignored scratch
# directory inside the snapshot gpurun ships (.refstage/), runs the command with NRNERF_REFERENCE pointing at it, and
# removes the directory again.  Nothing of the reference is ever committed.
#   tools/with_reference.sh [--timeout S] -- '<command>'
set -euo pipefail
REPO="$(cd "$(dirname "$0")/.." && pwd)"
REF="${NRNERF_REFERENCE_SRC:-/root/reference}"
STAGE="$REPO/.refstage"
TIMEOUT=900
if [ "${1:-}" = "--timeout" ]; then TIMEOUT="$2"; shift 2; fi
[ "${1:-}" = "--" ] && shift
trap 'rm -rf "$STAGE"' EXIT
rm -rf "$STAGE"; mkdir -p "$STAGE"
cp "$REF/train.py" "$REF/run_nerf_helpers.py" "$REF/load_llff.py" "$STAGE/" 2>/dev/null || cp "$REF/train.py" "$REF/run_nerf_helpers.py" "$STAGE/"
/usr/local/graft/bin/gpurun --timeout "$TIMEOUT" -- "export NRNERF_REFERENCE=\$PWD/.refstage; $*"
