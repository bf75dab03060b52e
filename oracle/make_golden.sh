#!/bin/bash
# Regenerates tests/golden/* from the reference's own host C++ (see oracle/ref/refgen.cpp).
# Runs only where /root/reference exists (this container); the fixtures it writes are committed.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${REF:-/root/reference}"
make -C "$HERE" ref REF="$REF"
mkdir -p "$HERE/../tests/golden"
"$HERE/_ref/refgen" "$REF" "$HERE/../tests/golden"
