#!/bin/bash
# Regenerates tests/golden/* from the reference's own host C++ (see oracle/ref/refgen.cpp).
# Runs only where /root/reference exists (this container); the fixtures it writes are committed.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${REF:-/root/reference}"
make -C "$HERE" ref REF="$REF"
mkdir -p "$HERE/../tests/golden"
"$HERE/_ref/refgen" "$REF" "$HERE/../tests/golden"
# family (8): the reference's Splat4D::Draw / Splat3D::Draw on the CPU (oracle/ref/refdraw_main.cpp); reads the records refgen just wrote
make -C "$HERE" refdraw REF="$REF"
"$HERE/_ref/refdraw" "$REF" "$HERE/../tests/golden"
# the GPU half: the reference's GLSL programs executed by Mesa llvmpipe (oracle/ref/refgl_main.cpp) -> tests/golden/gl_*.npz
make -C "$HERE" refgl
python3 "$HERE/make_golden_gl.py"
