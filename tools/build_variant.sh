#!/bin/bash
# Builds one experimental variant of libcglb_hip.so (cglb_amd/csrc/Makefile EXTRA_DEFS) for tools/k1_variants.py.
#   tools/build_variant.sh NAME "-DFOO=1 -DBAR"   ->  cglb_amd/lib/variants/libcglb_NAME.so
# Every variant gets its OWN object directory, created from clean: make does not track EXTRA_DEFS, so objects of two define sets
# sharing one BUILD directory would be linked together without a rebuild (the kernels and the host code that sizes / fills their
# tables must come from the same define set).  Round 2's libcglb_d3fs1.so faulted on the GPU ("Memory access fault by GPU node")
# after being built in the shared directory; see profiles/r02_k1_variants.log.
set -euo pipefail
NAME="$1"; DEFS="${2:-}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
BUILD="$ROOT/build/variants/$NAME"
rm -rf "$BUILD"
mkdir -p "$BUILD" "$ROOT/cglb_amd/lib/variants"
make -j"${JOBS:-8}" -C "$ROOT/cglb_amd/csrc" BUILD="$BUILD" OUT="$ROOT/cglb_amd/lib/variants/libcglb_$NAME.so" EXTRA_DEFS="$DEFS"
echo "$DEFS" > "$ROOT/cglb_amd/lib/variants/libcglb_$NAME.defs"   # provenance, beside the library (build/ does not travel to the GPU box)
