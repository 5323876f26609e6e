#!/bin/bash
# Build the current tree into tools/ab/$1.so (for same-box A/B timing: DDSPP_LIB=tools/ab/$1.so python bench.py ...)
set -e
cd "$(dirname "$0")/.."
python -c "from ddsp_piano_amd import _lib; _lib.build(verbose=False)"
mkdir -p tools/ab
cp ddsp_piano_amd/libddspp.so tools/ab/$1.so
echo "tools/ab/$1.so"
