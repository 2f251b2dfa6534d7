#!/bin/bash
# incremental build of the library + refresh of its source-hash stamp (run from anywhere)
R=$(cd "$(dirname "$0")/.." && pwd)
make -s -C "$R/tinympc-julia_amd/csrc" -j8 2>&1 | grep -v "^$" | head -30
cd "$R" && python -c "
import tinympc_julia_amd as t
t.build(); t.load_library(); print('ok')"
