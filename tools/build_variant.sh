#!/bin/bash
# tools/build_variant.sh <name> [--solver-flags="..."] [extra hipcc flags...]  ->  forces_resilient_planner_amd/lib_<name>.so  (experiments only)
# The product sources with their per-source flags plus the extra flags; --solver-flags="" compiles the solver kernel's translation
# unit with the compiler's default code generation instead of build.py's CODEGEN_FLAGS.
set -e
cd "$(dirname "$0")/.."
python -m forces_resilient_planner_amd.build variant "$@"
