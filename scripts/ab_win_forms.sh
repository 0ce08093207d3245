#!/bin/bash
# same-process A/B of window-compressor variant libraries in the global-slot (wing) and dual (wind) forms, optional slot counts
cd "$(dirname "$0")/.."
for L in ${FORMS:-wing wind}; do
  MODE=compress LAYOUT=$L NB=${NB:-16383} DATA=${DATA:-html} REPS=${REPS:-4} timeout 600 python scripts/ab_libs.py "$@" 2>&1 | grep '^{'
done
