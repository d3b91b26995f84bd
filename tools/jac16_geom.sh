#!/bin/bash
# float16 band Jacobian: tile geometries side by side (tools/jac16_time.py): tools/jac16_geom.sh [B ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for g in "" 32x1 32x2 32x4; do
  echo "== SDFR_J16_GEOM=$g"; SDFR_J16_GEOM=$g timeout 300 python tools/jac16_time.py "${@:-1 2 4 8 64}" 2>&1 | tail -1
done
