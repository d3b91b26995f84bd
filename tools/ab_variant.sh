#!/bin/bash
# Build single-variant libraries into sdflabel_amd/lib/ab/:  tools/ab_variant.sh NAME=ENVVAR:"-Ddefs ..." ...
#   ENVVAR is one of the per-TU define hooks of csrc/build.sh (SDFR_FWD_DEFS, SDFR_F16_DEFS, SDFR_JAC_DEFS, SDFR_J16_DEFS, SDFR_SPLIT_DEFS)
# e.g. tools/ab_variant.sh j64=SDFR_JAC_DEFS:"-DSDFR_JAC_MS=32 -DSDFR_JAC_FT=2 -DSDFR_JAC_NP=2 -DSDFR_JAC_NW=8 -DSDFR_JAC_PF=2"
cd "$(dirname "$0")/.."
mkdir -p sdflabel_amd/lib/ab
for spec in "$@"; do
  name="${spec%%=*}"; rest="${spec#*=}"; var="${rest%%:*}"; defs="${rest#*:}"
  env "$var=$defs" SDFR_OUT=sdflabel_amd/lib/ab SDFR_LIBNAME=libsdfr_$name.so bash sdflabel_amd/csrc/build.sh 2>&1 | grep -E "error|built|warning: .*spill" 
done
bash sdflabel_amd/csrc/build.sh 2>&1 | grep -E "error|built"     # restore the default objects / library
