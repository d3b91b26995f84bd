#!/bin/bash
# Build single-variant EXPERIMENT libraries into sdflabel_amd/lib/ab/:  tools/ab_variant.sh NAME=ENVVAR:"-Ddefs ..." ...
#   ENVVAR is one of the per-TU define hooks of csrc/build.sh (SDFR_FWD_DEFS, SDFR_F16_DEFS, SDFR_JAC_DEFS, SDFR_J16_DEFS, SDFR_SPLIT_DEFS,
#   SDFR_LOSS_DEFS); the hooks are honoured only with SDFR_AB=1 (set here), the library is compiled with -DSDFR_EXPERIMENT, reports it in
#   sdfr_build_flags() and loads only with SDFR_ALLOW_AB=1 SDFR_LIB=sdflabel_amd/lib/ab/libsdfr_NAME.so.
# e.g. f32 forward rings:   tools/ab_variant.sh f3_2=SDFR_FWD_DEFS:"-DSDFR_FWD_PF=3 -DSDFR_FWD_PFB=2"
#      f16 forward geometry: tools/ab_variant.sh h=SDFR_F16_DEFS:"-DSDFR_H_FT=2 -DSDFR_H_NW=8 -DSDFR_H_PF=4 -DSDFR_H_PFB=2"
#      64-row half kernels:  tools/ab_variant.sh t64=SDFR_J16_DEFS:"-DSDFR_T64_PF=3 -DSDFR_T64_PFB=2"
#      ablation (wrong results, timing only): tools/ab_variant.sh nomfma=SDFR_F16_DEFS:"-DSDFR_ABL_NOMFMA"
cd "$(dirname "$0")/.."
mkdir -p sdflabel_amd/lib/ab
for spec in "$@"; do
  name="${spec%%=*}"; rest="${spec#*=}"; var="${rest%%:*}"; defs="${rest#*:}"
  env SDFR_AB=1 "$var=$defs" SDFR_OUT=sdflabel_amd/lib/ab SDFR_LIBNAME=libsdfr_$name.so bash sdflabel_amd/csrc/build.sh 2>&1 | grep -E "error|built|warning: .*spill"
done
bash sdflabel_amd/csrc/build.sh 2>&1 | grep -E "error|built"     # restore the default objects / library
