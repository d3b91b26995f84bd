#!/bin/bash
# single-variant libraries of the 64-row half kernels (forward on 64-row tiles + looping tail): args "PF,PFB" ...
cd "$(dirname "$0")/.."
mkdir -p sdflabel_amd/lib/ab
for cfg in "$@"; do
  IFS=, read pf pfb <<< "$cfg"
  SDFR_J16_DEFS="-DSDFR_T64_PF=$pf -DSDFR_T64_PFB=$pfb" SDFR_OUT=sdflabel_amd/lib/ab SDFR_LIBNAME=libsdfr_t64_${pf}_${pfb}.so bash sdflabel_amd/csrc/build.sh 2>&1 | grep -E "error|built"
done
bash sdflabel_amd/csrc/build.sh 2>&1 | grep -E "error|built"
