#!/bin/bash
# Build single-variant libraries of the f32 decoder-forward kernel (weight ring PF, activation ring PFB) into sdflabel_amd/lib/ab/.
# Each library contains exactly one instantiation of that kernel family, so variants do not perturb each other's codegen.
cd "$(dirname "$0")/.."
mkdir -p sdflabel_amd/lib/ab
for cfg in "$@"; do
  pf=${cfg%,*}; pfb=${cfg#*,}
  SDFR_FWD_DEFS="-DSDFR_FWD_PF=$pf -DSDFR_FWD_PFB=$pfb" SDFR_OUT=sdflabel_amd/lib/ab SDFR_LIBNAME=libsdfr_${pf}_${pfb}.so bash sdflabel_amd/csrc/build.sh 2>&1 | grep -E "error|built"
done
bash sdflabel_amd/csrc/build.sh 2>&1 | grep -E "error|built"     # restore the default objects / library
