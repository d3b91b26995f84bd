#!/bin/bash
# Build single-variant libraries of the f32 decoder-forward kernel (weight ring PF, activation ring PFB) into sdflabel_amd/lib/ab/.
# Each library contains exactly one instantiation of that kernel family, so variants do not perturb each other's codegen.
cd "$(dirname "$0")/.."
mkdir -p sdflabel_amd/lib/ab
# args: "PF,PFB" or "PF,PFB,FT,NW,NP"
for cfg in "$@"; do
  IFS=, read pf pfb ft nw np <<< "$cfg"
  defs="-DSDFR_FWD_PF=$pf -DSDFR_FWD_PFB=$pfb"
  [ -n "$ft" ] && defs="$defs -DSDFR_FWD_FT=$ft -DSDFR_FWD_NW=$nw -DSDFR_FWD_NP=$np"
  [ "$nw" = "4" ] && defs="$defs -DSDFR_MLP_WPE=2"
  SDFR_FWD_DEFS="$defs" SDFR_OUT=sdflabel_amd/lib/ab SDFR_LIBNAME=libsdfr_${cfg//,/_}.so bash sdflabel_amd/csrc/build.sh 2>&1 | grep -E "error|built"
done
bash sdflabel_amd/csrc/build.sh 2>&1 | grep -E "error|built"     # restore the default objects / library
