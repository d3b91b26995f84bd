#!/bin/bash
# single-variant libraries of the f16 decoder-forward kernel: args "FT,NW,PF,PFB"
cd "$(dirname "$0")/.."
mkdir -p sdflabel_amd/lib/ab
for cfg in "$@"; do
  IFS=, read ft nw pf pfb <<< "$cfg"
  SDFR_F16_DEFS="-DSDFR_H_FT=$ft -DSDFR_H_NW=$nw -DSDFR_H_PF=$pf -DSDFR_H_PFB=$pfb" SDFR_OUT=sdflabel_amd/lib/ab SDFR_LIBNAME=libsdfr_h_${ft}_${nw}_${pf}_${pfb}.so bash sdflabel_amd/csrc/build.sh 2>&1 | grep -E "error|built"
done
bash sdflabel_amd/csrc/build.sh 2>&1 | grep -E "error|built"
