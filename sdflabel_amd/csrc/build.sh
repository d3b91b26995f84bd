#!/bin/bash
# Build libsdfr_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${SDFR_OUT:-$HERE/../lib}"
mkdir -p "$OUT" "$HERE/obj"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
# Product builds take NO compile-time options from the environment.  The per-TU define hooks (kernel geometry A/B, timing-only ablations) exist
# only for tools/ab_variant.sh, which sets SDFR_AB=1: such a library is compiled with -DSDFR_EXPERIMENT, reports it through sdfr_build_flags()
# and is refused by sdflabel_amd/_lib.py unless SDFR_ALLOW_AB=1.
if [ "${SDFR_AB:-0}" = "1" ]; then
  ALLDEFS="$SDFR_FWD_DEFS $SDFR_F16_DEFS $SDFR_SPLIT_DEFS $SDFR_J16_DEFS $SDFR_JAC_DEFS $SDFR_LOSS_DEFS"
  COMMON="$COMMON -DSDFR_EXPERIMENT=1 $SDFR_ALL_DEFS"         # SDFR_ALL_DEFS: a define for every translation unit (e.g. -DSDFR_BOX_PAD=...)
else
  for v in SDFR_FWD_DEFS SDFR_F16_DEFS SDFR_SPLIT_DEFS SDFR_J16_DEFS SDFR_JAC_DEFS SDFR_LOSS_DEFS; do
    [ -n "${!v}" ] && echo "build.sh: ignoring $v (set SDFR_AB=1 for an experiment build)" >&2
  done
  SDFR_FWD_DEFS=; SDFR_F16_DEFS=; SDFR_SPLIT_DEFS=; SDFR_J16_DEFS=; SDFR_JAC_DEFS=; SDFR_LOSS_DEFS=; ALLDEFS=
  # (SDFR_OUT alone is fine for a product build -- packaging writes the library outside the source tree; only the define hooks and the
  # variant library NAMES are locked)
  if [ -n "$SDFR_LIBNAME" ]; then echo "build.sh: SDFR_LIBNAME needs SDFR_AB=1" >&2; exit 2; fi
fi
if [ "${SDFR_BUILD_DRYRUN:-0}" = "1" ]; then      # (tests/test_host_cpu.py: what WOULD be passed to the compiler)
  echo "defs:[$(echo $ALLDEFS $SDFR_FWD_DEFS $SDFR_F16_DEFS $SDFR_SPLIT_DEFS $SDFR_J16_DEFS $SDFR_JAC_DEFS $SDFR_LOSS_DEFS)] common:[$COMMON]"; exit 0
fi
# the MLP uses MFMA + explicit fmaf; the geometric kernels keep separate roundings like the reference's ATen ops
$HIPCC $COMMON $ALLDEFS -c "$HERE/common.hip"  -o "$HERE/obj/common.o" &
$HIPCC $COMMON -c "$HERE/mlp.hip"     -o "$HERE/obj/mlp.o" &
$HIPCC $COMMON $SDFR_FWD_DEFS -c "$HERE/mlp_fwd32.hip" -o "$HERE/obj/mlp_fwd32.o" &
$HIPCC $COMMON $SDFR_F16_DEFS -c "$HERE/mlp_fwd16.hip" -o "$HERE/obj/mlp_fwd16.o" &
$HIPCC $COMMON $SDFR_SPLIT_DEFS -c "$HERE/mlp_split.hip" -o "$HERE/obj/mlp_split.o" &
$HIPCC $COMMON $SDFR_J16_DEFS -c "$HERE/mlp_jac16.hip" -o "$HERE/obj/mlp_jac16.o" &
$HIPCC $COMMON $SDFR_JAC_DEFS -c "$HERE/mlp_jac.hip"   -o "$HERE/obj/mlp_jac.o" &
$HIPCC $COMMON -c "$HERE/mlp_persist.hip" -o "$HERE/obj/mlp_persist.o" &
$HIPCC $COMMON -c "$HERE/mlp_small.hip" -o "$HERE/obj/mlp_small.o" &
$HIPCC $COMMON -c "$HERE/mlp_ln.hip"    -o "$HERE/obj/mlp_ln.o" &
$HIPCC $COMMON -ffp-contract=off -c "$HERE/surface.hip" -o "$HERE/obj/surface.o" &
$HIPCC $COMMON -ffp-contract=off -c "$HERE/project.hip" -o "$HERE/obj/project.o" &
$HIPCC $COMMON -ffp-contract=off -c "$HERE/splat.hip"   -o "$HERE/obj/splat.o" &
$HIPCC $COMMON -ffp-contract=off -c "$HERE/params.hip"  -o "$HERE/obj/params.o" &
$HIPCC $COMMON -ffp-contract=off $SDFR_LOSS_DEFS -c "$HERE/losses.hip"  -o "$HERE/obj/losses.o" &
$HIPCC $COMMON -c "$HERE/trace.hip"   -o "$HERE/obj/trace.o" &
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/${SDFR_LIBNAME:-libsdfr_hip.so}" "$HERE"/obj/{common,mlp,mlp_fwd32,mlp_fwd16,mlp_split,mlp_jac,mlp_jac16,mlp_persist,mlp_small,mlp_ln,surface,project,splat,params,losses,trace}.o
echo "built $OUT/${SDFR_LIBNAME:-libsdfr_hip.so}"
