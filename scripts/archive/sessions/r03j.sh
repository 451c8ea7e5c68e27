#!/bin/bash
# round 3, GPU session J: where K1s' distance to the bare stream goes — builds without the row part, the column part,
# the LDS accumulation (wrong results: memory skeletons of the same structure), alone on the GPU, one process
cd "$(dirname "$0")/../.."
O=gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
SRC=scripts/micro/xk_symm_r03_knobs.hip bash scripts/k1s_build_ab.sh $O "norow:-DXK_SYMM_NOROW" "nocol:-DXK_SYMM_NOCOL" "norowcol:-DXK_SYMM_NOROW -DXK_SYMM_NOCOL" "nolds:-DXK_SYMM_NORED" "skeleton:-DXK_SYMM_NOROW -DXK_SYMM_NOCOL -DXK_SYMM_NORED"
