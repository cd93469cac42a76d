#!/bin/bash
# usage (ON the GPU box): scripts/run_ab_lib.sh NAME [NAME ...]  - GPU conv tests with variant library NAME, then cfg2 step A/B base vs the variants
for v in "$@"; do
  echo "== tests with variant $v"
  XV2_LIB=$PWD/xview2_amd/abl/xv2_$v.so python -m pytest tests/test_conv_shapes_gpu.py tests/test_f16x2_gpu.py tests/test_ops_gpu.py -q -x 2>&1 | tail -2
done
SETS=("XV2_NOP=1")
for v in "$@"; do SETS+=("XV2_LIB=$PWD/xview2_amd/abl/xv2_$v.so"); done
scripts/ab_multi.sh "${SETS[@]}" -- --steps 30 --warmup 8
