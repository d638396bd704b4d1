#!/bin/bash
# builds stanford-ctc_amd/libvar_<BK>_<OCC>.so: the library with a GEMM tiling variant
# usage: tools/build_variant.sh BK OCC
set -e
BK=$1; OCC=$2; EXTRA=$3
CS=/root/repo/stanford-ctc_amd/csrc
make -s -j8 -C $CS >/dev/null
mkdir -p /tmp/var_${BK}_${OCC}
cd /tmp/var_${BK}_${OCC}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSCTC_GEMM_BK=$BK -DSCTC_GEMM_OCC=$OCC $EXTRA -c $CS/gemm_f32.hip -o gemm_f32.o -save-temps=obj 2>/dev/null
OBJS=$(ls $CS/build/*.o | grep -v gemm_f32.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/stanford-ctc_amd/libvar_${BK}_${OCC}${EXTRA:+_x}.so gemm_f32.o $OBJS
echo built /root/repo/stanford-ctc_amd/libvar_${BK}_${OCC}${EXTRA:+_x}.so
