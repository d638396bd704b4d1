#!/bin/bash
# builds stanford-ctc_amd/libvar_<NAME>.so: the library with ONE source recompiled with extra
# flags (kernel-variant experiments; the variant is selected with SCTC_LIB_PATH)
# usage: tools/build_variant.sh NAME SOURCE.hip "EXTRA FLAGS"
set -e
NAME=$1; SRC=$2; EXTRA=$3
CS=/root/repo/stanford-ctc_amd/csrc
make -s -j8 -C $CS >/dev/null
mkdir -p /tmp/var_$NAME
cd /tmp/var_$NAME
BASE=$(basename $SRC .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $EXTRA -I$CS -c $CS/$SRC -o $BASE.o -save-temps=obj 2>/dev/null
OBJS=$(ls $CS/build/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/stanford-ctc_amd/libvar_$NAME.so $BASE.o $OBJS
echo built /root/repo/stanford-ctc_amd/libvar_$NAME.so
