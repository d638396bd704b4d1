#!/bin/bash
# like build_variant.sh, for flags that several sources must agree on:
# usage: tools/build_variant2.sh NAME "EXTRA FLAGS" SOURCE.hip [SOURCE.hip ...]
set -e
NAME=$1; EXTRA=$2; shift 2
CS=/root/repo/stanford-ctc_amd/csrc
make -s -j8 -C $CS >/dev/null
mkdir -p /tmp/var_$NAME
cd /tmp/var_$NAME
SKIP=""
for SRC in "$@"; do
  BASE=$(basename $SRC .hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $EXTRA -I$CS -c $CS/$SRC -o $BASE.o &
  SKIP="$SKIP -e /$BASE.o"
done
wait
OBJS=$(ls $CS/build/*.o | grep -v $SKIP)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/stanford-ctc_amd/libvar_$NAME.so *.o $OBJS
echo built /root/repo/stanford-ctc_amd/libvar_$NAME.so
