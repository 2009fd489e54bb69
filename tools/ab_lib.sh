#!/bin/bash
# tools/ab_lib.sh ROUNDS A.so B.so -- command ...   alternates the two builds of libnpm_hip.so under one command (A/B inside one
# GPU session; box-to-box spread on this pool is larger than most effects worth measuring)
rounds=$1; a=$2; b=$3; shift 4
lib=np_modeling_amd/lib/libnpm_hip.so
cp $lib /tmp/libnpm_hip_saved.so
for r in $(seq $rounds); do
  for v in A B; do
    if [ $v = A ]; then cp $a $lib; else cp $b $lib; fi
    echo "== build $v, round $r"; "$@" || { cp /tmp/libnpm_hip_saved.so $lib; exit 1; }
  done
done
cp /tmp/libnpm_hip_saved.so $lib
