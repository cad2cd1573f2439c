#!/bin/bash
# same-call A/B of builds of libc8.so kept in tools/_ab/*.so: K1 on the bench workload through tools/time_k1.py, each build twice
# usage (GPU box): bash tools/ab_k1.sh [time_k1.py arguments]
cd $GRAFT_REPO_ROOT
cp calibr8_amd/libc8.so /tmp/libc8_keep.so
for round in 1 2; do
  for f in tools/_ab/*.so; do
    cp $f calibr8_amd/libc8.so
    timeout -k 10 300 python3 tools/time_k1.py --tag $(basename $f .so) "$@" 2>/dev/null
  done
done
cp /tmp/libc8_keep.so calibr8_amd/libc8.so
