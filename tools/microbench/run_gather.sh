#!/bin/bash
# Runs on the GPU box: the node-gather microbenchmark over table sizes (teapot 2 MB, bunny 37 MB, 1.15 M-triangle stand-in
# 74 MB) and hot-set shares (0 = uniformly random, 6/8 and 7/8 of the steps revisit a 16 KB hot set like the top of a tree).
cd "$(dirname "$0")"
for N in 31410 576186 1152371; do
  for HOT in 0 6 7; do
    echo "== n_nodes $N hot $HOT/8"
    timeout -k 10 120 ./gather_nodes $N 64 $HOT || exit 1
  done
done
