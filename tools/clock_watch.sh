#!/bin/bash
# Runs on the GPU box: the default bench twice in a row, with the engine clock and power sampled meanwhile
# (is the level-1 kernel's time -- the one instruction-bound kernel of the step -- a matter of the clock the box grants?)
# usage: tools/clock_watch.sh <tag> [bench args]
TAG=$1; shift
for i in 1 2; do
  ( timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/cw_${TAG}_$i.json 2> /dev/null ) &
  BP=$!
  while kill -0 $BP 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|mclk" | tr '\n' ' ' >> gpurun_out/cw_${TAG}_$i.clk
    echo >> gpurun_out/cw_${TAG}_$i.clk
    sleep 0.3
  done
  wait $BP || exit 1
done
