#!/bin/bash
# Samples the shader clock / power while a workload runs (is the chip clock-throttled when kernels of several stages co-run?)
# usage: bash tools/dev/clock_probe.sh <label> <command...>
label=$1; shift
"$@" > /dev/null 2>&1 &
pid=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket" | tr '\n' ' '
  echo
  sleep 1
done | sed "s/^/$label: /"
wait $pid
