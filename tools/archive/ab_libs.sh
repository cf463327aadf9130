#!/bin/bash
# Diagnostic: A/B of two builds of the engine (PGX_LIB) in alternating processes.  usage: tools/ab_libs.sh old.so "cfg2 cfg4"
OLD=$1; shift
for rep in 1 2 3; do for wl in ${1:-cfg2}; do
  for lib in "$OLD" ""; do
    PGX_LIB=$lib python tools/ab_inproc.py $wl PGX_FLAGS=0 2>&1 | grep -v amdgpu | sed "s#PGX_FLAGS=0#$( [ -n "$lib" ] && echo old || echo new)#"
  done; done; done
