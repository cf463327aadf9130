#!/bin/bash
# Diagnostic: on a box whose zone walk finds nothing, does another workgroup -> slice mapping or launch shape stream faster?
python tools/ab_inproc.py cfg2 PGX_FLAGS=0 PGX_FLAGS=8 PGX_WAVES=1 PGX_WAVES=1,PGX_FLAGS=8 PGX_STORE=nt PGX_STORE=plain PGX_FLAGS=0 2>&1 | grep -v amdgpu | sed 's/, .observe_us_zone.*//' | tail -10
