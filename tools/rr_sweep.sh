#!/bin/bash
# chunk sizes of the fused residual + restriction kernel (one process per setting: the library reads the variable once)
for kc in 0 1 2 4 8 16; do PMG_GRID_RR_CHUNK=$kc timeout -k 5 120 python tools/rrbench.py 2>&1 | grep "\^3" || exit 1; done
