#!/bin/bash
# BA-25 / BA-512 quick numbers for the library in ESFM_LIB (or the in-tree one)
python scratch/ba25.py 2>&1 | grep iters | awk '{print "ba25 it/s", $6}'
python scratch/ba512.py 2>&1 | grep -E "iters|schur|linearize|solve" | awk '{printf "%s %s %s | ", $1, $2, $6} END {print ""}'
