#!/bin/bash
# scan stage of the dickens-sized bytes as N chunks: one workgroup per chunk (forced) against the chip-wide sort (forced), for the cost model of x3_scan_seg_applies
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "per chunk (X3H_SEG_MIN=1)"; X3H_SEG_MIN=1 python3 tools/chunked_dickens.py 48 64 80 96 128 256 2>&1 | grep -a chunks | sed 's/ratio.*total/total/'
echo "chip wide (X3H_SEG_MIN=0)"; X3H_SEG_MIN=0 python3 tools/chunked_dickens.py 48 64 80 96 128 256 2>&1 | grep -a chunks | sed 's/ratio.*total/total/'
