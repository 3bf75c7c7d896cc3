#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export DBG_PAIRS=${DBG_PAIRS:-0}
RGBDFE_RANSAC_SPLIT=0 python tools/r04_debug_split.py run ref 2>&1 | tail -1
RGBDFE_RANSAC_SPLIT=1 RGBDFE_SPLIT_DEBUG=0 python tools/r04_debug_split.py run s_0 2>&1 | tail -1



python tools/r04_debug_split.py cmp
