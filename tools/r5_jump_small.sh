#!/usr/bin/env bash
# GPU box: the small-batch kernels (k_search_g8) with the search started at the lowest cell that holds the warm ball, against the root
for j in 1 0; do echo "TDTK_JUMP_START=$j"; TDTK_JUMP_START=$j python tools/small_iter_probe.py 2>&1 | tail -4; TDTK_JUMP_START=$j python tools/small_scan_probe.py 2>&1 | grep -E "timing=0 rep 2|match with (10|40)"; done
