#!/bin/bash
# build_all_counted.sh: every instrumented library profiles/collect.py uses (.ab/cnt_*: basic-block counts of one kernel each)
# and the two lane-count builds (.ab/lan_*), from the working tree.  ~1 minute each after the first (shared device assembly).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/profiles/tools/build_counted.sh
bash $B cnt_fn k_bounceILi0ELb1ELi0ELb1ELb0ELb0E      # C2 / C3 / C5: fused + compaction
bash $B cnt_fg k_bounceILi0ELb1ELi0ELb1ELb1ELb0E      # ... generating bounce 0's camera rays
bash $B cnt_sn k_bounceILi0ELb1ELi0ELb1ELb0ELb1E      # material sort folded into the compaction
bash $B cnt_sg k_bounceILi0ELb1ELi0ELb1ELb1ELb1E
bash $B cnt_tn k_bounceILi0ELb1ELi1ELb0ELb0ELb0E      # the loop over every triangle (C4 as stated)
bash $B cnt_tg k_bounceILi0ELb1ELi1ELb0ELb1ELb0E
bash $B cnt_pn k_bounceILi0ELb1ELi3ELb1ELb0ELb0E      # PT_MESH_BVH: k_bounce behind the mesh pre-pass
bash $B cnt_pg k_bounceILi0ELb1ELi3ELb1ELb1ELb0E
bash $B cnt_km k_meshILb1E
bash $B cnt_it k_iterationILb1E
COUNT_MODE=lanes bash $B lan_fn k_bounceILi0ELb1ELi0ELb1ELb0ELb0E
COUNT_MODE=lanes bash $B lan_km k_meshILb1E
