#!/bin/bash
# conv_microbench of one shape set / match in the round-start tree (ab_base/) and in HEAD on ONE box:  tools/ab_micro.sh <shapes> <match> <only>
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
for tree in ab_base . ab_base .; do
  echo "== tree $tree"
  (cd $tree && python tools/conv_microbench.py --shapes $1 --match "$2" --only $3 --iters 20 --repeat 3 2>&1 | grep "$2")
done
