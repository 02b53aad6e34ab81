#!/bin/bash
# A/B of the tail schedule of the one-wave compositing kernel (option render_tail = N: from at most N unsaturated pixels per quadrant on)
# on the C2 recipe, sigma0 = 0.05, the larger synthetic set and the real image's merged set. Frame hashes must agree per workload.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in "--gaussians 196608 --views 120" "--gaussians 196608 --views 120 --sigma0 0.05" "--real --views 128" "--gaussians 589824 --views 128"; do
  for t in ${TAILS:-0 4 8 12 16 24 32}; do
    F3DG_OPT_RENDER_TAIL=$t python tools/ab_render.py $wl --steps ${STEPS:-6} --label "tail=$t" --counts ${EXTRA}
  done
done
