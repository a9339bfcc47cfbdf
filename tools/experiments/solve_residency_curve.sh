#!/bin/bash
# Saturated solve-only rate against the number of resident solve waves per CU (capped through the dynamic LDS request; needs the
# library built with tools/experiments/patches/r06_solve_lds_rt.patch).  Where does the stretch of a wave at 8 per CU come from:
# from sharing its SIMD with one other wave (then 1..4 per CU scale linearly) or from sharing the CU (LDS, scalar, fetch)?
# The request sits half-way between 160 KB / n and 160 KB / (n + 1), so that allocation granules cannot change the count.
cd $GRAFT_REPO_ROOT
for n in 8 7 6 5 4 3 2 1; do
  lds=$(python -c "print(int(163840 / ($n + 0.5)) // 256 * 256)")
  echo "cap $n per CU (lds request $lds): $(AMK_REPS=${AMK_REPS:-256} AMK_SOLVE_LDS_MIN_RT=$lds python tools/experiments/solve_rate.py 2>/dev/null | grep solve-only)"
done
