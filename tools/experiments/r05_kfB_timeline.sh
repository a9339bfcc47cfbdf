# In-flight kernel durations of the yaml-configuration flights with the keyframe map (12 slots x 4) beside the single-stream ones
# (rocprofv3 slows this many-small-kernels workload ~2.7x: read the RATIOS, not the rates)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05kfB_tl; mkdir -p $O; : > $O/err.txt
B="--workload flight --keyframes 100 --points 3072 --T 1.0 --K 3 --no-parity --no-cpu-baseline --periods 60 --gang 4"
for st in 12 1; do
  rm -rf $O/kt
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py $B --streams $st > $O/B_$st.json 2>> $O/err.txt
  db=$(find $O/kt -name "*.db" | head -1)
  echo "== streams $st"; python tools/rocprof_summary.py $db | grep -v "at::native\|Cijk\|rocprim" | head -16 | cut -c1-150
done
rm -rf $O/kt
