# (history: AMK_SWEEP_VARIANT / AMK_SWEEP_TARGET=1 meant the persistent-lane kernel / the one-tile target when this ran; both are patches now
#  -- tools/experiments/patches/r05_sweep_persistent_lanes.patch -- and AMK_SWEEP_TARGET=1 is the fine hashed grid, the default)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05kfAp; mkdir -p $O; : > $O/err.txt
for tg in 1 0; do
rm -rf $O/kt; AMK_SWEEP_TARGET=$tg AMK_SWEEP_VARIANT=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --workload flight --keyframes 3 --streams 1 --gang 2 --no-parity --no-cpu-baseline > /dev/null 2>> $O/err.txt
db=$(find $O/kt -name "*.db" | head -1); echo "== target $tg"; python tools/rocprof_summary.py $db | grep -v "at::native\|Cijk\|rocprim" | head -14 | cut -c1-150
done
rm -rf $O/kt
