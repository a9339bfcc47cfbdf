cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
[ -z "$SKIPTESTS" ] && timeout 1500 python -m pytest tests/test_kfmap_gpu.py tests/test_step_frames_gpu.py tests/test_keyframe_gpu.py tests/test_cpp_adapters_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -5
B="--workload flight --config yaml --keyframes 100"
O=gpurun_out/r05kf2; mkdir -p $O
timeout 900 python bench.py $B > $O/B_12x4.json 2>$O/err.txt
timeout 600 python bench.py $B --no-parity --no-cpu-baseline --streams 16 > $O/B_16x4.json 2>>$O/err.txt
timeout 900 python bench.py --workload flight --keyframes 3 --streams 10 --gang 2 > $O/A_10x2.json 2>>$O/err.txt
timeout 600 python bench.py --workload flight --config yaml --no-parity --no-cpu-baseline > $O/B0_12x4.json 2>>$O/err.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $O/ktB -o kt -- python bench.py $B --no-parity --no-cpu-baseline --streams 1 > /dev/null 2>> $O/err.txt
db=$(find $O/ktB -name "*.db" | head -1); python tools/rocprof_summary.py $db | grep -v "at::native\|Cijk" | head -16 | cut -c1-160 | tee $O/kernel_stats_B.md
find $O -name "*.db" -size +6M -delete; find $O -name "*.csv" -size +4M -delete
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(f.split("/")[-1], d.get("value"), d["flight"]["solves_per_step"], d["flight"]["x_final_mean_m"], d.get("parity"))
    except Exception as e:
        print(f.split("/")[-1], "UNREADABLE", e)
PY
tail -3 $O/err.txt
