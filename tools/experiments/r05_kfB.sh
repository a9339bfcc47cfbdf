cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_kfmap_gpu.py tests/test_step_frames_gpu.py tests/test_keyframe_gpu.py -x -q 2>&1 | tail -5
B="--workload flight --keyframes 100 --points 3072 --T 1.0 --K 3 --no-parity --no-cpu-baseline"
mkdir -p gpurun_out/r05kf2
timeout 600 python bench.py $B --streams 8 --gang 4 --periods 120 > gpurun_out/r05kf2/B_8x4.json 2>gpurun_out/r05kf2/err.txt
timeout 600 python bench.py $B --streams 4 --gang 2 --periods 120 > gpurun_out/r05kf2/B_4x2.json 2>>gpurun_out/r05kf2/err.txt
timeout 600 python bench.py --workload flight --points 3072 --T 1.0 --K 3 --no-parity --no-cpu-baseline --streams 8 --gang 4 --periods 120 > gpurun_out/r05kf2/B0_8x4.json 2>>gpurun_out/r05kf2/err.txt
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r05kf2/ktB -o kt -- python bench.py $B --streams 1 --gang 2 --periods 120 > /dev/null 2>> gpurun_out/r05kf2/err.txt
db=$(find gpurun_out/r05kf2/ktB -name "*.db" | head -1); python tools/rocprof_summary.py $db | grep -v "at::native\|Cijk" | head -14
find gpurun_out/r05kf2 -name "*.db" -size +6M -delete; find gpurun_out/r05kf2 -name "*.csv" -size +4M -delete
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r05kf2/*.json")):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(f.split("/")[-1], d.get("value"), d["flight"]["solves_per_step"], d["flight"]["x_final_mean_m"])
    except Exception as e:
        print(f.split("/")[-1], "UNREADABLE", e)
PY
tail -3 gpurun_out/r05kf2/err.txt
