# Is the yaml-configuration flight with the keyframe map bound by the host's submit loop?  Same loop with 8 flights per batch (the GPU
# work is ~nothing: frames per second = what the host can submit), then more batches in flight at the full 256.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="--workload flight --keyframes 100 --points 3072 --T 1.0 --K 3 --no-parity --no-cpu-baseline --periods 120"
B0="--workload flight --points 3072 --T 1.0 --K 3 --no-parity --no-cpu-baseline --periods 120"
O=gpurun_out/r05kf3; mkdir -p $O
python bench.py $B --streams 8 --gang 4 --scenes 8 > $O/host_B_8x4_s8.json 2> $O/err.txt
python bench.py $B0 --streams 8 --gang 4 --scenes 8 > $O/host_B0_8x4_s8.json 2>> $O/err.txt
python bench.py $B --streams 8 --gang 8 > $O/B_8x8.json 2>> $O/err.txt
python bench.py $B --streams 12 --gang 4 > $O/B_12x4.json 2>> $O/err.txt
python bench.py $B --streams 16 --gang 2 > $O/B_16x2.json 2>> $O/err.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(f.split("/")[-1], d.get("value"), "frames/s", round(d["value"] / d["config"]["scenes_per_gpu"]), "host ms/frame", d["config"]["host_submit_ms_per_step"])
    except Exception as e:
        print(f.split("/")[-1], "UNREADABLE", e)
PY
tail -3 $O/err.txt
