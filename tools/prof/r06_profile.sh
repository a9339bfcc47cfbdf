#!/bin/bash
# Round-6 profile set (run on the GPU box through gpurun; outputs under gpurun_out/$TAG, summaries are copied to profiles/
# by tools/prof/r06_collect.py).  Usage: tools/prof/r06_profile.sh <tag>
TAG=${1:-r06a}
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp; cd $ROOTD
OUT=gpurun_out/$TAG; mkdir -p $OUT; : > $OUT/bench.err
# 2. kernel trace + stats of the same command (10 slots x 4 steps per launch in flight) and of the single-stream run (clean kernel times)
rocprofv3 --kernel-trace --stats -d $OUT/kt20 -o kt -- python bench.py --steps 256 --no-cpu-baseline --no-parity --steady-steps 0 > $OUT/bench_under_rocprof.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt1 -o kt -- python bench.py --steps 64 --warmup 4 --streams 1 --no-cpu-baseline --no-parity --steady-steps 0 > $OUT/bench_streams1.json 2>> $OUT/bench.err
# 3. HBM traffic (separate PMC passes, single stream so that launches do not overlap)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --no-parity --steady-steps 0 > /dev/null 2>> $OUT/bench.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --no-parity --steady-steps 0 > /dev/null 2>> $OUT/bench.err
# 4. what bounds the solve: issue / LDS counters (two passes of 8 SQ counters); --gang 1: 256-scene launches = ONE wave per CU
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq_a -o a -- python bench.py --steps 8 --warmup 2 --streams 1 --gang 1 --no-cpu-baseline --no-parity --steady-steps 0 > /dev/null 2>> $OUT/bench.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq_b -o b -- python bench.py --steps 8 --warmup 2 --streams 1 --gang 1 --no-cpu-baseline --no-parity --steady-steps 0 > /dev/null 2>> $OUT/bench.err
find $OUT -name "*.csv" -o -name "*.db" | head -40
tail -3 $OUT/bench.err
# 1. the summaries bench.py quotes (profiles/r06_*: rocprofv3's durations and PMC traffic of these very commands), then the bench
#    line itself (default: 2048 steps) and the 20-step line the driver asks for
python tools/experiments/ms_parts.py > $OUT/ms_parts.txt 2>> $OUT/bench.err
python tools/prof/r06_collect.py $TAG --profiles-only
python bench.py > $OUT/bench.json 2>> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $OUT/bench_20steps.json 2>> $OUT/bench.err
# 5. the collective path on one GPU: bench.py under torch.distributed.run with one rank (one ncclAllGather per sweep, amk_shard_gather)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-parity > $OUT/bench_torchrun_1rank.json 2>> $OUT/bench.err
# 6. other BASELINE sizes and launch shapes (not the headline)
python bench.py --points 200000 --T 1.0 --steps 64 --warmup 4 --streams 8 --no-cpu-baseline --no-parity --steady-steps 0 > $OUT/bench_c5size.json 2>> $OUT/bench.err
python bench.py --points 5000 --T 0.33 --K 3 --steps 512 --no-cpu-baseline --no-parity --steady-steps 0 > $OUT/bench_c1size.json 2>> $OUT/bench.err
python bench.py --steps 64 --warmup 4 --scenes 1 --streams 1 --gang 1 --no-cpu-baseline --no-parity --steady-steps 0 > $OUT/bench_single_robot.json 2>> $OUT/bench.err
python bench.py --steps 32 --warmup 4 --scenes 2048 --streams 1 --gang 1 --no-cpu-baseline --no-parity --steady-steps 0 > $OUT/bench_s2048_streams1.json 2>> $OUT/bench.err
python bench.py --ipm-max-iter 40 --no-cpu-baseline --no-parity > $OUT/bench_cap40.json 2>> $OUT/bench.err
python bench.py --streams 20 --gang 1 --no-cpu-baseline --no-parity > $OUT/bench_20x1.json 2>> $OUT/bench.err
python bench.py --streams 20 --gang 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $OUT/bench_20x1_20steps.json 2>> $OUT/bench.err
# 7. round 4: the closed-loop workload, the reference's tie order, bench.py launching its own rank
python bench.py --workload flight > $OUT/bench_flight.json 2>> $OUT/bench.err
python bench.py --tie-order 1 --steps 512 --steady-steps 0 > $OUT/bench_tie_order1.json 2>> $OUT/bench.err
python bench.py --gpus 1 --no-cpu-baseline --no-parity > $OUT/bench_self_launch.json 2>> $OUT/bench.err
python -m pytest tests/test_flight_gpu.py -x -q -s > $OUT/flight_tests.txt 2>&1; cp gpurun_out/flight_c2_gpu_vs_oracle.json $OUT/ 2>/dev/null
rocprofv3 --kernel-trace --stats -d $OUT/kt_flight -o kt -- python bench.py --workload flight --periods 24 --no-parity > /dev/null 2>> $OUT/bench.err
python tools/experiments/burst_timeline.py 10 4 20 > $OUT/burst_timeline_10x4.txt 2>> $OUT/bench.err
# 8. round 5-6: the keyframe map in the closed loop (10 slots x gang 2: the pool holds (N + 2) index slots per scene), the budgeted solve
python bench.py --workload flight --streams 10 --gang 2 > $OUT/bench_flight_10x2.json 2>> $OUT/bench.err
python bench.py --workload flight --streams 10 --gang 2 --keyframes 3 > $OUT/bench_flight_keyframes3.json 2>> $OUT/bench.err
# 8b. the reference's own configuration (3072-point frames, N = 30, K = 3), with its default keyframe map (max_frame_count 100) and without
python bench.py --workload flight --config yaml --keyframes 100 > $OUT/bench_flight_yaml_keyframes100.json 2>> $OUT/bench.err
python bench.py --workload flight --config yaml --keyframes 100 --streams 12 --no-cpu-baseline --no-parity > $OUT/bench_flight_yaml_keyframes100_12x4.json 2>> $OUT/bench.err
python bench.py --workload flight --config yaml --no-cpu-baseline --no-parity > $OUT/bench_flight_yaml_single_frame.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt_flight_kf3 -o kt -- python bench.py --workload flight --keyframes 3 --streams 1 --gang 2 --no-parity --no-cpu-baseline > /dev/null 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt_flight_yaml -o kt -- python bench.py --workload flight --config yaml --keyframes 100 --streams 1 --no-parity --no-cpu-baseline > /dev/null 2>> $OUT/bench.err
python bench.py --solve-budget 16 --no-cpu-baseline --no-parity > $OUT/bench_budget16.json 2>> $OUT/bench.err
python bench.py --solve-budget 16 --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $OUT/bench_budget16_20steps.json 2>> $OUT/bench.err
cat $OUT/ms_parts.txt
# the summaries (profiles/r06_*) are made HERE, on the box, and travel back under $OUT/profiles; the raw traces are trimmed to
# what fits gpurun's 64 MiB return channel (kernel-trace databases of the short runs stay, the long ones go)
python tools/prof/r06_collect.py $TAG
mkdir -p $OUT/profiles && cp profiles/r06_* $OUT/profiles/
find $OUT -name "*.db" -size +6M -delete
find $OUT -name "*_kernel_trace.csv" -size +4M -delete
du -sh $OUT; ls $OUT $OUT/profiles
tail -5 $OUT/bench.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench*.json")):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(f.split("/")[-1], d.get("value"), d.get("value_steady_state"))
    except Exception as e:
        print(f.split("/")[-1], "UNREADABLE", e)
PY
