# Riccati temporaries in the trial states' storage (LdsMap): the solver tests, then the rates that depend on the solve's LDS
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05lds; mkdir -p $O; : > $O/err.txt
timeout 1500 python -m pytest tests/test_mpc_solve_gpu.py tests/test_mpc_resume_gpu.py tests/test_mpc_fp32_gpu.py tests/test_step_gpu.py tests/test_mpc_eval_gpu.py -x -q 2>&1 | tail -4
B0="--workload flight --points 3072 --T 1.0 --K 3 --no-parity --no-cpu-baseline --periods 120 --streams 12 --gang 4"
B="--workload flight --keyframes 100 --points 3072 --T 1.0 --K 3 --no-parity --no-cpu-baseline --periods 120 --streams 12 --gang 4"
python bench.py --no-cpu-baseline --no-parity > $O/default.json 2>> $O/err.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/steps20.json 2>> $O/err.txt
python bench.py --points 200000 --T 1.0 --steps 64 --warmup 4 --streams 8 --no-cpu-baseline --no-parity --steady-steps 0 > $O/c5size.json 2>> $O/err.txt
python bench.py $B0 > $O/B0.json 2>> $O/err.txt
python bench.py $B > $O/B.json 2>> $O/err.txt
python bench.py --workload flight > $O/flight.json 2>> $O/err.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(f.split("/")[-1], d.get("value"), d.get("value_steady_state"))
    except Exception as e:
        print(f.split("/")[-1], "UNREADABLE", e)
PY
tail -3 $O/err.txt
