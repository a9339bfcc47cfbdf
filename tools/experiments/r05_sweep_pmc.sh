# SQ / cache counters of the sweep's mark kernel (regime A, one stream), two passes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05swpmc; rm -rf $O; mkdir -p $O
CMD="python bench.py --workload flight --keyframes 3 --streams 1 --gang 2 --periods 20 --no-parity --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $O/a -o a -- $CMD > /dev/null 2> $O/err_a.txt
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b -o b -- $CMD > /dev/null 2> $O/err_b.txt
python - <<PY
import csv, glob, collections
for sub in "ab":
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs:
        print(sub, "no counter file"); print(open("$O/err_%s.txt" % sub).read()[-600:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "sweep_mark" in k or "mpc_solve" in k:
            acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: round(sum(v[len(v)//2:]) / max(1, len(v) - len(v)//2)) for c, v in cs.items()})
PY
rm -rf $O/a $O/b
