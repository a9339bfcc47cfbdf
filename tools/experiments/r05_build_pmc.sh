# SQ counters of the index build and the step's search kernel (cold step, one stream): what inside a CU bounds them?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05bpmc; rm -rf $O; mkdir -p $O
CMD="python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --no-parity --steady-steps 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/a -o a -- $CMD > /dev/null 2> $O/err_a.txt
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b -o b -- $CMD > /dev/null 2> $O/err_b.txt
python - <<PY
import csv, glob, collections
for sub in "ab":
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs:
        print(sub, "no counter file"); print(open("$O/err_%s.txt" % sub).read()[-600:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "kd_build_kernel" in k or "step_knn_grid" in k or "plan_pack" in k:
            acc[k.split("(")[0][-36:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: round(sum(v[len(v)//2:]) / max(1, len(v) - len(v)//2)) for c, v in cs.items()})
PY
rm -rf $O/a $O/b
