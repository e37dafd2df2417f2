"""Per-kernel PMC averages from several rocprofv3 passes (one counter group per pass, counters never combined with traces).
usage: python3 scripts/pmc.py OUT.json GROUP[,GROUP...] -- python3 prog.py args...     (GROUP = counters joined by '+')
Run on the GPU box from /tmp-safe cwd (the caller does: cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp)."""
import csv, glob, json, os, shutil, subprocess, sys

out_json = sys.argv[1]
groups = sys.argv[2].split(",")
cmd = sys.argv[sys.argv.index("--") + 1:]
res = {}
for gi, grp in enumerate(groups):
    d = os.path.join("gpurun_out", "pmc_tmp_%d" % gi)
    shutil.rmtree(d, ignore_errors=True)
    full = ["rocprofv3", "--pmc"] + grp.split("+") + ["--output-format", "csv", "-d", d, "--"] + cmd
    with open(os.path.join("gpurun_out", "pmc_tmp_%d.log" % gi), "w") as lf:
        rc = subprocess.call(full, stdout=lf, stderr=subprocess.STDOUT)
    print("pass %d (%s): rc=%d" % (gi, grp, rc), flush=True)
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            key = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            disp = (row.get("Dispatch_Id"),)
            acc.setdefault(key, {}).setdefault(disp, 0.0)
            acc[key][disp] += float(row["Counter_Value"])
    for (kern, ctr), per in acc.items():
        res.setdefault(kern, {})[ctr] = sum(per.values()) / len(per)
        res[kern]["launches"] = len(per)
    shutil.rmtree(d, ignore_errors=True)
json.dump(res, open(out_json, "w"), indent=1, sort_keys=True)
print("wrote", out_json)
