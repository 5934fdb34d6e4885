mkdir -p gpurun_out/ppmc
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $ROOT/gpurun_out/ppmc/$cnt -- $ROOT/build_variants/sweep_pair_probe 512 512 > /dev/null 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, re
for cnt, corr in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    f = glob.glob(f"gpurun_out/ppmc/{cnt}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(cnt, k[:40], "launches", len(v), "MB/launch %.1f" % (corr * sum(v) / len(v) * 1024 / 1e6))
PY
