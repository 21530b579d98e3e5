export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wwt -o p -- python /root/repo/scripts/wino_wgrad_check.py 8 > /tmp/ww.log 2>&1
tail -9 /tmp/ww.log
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/wwt/**/p_kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-90s n=%5s avg=%9.1f us tot=%8.2f ms"%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
