cd /root/repo
fmt='
import sys, json
j = json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print("%s value %.2f M pairs/s  ms/step %.2f" % (sys.argv[1], j["value"] / 1e6, j["ms_per_step"]), {n: round(v["ms_total"] / j["steps"], 2) for n, v in j["kernels"].items()})
'
python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "$fmt" default
CM_LIB=tests/_hostemu/libcmhot_pw2.so python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "$fmt" pw2
CM_LIB=tests/_hostemu/libcmhot_pw3.so python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "$fmt" pw3
