#!/bin/bash
# Board power and shader clock while the headline forward runs (rocm-smi sampled every 0.25 s beside bench.py), Winograd and direct
# kernels: are the block convs clock-bound or power-bound?   bash tools/power_probe.sh [outdir]
O=${1:-gpurun_out/power}; mkdir -p $O
rocm-smi --showmaxpower --showclocks --showpower > $O/idle.txt 2>&1
for mode in 1 0; do
  python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-secondary --winograd $mode > $O/bench_$mode.json 2> $O/bench_$mode.err &
  BP=$!
  : > $O/smi_$mode.txt
  while kill -0 $BP 2>/dev/null; do
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|junction\|mclk" >> $O/smi_$mode.txt
    echo "---" >> $O/smi_$mode.txt
    sleep 0.25
  done
  wait $BP
done
python3 - "$O" <<'PY'
import json, re, sys
o = sys.argv[1]
print(open(o + '/idle.txt').read()[:1500])
for mode in ('1', '0'):
    txt = open(f'{o}/smi_{mode}.txt').read()
    pw = [float(x) for x in re.findall(r'Package Power \(W\):\s*([0-9.]+)', txt)]
    sc = [float(x) for x in re.findall(r'sclk clock level: \w+: \((\d+)Mhz\)', txt)]
    tj = [float(x) for x in re.findall(r'junction\) \(C\):\s*([0-9.]+)', txt)]
    line = [l for l in open(f'{o}/bench_{mode}.json') if l.startswith('{')]
    d = json.loads(line[-1]) if line else {}
    top = sorted(pw)[len(pw) // 2:] if pw else [0]
    print(f'--winograd {mode}: {d.get("value")} frames/s, frac {d.get("roofline", {}).get("frac")};  {len(pw)} samples: power median of the upper half {sorted(top)[len(top)//2]:.0f} W, max {max(pw or [0]):.0f} W; '
          f'sclk samples {sorted(set(sc))[-6:]} MHz median {sorted(sc)[len(sc)//2] if sc else None}; junction max {max(tj or [0]):.0f} C')
PY
