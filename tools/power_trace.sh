#!/bin/bash
# GPU box: socket power / shader clock sampled by rocm-smi (read-only) while the headline leg runs (4 videos in flight), and while the
# register-operand MFMA probe runs: is the clock the chip holds under the conv GEMMs a power limit?
# usage: bash tools/power_trace.sh  -> gpurun_out/power/{bench,idle}.txt + a summary on stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/power; rm -rf $O; mkdir -p $O; cd $R
sample() {   # $1 = output file, runs until the file $O/stop exists
  while [ ! -e $O/stop ]; do
    rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | tr -d '\n' >> $1; echo >> $1
    sleep 0.25
  done
}
rocm-smi --showpower --showclocks --showtemp --showmaxpower --json 2>/dev/null > $O/idle.txt
rocm-smi --showmaxpower --showperflevel 2>/dev/null | grep -v "^$" | head -12
rm -f $O/stop; sample $O/bench.txt & SP=$!
python bench.py --streams 4 --steps 48 --warmup 4 --cpu-frames 0 --no-r2 --no-config3 --no-memread-roofline --no-davis-val --no-drivers --no-session --no-power --value-repeats 1 --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline %.1f frames/s' % d['value'])"
touch $O/stop; wait $SP
python - <<PY
import json, re, statistics as st
rows = []
for line in open("$O/bench.txt"):
    try: c = json.loads(line).get("card0", {})
    except Exception: continue
    m = re.search(r"(\d+)Mhz", c.get("sclk clock speed:", ""))
    rows.append((float(c.get("Current Socket Graphics Package Power (W)", "nan")), float(m.group(1)) if m else float("nan"),
                 float(c.get("Temperature (Sensor junction) (C)", "nan"))))
for r in rows: print("  %6.0f W  sclk %5.0f MHz  junction %3.0f C" % r)
hot = [r for r in rows if r[0] > 0.8 * max(x[0] for x in rows)]
print("under load (%d samples): socket power median %.0f W, max %.0f W (cap: see above); sclk median %.0f MHz; junction max %.0f C" % (
    len(hot), st.median(r[0] for r in hot), max(r[0] for r in hot), st.median(r[1] for r in hot), max(r[2] for r in hot)))
PY
