#!/bin/bash
# Samples socket power and clocks (rocm-smi) while bench.py runs the train step: tools/power_sample.sh [steps] > gpurun_out/power.txt
# The dense loops run below the 2.4 GHz the MFMA peak is quoted at; this records how far below and at what power.
STEPS="${1:-4000}"
python bench.py --steps "$STEPS" --warmup 5 --no-cpu-baseline --profile-steps 0 > /tmp/power_bench.json 2>/dev/null &
BP=$!
sleep 10     # import + allocation of the engine
for i in $(seq 1 40); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  echo "--- sample $i"
  rocm-smi --showpower --showclocks --showuse --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|GPU use|Temperature \(Sensor (edge|junction|memory)" | sed 's/^GPU\[0\]\s*: //'
  sleep 1
done
wait $BP
echo "--- bench"; cat /tmp/power_bench.json
echo "--- idle"
sleep 3
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | sed 's/^GPU\[0\]\s*: //'
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | sed 's/^GPU\[0\]\s*: //'
