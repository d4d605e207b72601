#!/bin/bash
# sample power / clocks while a command runs: tools/smi_sample.sh <cmd...>
"$@" &
PID=$!
sleep 16
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use|fclk" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.7
done
wait $PID
