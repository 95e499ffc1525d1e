rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -i "sclk\|power\|perf" | head -8
( for i in 1 2 3 4 5 6; do sleep 0.25; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Power\|Average Graphics" | tr '\n' ' '; echo; done ) &
timeout 60 tools/exp/clock_test
wait
rocm-smi --showmaxpower --showpowerplay 2>/dev/null | head -20
