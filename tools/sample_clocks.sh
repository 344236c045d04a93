#!/bin/bash
# Sample GPU clock and power while a command runs: tools/sample_clocks.sh <outfile> -- <command...>
OUT=$1; shift; shift
( while true; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.2; done ) > $OUT &
SPID=$!
"$@"
kill $SPID 2>/dev/null
