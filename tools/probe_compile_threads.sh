#!/bin/bash
# probe_compile_threads.sh — dev: the drop-in's scene upload with 8 / 16 / 32 / 64 / 128 layout-compile threads (CRH_COMPILE_THREADS), one process alone and eight
# processes at once (what eight ranks of one node do to each other's compile): upload_ms and the compile phases per run.
cd "$(dirname "$0")/.." || exit 1
pick() { grep -o "upload_ms[^,]*\|layout compile [0-9.]* ms\|textures [0-9.]* ms\|triangles [0-9.]* ms" | paste - - - - ; }
for t in ${THREADS:-16 32 64}; do
	echo "== $t threads, one process"
	CRH_COMPILE_THREADS=$t RUNS=3 CRH_TRACE_UPLOAD=1 timeout 80 python tools/probe_dropin.py cfg2 2>&1 | pick
	[ -n "$ONE" ] && continue
	echo "== $t threads, eight processes at once"
	for p in 1 2 3 4 5 6 7 8; do (CRH_COMPILE_THREADS=$t RUNS=2 CRH_TRACE_UPLOAD=1 timeout 120 python tools/probe_dropin.py cfg2 2>&1 | pick > /tmp/pct_$t_$p.log) & done
	wait
	cat /tmp/pct_$t_*.log
done
