#!/bin/bash
# Round-4 profiles, second half: the content paths after the colour / row-search work (the headline step's kernels did not change
# since scripts/prof_r04.sh ran), and the plain bench lines of the final code.  Raw output joins gpurun_out/r04prof; the same
# summariser (scripts/summarise_r04.py) writes profiles/r04/.
# rocprofv3 rules on this pool: the program itself directly after `--`; --pmc passes with --kernel-trace only.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04prof"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
st() { rm -rf "$O/$1"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$1" -o s -- python3 "$B" "${@:2}" > "$O/$1.log" 2>&1 || echo "$1 failed"; }
pm() { local name=$1 set=$2; shift 2; rm -rf "$O/$name"; timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/$name" -o p -- python3 "$B" "$@" > "$O/$name.log" 2>&1; echo "$name [$set] rc=$?"; }
st stats_graph --steps 200 --no-extras --no-cpu-baseline
st stats_content --content-only --steps 50 --no-graph
pm pmcC_fetch "FETCH_SIZE" --content-only --steps 10 --no-graph
pm pmcC_write "WRITE_SIZE" --content-only --steps 10 --no-graph
st stats_content_full --content-full-only --steps 30 --no-graph
pm pmcF_fetch "FETCH_SIZE" --content-full-only --steps 10 --no-graph
pm pmcF_write "WRITE_SIZE" --content-full-only --steps 10 --no-graph
cd "$R"
timeout -k 10 600 python3 bench.py > "$O/bench_line.json" 2> "$O/bench_line.err"; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --points 8000000 --steps 20 --no-extras --no-cpu-baseline > "$O/bench_8M.json" 2> "$O/bench_8M.err"; echo "bench 8M rc=$?"
timeout -k 10 300 python3 bench.py --points 32000000 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > "$O/bench_32M.json" 2> "$O/bench_32M.err"; echo "bench 32M rc=$?"
python3 "$R/scripts/summarise_r04.py" "$O" "$O/summary" 2>&1 | tail -40
