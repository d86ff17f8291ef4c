#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 600 python3 bench.py --timed-only --steps 5 --warmup 1 > $O/bench_timed_7.json 2> $O/bench_timed_7.err
python3 -c "import json; d=json.loads([l for l in open('$O/bench_timed_7.json') if l.startswith('{')][-1]); print('timed-only', d['value'], d['ms_per_step'])"
timeout 900 python3 bench.py --steps 5 --no-cpu-baseline --sequence-pairs 0 --full-size-pairs 0 > $O/bench_full_7.json 2> $O/bench_full_7.err
python3 -c "
import json; d=json.loads([l for l in open('$O/bench_full_7.json') if l.startswith('{')][-1]); print('full', d['value'], d['ms_per_step'])
for w in d['other_workloads']: print(round(w['value'] / 1e9, 3), w.get('ms_per_step'), w['workload'][:90])"
timeout 600 python3 bench.py --timed-only --steps 5 --warmup 1 > $O/bench_timed_7b.json 2> $O/bench_timed_7b.err
python3 -c "import json; d=json.loads([l for l in open('$O/bench_timed_7b.json') if l.startswith('{')][-1]); print('timed-only again', d['value'], d['ms_per_step'])"
